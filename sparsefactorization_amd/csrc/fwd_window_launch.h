// fwd_window_launch.h — host-side launcher interface of the LDS-window forward kernels.
//
// The kernels are templates over (L, TGS, R, NT, RES, EDGE): several hundred instances. They are compiled in one
// translation unit per (channel-group shift TGS, workgroup size NT) pair (fwd_window_inst.hip built with
// -DPSF_TGS=.. -DPSF_NT=.., in parallel) and reached through launch_fwd_win<TGS, NT>, which this header declares
// for the dispatcher in psf_chord.hip.
#pragma once

#include "psf_common.h"

namespace psf {

constexpr int kWinLmin = 4, kWinLmax = 20;  // compiled link counts
constexpr int kLdsPerCu = 160 * 1024;       // gfx950
constexpr int kWinTgsMax = 6;               // TG = 1 << TGS <= 64 lanes share a row

// Compiled (TGS, NT) pairs: every TGS at 256 threads, plus the "wide-row" configuration: 8 lanes per row
// (32 channels = one 128-byte line per row chunk; wider rows are split into channel chunks) at 1024 threads,
// i.e. 256-row tiles. For C >= 64 that turns all but L-10 links into near links (see fwd_window.h).
constexpr int kWideTgs = 3, kWideThreads = 1024;
// (Round 4 also built 512 threads for rows of exactly 32 channels — equal at the genome and IMDb shapes, profiles/r04ai_* —
// and round 5 removed it: no automatic rule selected it.)
// (... and for the chunks of wider rows in round 5: 3-8 % slower than the 1024-thread chunks at N <= 4096, 3 % faster at
// N = 16384 x 64 channels, profiles/r05m_fwd_wide_mid.log — not kept either.)
constexpr bool win_pair_compiled(int tgs, int nt) {
  return (nt == 256 && tgs >= 0 && tgs <= kWinTgsMax) || (nt == kWideThreads && tgs == kWideTgs);
}

// Rows per thread: the forward and dV kernels are compiled for R = 2 (dV also 512 threads x 1 row), dW for R = 1.
// r01 measurements (us per launch): cfg2 (C=8) R=2 27.5 < R=1 29.1; cfg3 (C=128) R=2 16.1 < R=4 16.9 < R=8 19.8;
// cfg4 (C=32) R=1 6.0 ~ R=2 6.1 < R=4 6.4; dV R=2 31.3 vs R=1 32.7; dW R=1 22.9 vs R=2 28.7. The other instances were
// reachable through tuning knobs only ("fwd_rows", "bwd_rows") and went with them in round 5.

struct FwdWinArgs {
  const float* W;
  const float* V;
  const float* res;  // nullptr: no residual (selects the RES = false kernels)
  float* out;
  Geom gm;
  Offsets offs;
  int64_t w_total;  // B*N*L
  bool edge;        // true: the launch contains tiles that are not full (EDGE = true kernels)
  int wg_per_cu;    // 0: whatever fits; n > 0: at most n workgroups per CU (enforced by requesting more LDS)
  hipStream_t stream;
};

// Launch the instance (L, TGS, R = rows, NT). Returns hipErrorInvalidValue when that instance is not compiled.
template <int TGS, int NT>
hipError_t launch_fwd_win(int rows, int L, const FwdWinArgs& a);

// window geometry for a (TGS, rows, NT) triple — mirrors FwdWinCfg
inline int win_tile_rows(int tgs, int rows, int nt) { return (nt >> tgs) * rows; }
// Four rows per thread (forward only): the (TGS, NT) pairs it is compiled for — rows of 16, 32 and 64 channels on 256-thread
// workgroups (tiles of 256 / 128 / 64 rows: one far link fewer than with two rows per thread). Round 6, chains that keep every
// step, W rotating beyond the Infinity Cache (profiles/r06j_fwd_rows_sweep.log, us per step, two rows / four): C = 16 x 16384:
// 41.7 / 39.3; C = 32 x 16384 (genome): 22.6 / 21.9; x 4096: 11.0 / 10.8; x 1024 (Pathfinder): 6.34 / 6.20; C = 64 x 16384: 23.3 /
// 22.2 — and the other way round at 8 channels (25.8 / 27.1: 62 KB of LDS per 512-row tile) and a wash at 128 (44.6 / 44.1): not
// compiled there. (A lab build widens the range: PSF_HIPCC_EXTRA="-DPSF_ROWS4_TGS_MIN=1 -DPSF_ROWS4_TGS_MAX=5".)
#ifndef PSF_ROWS4_TGS_MIN
#define PSF_ROWS4_TGS_MIN 2
#define PSF_ROWS4_TGS_MAX 4
#endif
constexpr bool win_rows4_compiled(int tgs, int nt) { return nt == 256 && tgs >= PSF_ROWS4_TGS_MIN && tgs <= PSF_ROWS4_TGS_MAX; }

}  // namespace psf
