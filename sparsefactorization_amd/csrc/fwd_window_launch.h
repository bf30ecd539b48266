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
// ... and 512 threads for rows of exactly 32 channels (8 lanes per row): 128-row tiles with 2 rows per thread, one far link
// fewer than the 64-row tiles of 256 threads (knob "fwd_wide" = 3; measured in profiles/r04ai_*).
constexpr int kFwdMidTgs = 3, kFwdMidThreads = 512;
constexpr bool win_pair_compiled(int tgs, int nt) {
  return (nt == 256 && tgs >= 0 && tgs <= kWinTgsMax) || (nt == kWideThreads && tgs == kWideTgs) ||
         (nt == kFwdMidThreads && tgs == kFwdMidTgs);
}

// Rows per thread R that are compiled, and the default.
// r01 measurements (us per launch): cfg2 (C=8) R=2 27.5 < R=1 29.1; cfg3 (C=128) R=2 16.1 < R=4 16.9 < R=8 19.8;
// cfg4 (C=32) R=1 6.0 ~ R=2 6.1 < R=4 6.4.
constexpr bool win_rows_compiled(int r) { return r == 1 || r == 2; }
constexpr int win_default_rows(int /*tgs*/) { return 2; }

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
  float* wfar;      // training: link-major side copy of W's columns >= far_k0 ([B, L-far_k0, N]), or nullptr
  int far_k0;
  hipStream_t stream;
};

// Launch the instance (L, TGS, R = rows, NT). Returns hipErrorInvalidValue when that instance is not compiled.
template <int TGS, int NT>
hipError_t launch_fwd_win(int rows, int L, const FwdWinArgs& a);

// window geometry for a (TGS, rows, NT) triple — mirrors FwdWinCfg
inline int win_tile_rows(int tgs, int rows, int nt) { return (nt >> tgs) * rows; }

}  // namespace psf
