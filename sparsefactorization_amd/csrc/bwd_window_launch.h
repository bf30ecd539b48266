// bwd_window_launch.h — host-side launcher interface of the LDS-window backward kernels (see bwd_window.h).
// Compiled per (TGS, NT) pair like the forward instances (bwd_window_inst.hip, -DPSF_TGS=.. [-DPSF_NT=1024]).
// dW is compiled at 256 threads only: its per-row reduction needs every channel of a row in one workgroup, so it
// cannot use the channel-chunked wide-row configuration.
#pragma once

#include "fwd_window_launch.h"

namespace psf {

struct BwdWinArgs {
  const float* dZ;
  const float* WV;  // dV: W [B,N,L];  dW: V [B or 1,N,C]
  float* out;       // dV: dV [B,N,C]; dW: dW [B,N,L]
  Geom gm;
  Offsets offs;
  int64_t w_total;  // B*N*L
  bool edge;
  hipStream_t stream;
  const float* V2 = nullptr;    // fused step (bwd_fused.h): WV = W, V2 = V, out = dV, out2 = dW
  float* out2 = nullptr;
  int ablate = 0;               // fused step, diagnostic builds only (bwd_fused.h: ABL)
  int wg_per_cu = 0;            // fused step: 0 = whatever fits; n > 0: at most n workgroups per CU (by requesting more LDS)
};

template <int TGS>
hipError_t launch_dw_win(int rows, int L, const BwdWinArgs& a);
// chunk-looping dW (bwd_dw_chunk.h) for rows of >= 32 channels: 8 or 16 lanes per row chunk (TGS 3 / 4), 256 threads,
// one row per thread. r02 sweep (profiles/r02c_dw_sweep*.log, us per launch at ListOps N=2000 C=128 / genome C=32):
// TG=8 R=1 21.0 (15.3 in one launch) / 20.6; TG=16 R=1 15.2 / 20.7; R=2 25.5 / 24.8; 1024 threads x 1 row 28.2 / 33.5;
// TG=4 27.6 / 26.3; whole-row kernel 29.0 / 26.1 — short tiles with many workgroups per CU win, so only R = 1 is built.
constexpr int kDwChunkTgsMin = 3, kDwChunkTgsMax = 4;
template <int TGS>
hipError_t launch_dw_chunk(int L, const BwdWinArgs& a);
// dV: the forward's (TGS, NT) pairs plus 512 threads x 1 row per thread for narrow rows (TGS <= 3): the same 256-row
// tile as 256 threads x 2 rows at C = 8, with twice the waves per workgroup sharing the staged W tiles — r02 lab
// (profiles/dvlab.hip): 28.65 vs 29.05 us at cfg2 next to the shipped kernel's 30.07.
constexpr int kDvMidThreads = 512, kDvMidTgsMax = 1;  // rows of <= 8 channels, where it is the automatic choice (the instances
                                                       // for 16 and 32 channels were reachable by knob only: removed in round 5)
constexpr int kFusedTgsMax = 5;                        // fused step: rows of <= 128 channels (psf_chord.hip: fused_step_width)
constexpr bool dv_pair_compiled(int tgs, int nt) {
  return win_pair_compiled(tgs, nt) || (nt == kDvMidThreads && tgs >= 0 && tgs <= kDvMidTgsMax);
}
template <int TGS, int NT>
hipError_t launch_dv_win(int rows, int L, const BwdWinArgs& a);
// fused dV + dW step (bwd_fused.h): 256 threads x 1 row (tile = 256 >> TGS rows), rows of exactly 4 << TGS channels,
// TGS <= kFusedTgsMax; compiled in the -DPSF_NT=512 units (whose 512-thread fused instances round 5 removed: 256-thread
// tiles stayed ahead in every sweep, profiles/r04am_bwd_fused_wg_sweep.log)
template <int TGS>
hipError_t launch_bwd_fused(int L, const BwdWinArgs& a);
// the same for any sequence length >= two tiles and any far offsets (per-lane wrap, partial last tile, any W / dW alignment)
template <int TGS>
hipError_t launch_bwd_fused_edge(int L, const BwdWinArgs& a);

}  // namespace psf
