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
};

template <int TGS>
hipError_t launch_dw_win(int rows, int L, const BwdWinArgs& a);
template <int TGS, int NT>
hipError_t launch_dv_win(int rows, int L, const BwdWinArgs& a);

}  // namespace psf
