// bwd_window_launch.h — host-side launcher interface of the LDS-window backward kernels (see bwd_window.h).
// Compiled once per channel-group shift, like the forward instances (bwd_window_inst.hip, -DPSF_TGS=0..6).
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
hipError_t launch_dw_win_tgs(int rows, int L, const BwdWinArgs& a);
template <int TGS>
hipError_t launch_dv_win_tgs(int rows, int L, const BwdWinArgs& a);

}  // namespace psf
