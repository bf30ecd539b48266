// bwd_window_inst.hip — instantiates the LDS-window backward kernels for ONE (channel-group shift, workgroup
// size) pair. Built once per compiled pair (-DPSF_TGS=0..6 [-DPSF_NT=1024]); see build.py.
#ifndef PSF_TGS
#error "compile with -DPSF_TGS=<0..6>"
#endif
#ifndef PSF_NT
#define PSF_NT 256
#endif

#include <atomic>

#include "bwd_dw_chunk.h"
#include "bwd_fused.h"
#include "bwd_window.h"
#include "bwd_window_launch.h"

namespace psf {
namespace {

template <typename K>
hipError_t raise_lds_limit(K kern, int bytes, std::atomic<int>& done) {
  if (bytes > 48 * 1024 && !done.load()) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return e;
    done.store(1);
  }
  return hipSuccess;
}

template <int L, int TGS, int R, int NT, bool EDGE>
hipError_t launch_dw(const BwdWinArgs& a) {
  using Cfg = BwdWinCfg<float, L, TGS, R, NT>;
  auto kern = chord_dw_win_k<float, L, TGS, R, NT, EDGE>;
  static std::atomic<int> done{0};
  if (hipError_t e = raise_lds_limit(kern, Cfg::lds_dw, done); e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(a.gm.nblocks), dim3(NT), Cfg::lds_dw, a.stream, a.dZ, a.WV, a.out, a.gm, a.offs,
                     a.w_total);
  return hipGetLastError();
}

template <int L, int TGS, int R, int NT, bool EDGE>
hipError_t launch_dv(const BwdWinArgs& a) {
  using Cfg = BwdWinCfg<float, L, TGS, R, NT>;
  auto kern = chord_dv_win_k<float, L, TGS, R, NT, EDGE>;
  static std::atomic<int> done{0};
  if (hipError_t e = raise_lds_limit(kern, Cfg::lds_dv, done); e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(a.gm.nblocks), dim3(NT), Cfg::lds_dv, a.stream, a.dZ, a.WV, a.out, a.gm, a.offs,
                     a.w_total);
  return hipGetLastError();
}

template <int L, int TGS, int R, int NT, bool EDGE>
hipError_t launch_dwc(const BwdWinArgs& a) {
  using Cfg = DwChunkCfg<L, TGS, R, NT>;
  auto kern = chord_dw_chunk_k<L, TGS, R, NT, EDGE>;
  static std::atomic<int> done{0};
  if (hipError_t e = raise_lds_limit(kern, Cfg::lds_bytes, done); e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(a.gm.nblocks), dim3(NT), Cfg::lds_bytes, a.stream, a.dZ, a.WV, a.out, a.gm, a.offs,
                     a.w_total);
  return hipGetLastError();
}

template <int TGS, int R, int NT>
hipError_t launch_dwc_L(int L, const BwdWinArgs& a) {
  switch (L) {
#define PSF_CASE(LL) \
  case LL:           \
    return a.edge ? launch_dwc<LL, TGS, R, NT, true>(a) : launch_dwc<LL, TGS, R, NT, false>(a);
    PSF_CASE(4) PSF_CASE(5) PSF_CASE(6) PSF_CASE(7) PSF_CASE(8) PSF_CASE(9) PSF_CASE(10) PSF_CASE(11)
    PSF_CASE(12) PSF_CASE(13) PSF_CASE(14) PSF_CASE(15) PSF_CASE(16) PSF_CASE(17) PSF_CASE(18)
    PSF_CASE(19) PSF_CASE(20)
#undef PSF_CASE
    default:
      return hipErrorInvalidValue;
  }
}

template <int TGS, int R, int NT, bool DW>
hipError_t launch_L(int L, const BwdWinArgs& a) {
  switch (L) {
#define PSF_CASE(LL)                                                                                            \
  case LL:                                                                                                      \
    if constexpr (DW) return a.edge ? launch_dw<LL, TGS, R, NT, true>(a) : launch_dw<LL, TGS, R, NT, false>(a); \
    else return a.edge ? launch_dv<LL, TGS, R, NT, true>(a) : launch_dv<LL, TGS, R, NT, false>(a);
    PSF_CASE(4) PSF_CASE(5) PSF_CASE(6) PSF_CASE(7) PSF_CASE(8) PSF_CASE(9) PSF_CASE(10) PSF_CASE(11)
    PSF_CASE(12) PSF_CASE(13) PSF_CASE(14) PSF_CASE(15) PSF_CASE(16) PSF_CASE(17) PSF_CASE(18)
    PSF_CASE(19) PSF_CASE(20)
#undef PSF_CASE
    default:
      return hipErrorInvalidValue;
  }
}

}  // namespace

#if PSF_NT == 256
template <int TGS>
hipError_t launch_dw_win(int rows, int L, const BwdWinArgs& a) {
  if (rows == 1) return launch_L<TGS, 1, 256, true>(L, a);  // (dW: one row per thread only)
  return hipErrorInvalidValue;
}
template hipError_t launch_dw_win<PSF_TGS>(int rows, int L, const BwdWinArgs& a);

#if PSF_TGS >= 3 && PSF_TGS <= 4
template <int TGS>
hipError_t launch_dw_chunk(int L, const BwdWinArgs& a) {
  static_assert(TGS >= kDwChunkTgsMin && TGS <= kDwChunkTgsMax, "chunk lanes per row");
  return launch_dwc_L<TGS, 1, 256>(L, a);
}
template hipError_t launch_dw_chunk<PSF_TGS>(int L, const BwdWinArgs& a);
#endif
#endif

#if PSF_NT != 512 || PSF_TGS <= 1  // (dV on 512 threads x 1 row is compiled for rows of <= 8 channels: bwd_window_launch.h)
template <int TGS, int NT>
hipError_t launch_dv_win(int rows, int L, const BwdWinArgs& a) {
  static_assert(dv_pair_compiled(TGS, NT), "not a compiled (TGS, NT) pair");
  if constexpr (NT == kDvMidThreads) {  // 512 threads x 1 row, or two rows per thread
    if (rows == 1) return launch_L<TGS, 1, NT, false>(L, a);
  } else {
    if (rows == 2) return launch_L<TGS, 2, NT, false>(L, a);
  }
  return hipErrorInvalidValue;
}
template hipError_t launch_dv_win<PSF_TGS, PSF_NT>(int rows, int L, const BwdWinArgs& a);
#endif

#if PSF_NT == 512
namespace {
template <int L, int TGS, int NT, int ABL = 0>
hipError_t launch_fused(const BwdWinArgs& a) {
  using Cfg = BwdFusedCfg<L, TGS, NT>;
#ifdef PSF_BWD_ABLATE_LAB  // diagnostic builds: the cfg2 instance with parts left out (bwd_fused.h, ABL)
  if constexpr (ABL == 0 && L == 15 && TGS == 1 && NT == 256) {
    switch (a.ablate) {
#define PSF_ABL(X) \
  case X:          \
    return launch_fused<L, TGS, NT, X>(a);
      PSF_ABL(1) PSF_ABL(2) PSF_ABL(3) PSF_ABL(4) PSF_ABL(8) PSF_ABL(12) PSF_ABL(15) PSF_ABL(16) PSF_ABL(32) PSF_ABL(48)
      PSF_ABL(63) PSF_ABL(64) PSF_ABL(79) PSF_ABL(112) PSF_ABL(115) PSF_ABL(124) PSF_ABL(128) PSF_ABL(512) PSF_ABL(640)
      PSF_ABL(256) PSF_ABL(258) PSF_ABL(320) PSF_ABL(368)
#undef PSF_ABL
      default:
        break;
    }
  }
#endif
  auto kern = chord_bwd_fused_k<L, TGS, NT, ABL>;
  int lds = Cfg::lds_bytes + ((ABL & 256) ? 2 * BwdWinCfg<float, L, TGS, 1, NT>::NF * NT * 16 : 0);
  if (a.wg_per_cu > 0) {  // occupancy limiter as in the forward launcher (fwd_window_inst.hip)
    const int floor_bytes = kLdsPerCu / (a.wg_per_cu + 1) + 256;
    if (floor_bytes > lds && floor_bytes <= 64 * 1024) lds = floor_bytes;
  }
  static std::atomic<int> done{0};
  if (lds > 48 * 1024 && done.load() < lds) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    if (e != hipSuccess) return e;
    done.store(lds);
  }
  hipLaunchKernelGGL(kern, dim3(a.gm.nblocks), dim3(NT), lds, a.stream, a.dZ, a.WV, a.V2, a.out2, a.out,
                     a.gm, a.offs, a.w_total);
  return hipGetLastError();
}
}  // namespace
template <int TGS>
hipError_t launch_bwd_fused(int L, const BwdWinArgs& a) {
  switch (L) {
#define PSF_CASE(LL) \
  case LL:           \
    return launch_fused<LL, TGS, 256>(a);
    PSF_CASE(4) PSF_CASE(5) PSF_CASE(6) PSF_CASE(7) PSF_CASE(8) PSF_CASE(9) PSF_CASE(10) PSF_CASE(11)
    PSF_CASE(12) PSF_CASE(13) PSF_CASE(14) PSF_CASE(15) PSF_CASE(16) PSF_CASE(17) PSF_CASE(18)
    PSF_CASE(19) PSF_CASE(20)
#undef PSF_CASE
    default:
      return hipErrorInvalidValue;
  }
}
template hipError_t launch_bwd_fused<PSF_TGS>(int L, const BwdWinArgs& a);

namespace {
template <int L, int TGS, int NT>
hipError_t launch_fused_edge(const BwdWinArgs& a) {
  using Cfg = BwdFusedEdgeCfg<L, TGS, NT>;
  auto kern = chord_bwd_fused_edge_k<L, TGS, NT>;
  static std::atomic<int> done{0};
  if (hipError_t e = raise_lds_limit(kern, Cfg::lds_bytes, done); e != hipSuccess) return e;
  hipLaunchKernelGGL(kern, dim3(a.gm.nblocks), dim3(NT), Cfg::lds_bytes, a.stream, a.dZ, a.WV, a.V2, a.out2, a.out, a.gm, a.offs,
                     a.w_total);
  return hipGetLastError();
}
}  // namespace
template <int TGS>
hipError_t launch_bwd_fused_edge(int L, const BwdWinArgs& a) {
  switch (L) {
#define PSF_CASE(LL) \
  case LL:           \
    return launch_fused_edge<LL, TGS, 256>(a);
    PSF_CASE(4) PSF_CASE(5) PSF_CASE(6) PSF_CASE(7) PSF_CASE(8) PSF_CASE(9) PSF_CASE(10) PSF_CASE(11)
    PSF_CASE(12) PSF_CASE(13) PSF_CASE(14) PSF_CASE(15) PSF_CASE(16) PSF_CASE(17) PSF_CASE(18)
    PSF_CASE(19) PSF_CASE(20)
#undef PSF_CASE
    default:
      return hipErrorInvalidValue;
  }
}
template hipError_t launch_bwd_fused_edge<PSF_TGS>(int L, const BwdWinArgs& a);
#endif

}  // namespace psf
