// fwd_window_inst.hip — instantiates the LDS-window forward kernels for ONE (channel-group shift, workgroup
// size) pair. Built once per compiled pair (-DPSF_TGS=0..6 [-DPSF_NT=1024]) so the instances compile in
// parallel; see build.py.
#ifndef PSF_TGS
#error "compile with -DPSF_TGS=<0..6>"
#endif
#ifndef PSF_NT
#define PSF_NT 256
#endif

#include <atomic>

#include "fwd_window.h"
#include "fwd_window_launch.h"

namespace psf {
namespace {

template <int L, int TGS, int R, int NT, bool RES, int MODE>
hipError_t launch_one(const FwdWinArgs& a) {
  using Cfg = FwdWinCfg<float, L, TGS, R, NT>;
  auto kern = chord_fwd_win_k<float, L, TGS, R, NT, /*DMA=*/true, RES, MODE>;
  // Occupancy limiter: a CU takes floor(160 KB / LDS per workgroup) workgroups, so asking for just over
  // 160 KB / (n + 1) caps it at n. (cfg2: 3 per CU is 2-3 % faster than the 4 the registers allow — fewer
  // windows competing for the XCD's L2; 2 per CU is 10 % slower. DESIGN.md §4.1.)
  int lds = Cfg::lds_bytes;
  if (a.wg_per_cu > 0) {
    const int floor_bytes = kLdsPerCu / (a.wg_per_cu + 1) + 256;
    if (floor_bytes > lds && floor_bytes <= 64 * 1024) lds = floor_bytes;
  }
  if (lds > 48 * 1024) {
    static std::atomic<int> done{0};
    if (done.load() < lds) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      if (e != hipSuccess) return e;
      done.store(lds);
    }
  }
  hipLaunchKernelGGL(kern, dim3(a.gm.nblocks), dim3(NT), lds, a.stream, a.W, a.V, a.res, a.out, a.gm,
                     a.offs, a.w_total);
  return hipGetLastError();
}

template <int L, int TGS, int R, int NT>
hipError_t launch_flags(const FwdWinArgs& a) {
  if (a.res != nullptr)
    return a.edge ? launch_one<L, TGS, R, NT, true, 1>(a)
                  : (a.gm.aligned ? launch_one<L, TGS, R, NT, true, 2>(a) : launch_one<L, TGS, R, NT, true, 0>(a));
  return a.edge ? launch_one<L, TGS, R, NT, false, 1>(a)
                : (a.gm.aligned ? launch_one<L, TGS, R, NT, false, 2>(a) : launch_one<L, TGS, R, NT, false, 0>(a));
}

template <int TGS, int R, int NT>
hipError_t launch_L(int L, const FwdWinArgs& a) {
  switch (L) {
#define PSF_CASE(LL) \
  case LL:           \
    return launch_flags<LL, TGS, R, NT>(a);
    PSF_CASE(4) PSF_CASE(5) PSF_CASE(6) PSF_CASE(7) PSF_CASE(8) PSF_CASE(9) PSF_CASE(10) PSF_CASE(11)
    PSF_CASE(12) PSF_CASE(13) PSF_CASE(14) PSF_CASE(15) PSF_CASE(16) PSF_CASE(17) PSF_CASE(18)
    PSF_CASE(19) PSF_CASE(20)
#undef PSF_CASE
    default:
      return hipErrorInvalidValue;
  }
}

}  // namespace

template <int TGS, int NT>
hipError_t launch_fwd_win(int rows, int L, const FwdWinArgs& a) {
  static_assert(kWinLmin == 4 && kWinLmax == 20, "keep the PSF_CASE list in step with kWinLmin/kWinLmax");
  static_assert(win_pair_compiled(TGS, NT), "not a compiled (TGS, NT) pair");
  if (rows == 2) return launch_L<TGS, 2, NT>(L, a);  // (the only compiled rows per thread: fwd_window_launch.h)
  if constexpr (win_rows4_compiled(TGS, NT)) {
    if (rows == 4) return launch_L<TGS, 4, NT>(L, a);
  }
  return hipErrorInvalidValue;
}

template hipError_t launch_fwd_win<PSF_TGS, PSF_NT>(int rows, int L, const FwdWinArgs& a);

}  // namespace psf
