// fwd_mlp_step_inst.hip — instantiates the forward step that computes its own W tile (fwd_mlp_step.h) for ONE
// channel-group shift. Built once per -DPSF_TGS=0..3 so the instances compile in parallel; see build.py.
#ifndef PSF_TGS
#error "compile with -DPSF_TGS=<0..3>"
#endif

#include <atomic>

#include "fwd_mlp_step.h"

namespace psf {
namespace {

constexpr int kLdsPerCuBytes = 160 * 1024;

template <int L, int TGS, bool RES, bool EDGE>
hipError_t launch_one(const FwdMlpArgs& a) {
  if (a.in.kind != 0) return hipErrorInvalidValue;  // rows of `data` only (recipes: mixer_lds.h)
  using Cfg = MlpStepCfg<L, TGS>;
  auto kern = chord_fwd_mlp_k<L, TGS, RES, EDGE>;
  int lds = Cfg::img_off + a.nu * kImgBytes;
  if (a.wg_per_cu > 0) {  // occupancy limiter: asking for just over 160 KB / (n + 1) caps a CU at n workgroups
    const int floor_bytes = kLdsPerCuBytes / (a.wg_per_cu + 1) + 256;
    if (floor_bytes > lds) lds = floor_bytes;
  }
  if (lds > kLdsPerCuBytes) return hipErrorInvalidValue;
  if (lds > 48 * 1024) {
    static std::atomic<int> done{0};
    if (done.load() < lds) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      if (e != hipSuccess) return e;
      done.store(lds);
    }
  }
  hipLaunchKernelGGL(kern, dim3(a.gm.nblocks), dim3(256), lds, a.stream, a.in, a.V, a.res, a.out, a.images, a.nu, a.E, a.gm,
                     a.offs, a.ablate);
  return hipGetLastError();
}

template <int L, int TGS>
hipError_t launch_flags(const FwdMlpArgs& a) {
  if (a.res != nullptr) return a.edge ? launch_one<L, TGS, true, true>(a) : launch_one<L, TGS, true, false>(a);
  return a.edge ? launch_one<L, TGS, false, true>(a) : launch_one<L, TGS, false, false>(a);
}

}  // namespace

template <int TGS>
hipError_t launch_fwd_mlp(int L, const FwdMlpArgs& a) {
  switch (L) {
#define PSF_CASE(LL) \
  case LL:           \
    return launch_flags<LL, TGS>(a);
    PSF_CASE(4) PSF_CASE(5) PSF_CASE(6) PSF_CASE(7) PSF_CASE(8) PSF_CASE(9) PSF_CASE(10) PSF_CASE(11)
    PSF_CASE(12) PSF_CASE(13) PSF_CASE(14) PSF_CASE(15) PSF_CASE(16) PSF_CASE(17) PSF_CASE(18)
    PSF_CASE(19) PSF_CASE(20)
#undef PSF_CASE
    default:
      return hipErrorInvalidValue;
  }
}

template hipError_t launch_fwd_mlp<PSF_TGS>(int L, const FwdMlpArgs& a);

template <int TGS>
hipError_t launch_mixer_g(const FwdMlpArgs& a) {
  if (a.in.kind != 0) return hipErrorInvalidValue;
  const int lds = a.nu * kImgBytes;
  auto launch = [&](auto kern) {
    if (lds > 48 * 1024) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(kern, dim3(a.gm.nblocks), dim3(256), lds, a.stream, a.in, a.out, a.images, a.nu, a.E, a.gm);
    return hipGetLastError();
  };
  return a.edge ? launch(chord_mixer_g_k<TGS, true>) : launch(chord_mixer_g_k<TGS, false>);
}

template hipError_t launch_mixer_g<PSF_TGS>(const FwdMlpArgs& a);

}  // namespace psf
