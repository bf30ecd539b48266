// bwd_chain_lds_inst.hip — instances and launcher of the LDS-resident backward chain (bwd_chain_lds.h).
#include <atomic>

#include "bwd_chain_lds.h"

namespace psf {
namespace {

template <int L, int G, bool RES>
hipError_t launch_one(const ChainBwdArgs& a, const Offsets& offs, int B, hipStream_t s) {
  auto kern = chord_chain_bwd_lds_k<L, G, RES>;
  const int lds_bytes = 2 * G * kChainBwdRows * 16 + a.N * chain_bwd_wstride<L>() * 4;
  static std::atomic<int> raised{0};
  if (lds_bytes > 48 * 1024 && raised.load() < lds_bytes) {
    const int cap = 2 * G * kChainBwdRows * 16 + kChainBwdRows * chain_bwd_wstride<L>() * 4;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, cap);
    if (e != hipSuccess) return e;
    raised.store(cap);
  }
  const int threads = (a.N + 63) / 64 * 64;
  hipLaunchKernelGGL(kern, dim3((unsigned)B), dim3(threads), lds_bytes, s, a, offs);
  return hipGetLastError();
}

template <int L>
hipError_t launch_G(int G, bool res, const ChainBwdArgs& a, const Offsets& offs, int B, hipStream_t s) {
  if (G == 1) return res ? launch_one<L, 1, true>(a, offs, B, s) : launch_one<L, 1, false>(a, offs, B, s);
  if (G == 2) return res ? launch_one<L, 2, true>(a, offs, B, s) : launch_one<L, 2, false>(a, offs, B, s);
  return hipErrorInvalidValue;
}

}  // namespace

bool chain_bwd_lds_fits(int64_t N, int64_t C, int32_t L, int32_t M) {
  return N >= 1 && N <= kChainBwdRows && (C == 4 || C == 8) && L >= 2 && L <= 20 && M >= 1 && M <= kChainMaxSteps;
}

hipError_t launch_chain_bwd_lds(int L, int G, bool res, const ChainBwdArgs& a, const Offsets& offs, int B, hipStream_t s) {
  switch (L) {
#define PSF_CASE(LL) \
  case LL:           \
    return launch_G<LL>(G, res, a, offs, B, s);
    PSF_CASE(2) PSF_CASE(3) PSF_CASE(4) PSF_CASE(5) PSF_CASE(6) PSF_CASE(7) PSF_CASE(8) PSF_CASE(9) PSF_CASE(10)
    PSF_CASE(11) PSF_CASE(12) PSF_CASE(13) PSF_CASE(14) PSF_CASE(15) PSF_CASE(16) PSF_CASE(17) PSF_CASE(18)
    PSF_CASE(19) PSF_CASE(20)
#undef PSF_CASE
    default:
      return hipErrorInvalidValue;
  }
}

}  // namespace psf
