// fwd_chain_lds_inst.hip — instances and launcher of the LDS-resident fused chain kernel (fwd_chain_lds.h).
#include <atomic>

#include "fwd_chain_lds.h"
#include "fwd_chain_lds_launch.h"

namespace psf {
namespace {

template <int L, int CC, int R, bool RES, int NTMAX>
hipError_t launch_one(const ChainArgs& a, const Offsets& offs, int B, int threads, int lds_bytes, hipStream_t s) {
  auto kern = chord_chain_lds_k<L, CC, R, RES, NTMAX>;
  static std::atomic<int> raised{0};
  if (lds_bytes > 48 * 1024 && !raised.load()) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       kChainLdsMaxBytes);
    if (e != hipSuccess) return e;
    raised.store(1);
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(B * a.chunks)), dim3(threads), lds_bytes, s, a, offs);
  return hipGetLastError();
}

template <int L, int G, int R, int CAP, bool RES>
hipError_t launch_rows(const ChainArgs& a, const Offsets& offs, int B, int threads, int lds_bytes, hipStream_t s) {
  auto kern = chord_chain_rows_k<L, G, R, CAP, RES>;
  static std::atomic<int> raised{0};
  if (!raised.load()) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * G * CAP * 16);
    if (e != hipSuccess) return e;
    raised.store(1);
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)(B * a.chunks)), dim3(threads), lds_bytes, s, a, offs);
  return hipGetLastError();
}

template <int L, int CC, bool RES>
hipError_t launch_R(const ChainLdsPlan& p, const ChainArgs& a, const Offsets& offs, int B, hipStream_t s) {
  if constexpr (CC == 2) {
    if (p.big == 1) return launch_rows<L, 2, 2, kChainBigRows, RES>(a, offs, B, p.threads, p.lds_bytes, s);
  } else {
    if (p.big == 2) return launch_rows<L, 1, kChainLongRowsPerThread, kChainLongRows, RES>(a, offs, B, p.threads, p.lds_bytes, s);
  }
  if (p.big) return hipErrorInvalidValue;
  if (p.rows == 1) return launch_one<L, CC, 1, RES, 512>(a, offs, B, p.threads, p.lds_bytes, s);
  if (p.rows == 2 && p.threads <= 512) return launch_one<L, CC, 2, RES, 512>(a, offs, B, p.threads, p.lds_bytes, s);
  if (p.rows == 2) return launch_one<L, CC, 2, RES, 1024>(a, offs, B, p.threads, p.lds_bytes, s);
  if (p.rows == 3 && p.threads <= 768) return launch_one<L, CC, 3, RES, 768>(a, offs, B, p.threads, p.lds_bytes, s);
  return hipErrorInvalidValue;
}

template <int CC>
hipError_t launch_L(int L, bool res, const ChainLdsPlan& p, const ChainArgs& a, const Offsets& offs, int B, hipStream_t s) {
  switch (L) {
#define PSF_CASE(LL) \
  case LL:           \
    return res ? launch_R<LL, CC, true>(p, a, offs, B, s) : launch_R<LL, CC, false>(p, a, offs, B, s);
    PSF_CASE(2) PSF_CASE(3) PSF_CASE(4) PSF_CASE(5) PSF_CASE(6) PSF_CASE(7) PSF_CASE(8) PSF_CASE(9) PSF_CASE(10)
    PSF_CASE(11) PSF_CASE(12) PSF_CASE(13) PSF_CASE(14) PSF_CASE(15) PSF_CASE(16) PSF_CASE(17) PSF_CASE(18)
    PSF_CASE(19) PSF_CASE(20)
#undef PSF_CASE
    default:
      return hipErrorInvalidValue;
  }
}

}  // namespace

bool plan_chain_lds(int64_t N, int64_t C, int32_t L, int32_t M, ChainLdsPlan* p, int cc_pref, int64_t B) {
  if (L < kChainLdsLmin || L > kChainLdsLmax || M < 1 || M > kChainMaxSteps || C % 4 != 0 || N < 1) return false;
  const int64_t CG = C / 4;
  int cc = CG >= 2 ? 2 : 1;
  if (cc_pref == 1) cc = 1;
  p->big = 0;
  if (cc == 2 && N * 2 > kChainLdsMaxSlots && N * 2 <= kChainBigSlots &&
      (cc_pref == 2 || (cc_pref == 0 && B * ((CG + 1) / 2) >= kChainBigMinWgs))) {
    p->cc = 2;
    p->rows = 2;  // a thread owns both channel groups of its two rows
    p->threads = (int)(((N + 1) / 2 + 63) / 64 * 64);
    p->chunks = (int)((CG + 1) / 2);
    p->lds_bytes = kChainBigBytes;
    p->big = 1;
    return true;
  }
  if (N > kChainLdsMaxSlots && N <= kChainLongRows && (cc_pref == 2 || (cc_pref == 0 && B * CG >= kChainBigMinWgs))) {
    p->cc = 1;
    p->rows = kChainLongRowsPerThread;
    p->threads = (int)(((N + p->rows - 1) / p->rows + 63) / 64 * 64);
    p->chunks = (int)CG;
    p->lds_bytes = kChainLongBytes;
    p->big = 2;
    return true;
  }
  if (N * cc > kChainLdsMaxSlots) cc = 1;
  if (N * cc > kChainLdsMaxSlots) return false;
  const int64_t slots = N * cc;                 // (row, channel group) pairs a workgroup owns
  const int R = slots <= 256 ? 1 : (slots <= kChainLdsSlots2 ? 2 : 3);  // 2 rows per thread: <= 512 threads up to 1024 slots
  if (R == 3 && L > 14) return false;           // 3 rows x 2 W rows of L floats: beyond L = 14 the 768-thread instance spills
  int64_t threads = (slots + R - 1) / R;
  threads = (threads + 63) / 64 * 64;           // whole waves; threads / cc row slots cover ceil(N / R) rows
  if (threads > 1024) return false;
  p->cc = cc;
  p->rows = R;
  p->threads = (int)threads;
  p->chunks = (int)((CG + cc - 1) / cc);
  p->lds_bytes = (int)(2 * slots * 16);
  return p->lds_bytes <= kChainLdsMaxBytes;
}

hipError_t launch_chain_lds(const ChainLdsPlan& p, int L, bool res, const ChainArgs& a, const Offsets& offs, int B,
                            hipStream_t s) {
  if (p.cc == 2) return launch_L<2>(L, res, p, a, offs, B, s);
  return launch_L<1>(L, res, p, a, offs, B, s);
}

}  // namespace psf
