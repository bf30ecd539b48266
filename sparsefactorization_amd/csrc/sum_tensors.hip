// sum_tensors.hip — out = ((s_0 + s_1) + s_2) + ... + s_{K-1}, elementwise, one pass.
//
// The residual of the chain, X_{m+1} = W_m X_m + V_0 (SyntheticExperiments/psf.py:187-188), sends every step's output
// gradient to V_0 as well: dV_0 = dX_0 + sum_{m>=1} dX_m. Accumulated step by step in the backward loop that is
// M kernels of 3 x B*N*C*4 bytes each (14 x 9.5 us = 4.5 % of the Temporal-Order training step,
// profiles/r02k_train_step_profile.log); the dX_m exist anyway (each is the next step's dZ), so they are summed once
// at the end: (M+1) reads + 1 write of B*N*C*4 bytes, HBM-bound, fixed left-to-right order (bit-identical to the
// step-by-step accumulation).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/psf_chord.h"

extern "C" int psf_internal_fail(int code, const char* message);

namespace {

constexpr int kMaxSrc = 32;

struct SumArgs {
  const float* src[kMaxSrc];
  float* out;
  int64_t n4;  // float4 elements
  int32_t count;
};

template <int K>
__global__ void __launch_bounds__(256) sum_tensors_k(const SumArgs a) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < a.n4; i += stride) {
    float4 v[K];
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = reinterpret_cast<const float4*>(a.src[k])[i];  // all loads in flight together
    float4 acc = v[0];
#pragma unroll
    for (int k = 1; k < K; ++k) {
      acc.x = __fadd_rn(acc.x, v[k].x);
      acc.y = __fadd_rn(acc.y, v[k].y);
      acc.z = __fadd_rn(acc.z, v[k].z);
      acc.w = __fadd_rn(acc.w, v[k].w);
    }
    reinterpret_cast<float4*>(a.out)[i] = acc;
  }
}

template <int K>
void launch(const SumArgs& a, hipStream_t s) {
  const int64_t blocks_needed = (a.n4 + 255) / 256;
  const unsigned grid = (unsigned)(blocks_needed < 8192 ? blocks_needed : 8192);
  hipLaunchKernelGGL(sum_tensors_k<K>, dim3(grid), dim3(256), 0, s, a);
}

}  // namespace

extern "C" int psf_sum_tensors_f32(const float* const* srcs, int32_t count, int64_t n, float* out, void* stream) {
  if (!srcs || !out) return psf_internal_fail(PSF_E_NULL, "psf_sum_tensors: NULL argument");
  if (count < 1 || count > kMaxSrc || n < 0 || (n & 3))
    return psf_internal_fail(PSF_E_SHAPE, "psf_sum_tensors: need 1 <= count <= 32 and n a multiple of 4");
  if (n == 0) return PSF_OK;
  SumArgs a;
  for (int k = 0; k < kMaxSrc; ++k) a.src[k] = k < count ? srcs[k] : nullptr;
  for (int k = 0; k < count; ++k)
    if (!srcs[k] || (reinterpret_cast<uintptr_t>(srcs[k]) & 15))
      return psf_internal_fail(PSF_E_ALIGN, "psf_sum_tensors: sources must be non-NULL and 16-byte aligned");
  if (reinterpret_cast<uintptr_t>(out) & 15) return psf_internal_fail(PSF_E_ALIGN, "psf_sum_tensors: out must be 16-byte aligned");
  a.out = out;
  a.n4 = n / 4;
  a.count = count;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  // up to 16 sources per pass (64 registers of loads in flight; the 15 terms of a 14-step chain are one pass: 72 -> 50 us
  // at Temporal Order N=16384, B=40); later passes take `out` as their first source (same left-to-right order)
  int done = 0;
  while (done < count) {
    const bool first = done == 0;
    int take = count - done;
    const int cap = first ? 16 : 15;
    if (take > cap) take = cap;
    SumArgs p = a;
    int k = 0;
    if (!first) p.src[k++] = out;
    for (int i = 0; i < take; ++i) p.src[k++] = srcs[done + i];
    switch (k) {
#define PSF_SUM_CASE(K) case K: launch<K>(p, s); break;
      PSF_SUM_CASE(1) PSF_SUM_CASE(2) PSF_SUM_CASE(3) PSF_SUM_CASE(4) PSF_SUM_CASE(5) PSF_SUM_CASE(6) PSF_SUM_CASE(7)
      PSF_SUM_CASE(8) PSF_SUM_CASE(9) PSF_SUM_CASE(10) PSF_SUM_CASE(11) PSF_SUM_CASE(12) PSF_SUM_CASE(13) PSF_SUM_CASE(14)
      PSF_SUM_CASE(15)
#undef PSF_SUM_CASE
      default: launch<16>(p, s); break;
    }
    done += take;
  }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? PSF_OK : psf_internal_fail((int)e, hipGetErrorString(e));
}
