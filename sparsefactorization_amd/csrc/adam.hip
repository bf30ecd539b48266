// adam.hip — Adam step over a list of small tensors, one launch per <= 40 tensors.
//
// The training loop of the reference ends in optim.Adam(net.parameters(), lr).step() (SyntheticExperiments/
// psf_training.py:50-53, psf_utils.py:71). A PSFNet has ~66 parameters, two of them (pos_embedding, final: 524 288
// floats each at N = 16384) holding 98 % of the elements; PyTorch's fused multi-tensor Adam hands each block a 65 536-
// element chunk, so those two tensors run on 16 workgroups: 2 x 43 us per step (3 % of the Temporal-Order training step,
// profiles/r02p_train_step_kernels.log) for 30 MB of traffic. Here a block takes 4096 elements: ~270 workgroups, HBM-bound.
//
// Same update as torch.optim.Adam (no weight decay, no amsgrad, not maximize):
//   m += (g - m) (1 - beta1);  v = beta2 v + (1 - beta2) g g;  p -= (lr / (1 - beta1^t)) * m / (sqrt(v) / sqrt(1 - beta2^t) + eps)
// The step count t comes from the host, or from a device scalar (capturable: a HIP-graph replay must see it advance).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/psf_chord.h"

extern "C" int psf_internal_fail(int code, const char* message);

namespace {

constexpr int kAdamMax = 40;
constexpr int kAdamChunk = 4096;  // elements per workgroup

struct AdamArgs {
  float* p[kAdamMax];
  const float* g[kAdamMax];
  float* m[kAdamMax];
  float* v[kAdamMax];
  int64_t n[kAdamMax];
  int32_t start[kAdamMax + 1];  // first workgroup of tensor t; start[count] = grid size
  int32_t vec4[kAdamMax];       // 1: all four arrays 16-byte aligned and n % 4 == 0
  int32_t count;
  float lr, beta1, beta2, eps, step_host;
  const float* step_dev;
};

__device__ __forceinline__ void adam1(float& p, float g, float& m, float& v, float one_m_b1, float b2, float one_m_b2,
                                      float step_size, float bc2_sqrt, float eps) {
  m = m + (g - m) * one_m_b1;
  v = b2 * v + one_m_b2 * g * g;
  const float denom = sqrtf(v) / bc2_sqrt + eps;
  p = p - step_size * (m / denom);
}

__global__ void __launch_bounds__(256) adam_k(const AdamArgs a) {
  int t = 0;
  while (t + 1 < a.count && (int)blockIdx.x >= a.start[t + 1]) ++t;  // <= 40 entries, block-uniform
  const int64_t chunk = (int64_t)((int)blockIdx.x - a.start[t]);
  const float step = a.step_dev ? *a.step_dev : a.step_host;
  const float bc1 = 1.0f - powf(a.beta1, step), bc2 = 1.0f - powf(a.beta2, step);
  const float step_size = a.lr / bc1, bc2_sqrt = sqrtf(bc2);
  const float one_m_b1 = 1.0f - a.beta1, one_m_b2 = 1.0f - a.beta2;
  float* __restrict__ p = a.p[t];
  const float* __restrict__ g = a.g[t];
  float* __restrict__ m = a.m[t];
  float* __restrict__ v = a.v[t];
  const int64_t lo = chunk * kAdamChunk, n = a.n[t];
  const int64_t hi = lo + kAdamChunk < n ? lo + kAdamChunk : n;
  if (a.vec4[t]) {
    for (int64_t i = lo / 4 + threadIdx.x; i < hi / 4; i += 256) {
      float4 pp = reinterpret_cast<float4*>(p)[i], mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
      const float4 gg = reinterpret_cast<const float4*>(g)[i];
      adam1(pp.x, gg.x, mm.x, vv.x, one_m_b1, a.beta2, one_m_b2, step_size, bc2_sqrt, a.eps);
      adam1(pp.y, gg.y, mm.y, vv.y, one_m_b1, a.beta2, one_m_b2, step_size, bc2_sqrt, a.eps);
      adam1(pp.z, gg.z, mm.z, vv.z, one_m_b1, a.beta2, one_m_b2, step_size, bc2_sqrt, a.eps);
      adam1(pp.w, gg.w, mm.w, vv.w, one_m_b1, a.beta2, one_m_b2, step_size, bc2_sqrt, a.eps);
      reinterpret_cast<float4*>(p)[i] = pp;
      reinterpret_cast<float4*>(m)[i] = mm;
      reinterpret_cast<float4*>(v)[i] = vv;
    }
  } else {
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
      float pp = p[i], mm = m[i], vv = v[i];
      adam1(pp, g[i], mm, vv, one_m_b1, a.beta2, one_m_b2, step_size, bc2_sqrt, a.eps);
      p[i] = pp;
      m[i] = mm;
      v[i] = vv;
    }
  }
}

}  // namespace

extern "C" int psf_adam_step_f32(float* const* params, const float* const* grads, float* const* exp_avg,
                                 float* const* exp_avg_sq, const int64_t* numels, int32_t count, float lr, float beta1,
                                 float beta2, float eps, float step, const float* step_dev, void* stream) {
  if (count < 0) return psf_internal_fail(PSF_E_SHAPE, "psf_adam_step: count < 0");
  if (count == 0) return PSF_OK;
  if (!params || !grads || !exp_avg || !exp_avg_sq || !numels) return psf_internal_fail(PSF_E_NULL, "psf_adam_step: NULL table");
  if (!step_dev && !(step >= 1.0f)) return psf_internal_fail(PSF_E_SHAPE, "psf_adam_step: step must be >= 1");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  for (int base = 0; base < count; base += kAdamMax) {
    const int take = count - base < kAdamMax ? count - base : kAdamMax;
    AdamArgs a;
    int64_t blocks = 0;
    for (int i = 0; i < kAdamMax; ++i) {
      a.p[i] = nullptr; a.g[i] = nullptr; a.m[i] = nullptr; a.v[i] = nullptr; a.n[i] = 0; a.vec4[i] = 0; a.start[i] = 0;
    }
    for (int i = 0; i < take; ++i) {
      const int k = base + i;
      if (!params[k] || !grads[k] || !exp_avg[k] || !exp_avg_sq[k] || numels[k] < 0)
        return psf_internal_fail(PSF_E_NULL, "psf_adam_step: NULL tensor pointer or negative size");
      a.p[i] = params[k]; a.g[i] = grads[k]; a.m[i] = exp_avg[k]; a.v[i] = exp_avg_sq[k]; a.n[i] = numels[k];
      const uintptr_t bits = reinterpret_cast<uintptr_t>(params[k]) | reinterpret_cast<uintptr_t>(grads[k]) |
                             reinterpret_cast<uintptr_t>(exp_avg[k]) | reinterpret_cast<uintptr_t>(exp_avg_sq[k]);
      if (bits & 3) return psf_internal_fail(PSF_E_ALIGN, "psf_adam_step: pointers must be 4-byte aligned");
      a.vec4[i] = ((bits & 15) == 0 && (numels[k] & 3) == 0) ? 1 : 0;
      a.start[i] = (int32_t)blocks;
      blocks += (numels[k] + kAdamChunk - 1) / kAdamChunk;
      if (blocks > 0x7fffffff) return psf_internal_fail(PSF_E_SHAPE, "psf_adam_step: too many elements");
    }
    for (int i = take; i <= kAdamMax; ++i) a.start[i] = (int32_t)blocks;
    a.count = take;
    a.lr = lr; a.beta1 = beta1; a.beta2 = beta2; a.eps = eps; a.step_host = step; a.step_dev = step_dev;
    if (blocks > 0) hipLaunchKernelGGL(adam_k, dim3((unsigned)blocks), dim3(256), 0, s, a);
  }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? PSF_OK : psf_internal_fail((int)e, hipGetErrorString(e));
}
