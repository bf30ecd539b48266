// stream_mix.hip — a plain streaming kernel with the byte mix of the forward chord step, for bench.py's second denominator.
//
// Per output vector (16 bytes) it reads two vectors of a W-like stream, one of a V-like and one of a residual-like stream:
// 32 + 16 + 16 bytes read, 16 written — W : V : res : out = 2 : 1 : 1 : 1, the step kernel's 15 : 8 : 8 : 8 at cfg2 (L = 15,
// C = 8). No gather, no LDS, no reuse: whatever rate this sustains on a box is what the memory system gives that mix, with the
// operands wherever the caller put them (all beyond the Infinity Cache, or — like the chain itself — W from HBM and the three
// V-sized streams re-used from one step to the next). bench.py reports the chain kernel's rate against both.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/psf_chord.h"

extern "C" int psf_internal_fail(int code, const char* message);

namespace {

__global__ void __launch_bounds__(256) stream_mix_k(const float4* __restrict__ w, const float4* __restrict__ v,
                                                    const float4* __restrict__ r, float4* __restrict__ o, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256;
  for (; i < n; i += stride) {
    const float4 w0 = w[2 * i], w1 = w[2 * i + 1], vv = v[i], rr = r[i];
    float4 x;
    x.x = w0.x * vv.x + w1.x + rr.x;
    x.y = w0.y * vv.y + w1.y + rr.y;
    x.z = w0.z * vv.z + w1.z + rr.z;
    x.w = w0.w * vv.w + w1.w + rr.w;
    o[i] = x;
  }
}

// the backward step's mix: reads W (2), V, dZ; writes dW (2), dV
__global__ void __launch_bounds__(256) stream_mix_bwd_k(const float4* __restrict__ w, const float4* __restrict__ v,
                                                        const float4* __restrict__ z, float4* __restrict__ dw,
                                                        float4* __restrict__ dv, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256;
  for (; i < n; i += stride) {
    const float4 w0 = w[2 * i], w1 = w[2 * i + 1], vv = v[i], zz = z[i];
    float4 a, b, c;
    a.x = zz.x * vv.x, a.y = zz.y * vv.y, a.z = zz.z * vv.z, a.w = zz.w * vv.w;
    b.x = zz.x + vv.x, b.y = zz.y + vv.y, b.z = zz.z + vv.z, b.w = zz.w + vv.w;
    c.x = w0.x * zz.x + w1.x, c.y = w0.y * zz.y + w1.y, c.z = w0.z * zz.z + w1.z, c.w = w0.w * zz.w + w1.w;
    dw[2 * i] = a;
    dw[2 * i + 1] = b;
    dv[i] = c;
  }
}

}  // namespace

extern "C" int psf_stream_mix_bwd_f32(const float* w, const float* v, const float* z, float* dw, float* dv, int64_t n_vec4,
                                      void* stream) {
  if (!w || !v || !z || !dw || !dv) return psf_internal_fail(PSF_E_NULL, "psf_stream_mix_bwd: NULL argument");
  if (n_vec4 < 0) return psf_internal_fail(PSF_E_SHAPE, "psf_stream_mix_bwd: n_vec4 < 0");
  if (((reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(z) |
        reinterpret_cast<uintptr_t>(dw) | reinterpret_cast<uintptr_t>(dv)) & 15) != 0)
    return psf_internal_fail(PSF_E_ALIGN, "psf_stream_mix_bwd: pointers must be 16-byte aligned");
  if (n_vec4 == 0) return PSF_OK;
  const int64_t blocks_needed = (n_vec4 + 255) / 256;
  const int grid = (int)(blocks_needed < 8192 ? blocks_needed : 8192);
  hipLaunchKernelGGL(stream_mix_bwd_k, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const float4*>(w), reinterpret_cast<const float4*>(v),
                     reinterpret_cast<const float4*>(z), reinterpret_cast<float4*>(dw), reinterpret_cast<float4*>(dv),
                     (size_t)n_vec4);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? PSF_OK : psf_internal_fail((int)e, hipGetErrorString(e));
}

extern "C" int psf_stream_mix_f32(const float* w, const float* v, const float* r, float* out, int64_t n_vec4, void* stream) {
  if (!w || !v || !r || !out) return psf_internal_fail(PSF_E_NULL, "psf_stream_mix: NULL argument");
  if (n_vec4 < 0) return psf_internal_fail(PSF_E_SHAPE, "psf_stream_mix: n_vec4 < 0");
  if (((reinterpret_cast<uintptr_t>(w) | reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(r) |
        reinterpret_cast<uintptr_t>(out)) & 15) != 0)
    return psf_internal_fail(PSF_E_ALIGN, "psf_stream_mix: pointers must be 16-byte aligned");
  if (n_vec4 == 0) return PSF_OK;
  const int64_t blocks_needed = (n_vec4 + 255) / 256;
  const int grid = (int)(blocks_needed < 8192 ? blocks_needed : 8192);  // r01_membench.log: 8192 x 256 is the fastest grid
  hipLaunchKernelGGL(stream_mix_k, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream),
                     reinterpret_cast<const float4*>(w), reinterpret_cast<const float4*>(v),
                     reinterpret_cast<const float4*>(r), reinterpret_cast<float4*>(out), (size_t)n_vec4);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? PSF_OK : psf_internal_fail((int)e, hipGetErrorString(e));
}
