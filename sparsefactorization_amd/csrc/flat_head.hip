// flat_head.hip — the FLATTEN head of PSFNet: out[b, j] = bias[j] + sum_i X[b, i] * W[j, i]
//
// `final = nn.Linear(n_vec * n_channels_V, n_class)` applied to V.view(B, -1) (SyntheticExperiments/psf.py:129-134,
// 189-190): [B <= 64, K = 131072] x [K, J <= 4]. As a GEMM the library takes 117 us (Temporal Order, J = 4) to 0.57 ms
// (Adding, J = 1) for what is one read of X (21-34 MB) and J rows of W; `addmv` per class is 25-40 us each.
// Here a workgroup owns a 4096-element chunk of K and a group of 8 batch rows: the chunk of the J weight rows stays
// in registers, each batch row's chunk is read once as 16-byte loads and dotted with all J rows, lane sums are
// combined by wave shuffles and across the four waves through LDS in a fixed order, and the per-chunk partial sums
// are added in chunk order by a second tiny kernel (no atomics: bit-reproducible). HBM-bound on X.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/psf_chord.h"

extern "C" int psf_internal_fail(int code, const char* message);

namespace {

constexpr int kChunk = 4096;   // floats of K per workgroup: 256 threads x 4 float4
constexpr int kRowsPerWg = 8;  // batch rows per workgroup
constexpr int kMaxJ = 8;

template <int J>
__global__ void __launch_bounds__(256)
flat_head_partial_k(const float* __restrict__ X, const float* __restrict__ W, int32_t B, int64_t K, float* __restrict__ part) {
  __shared__ float red[4][J];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int64_t k0 = (int64_t)blockIdx.x * kChunk;
  float4 w[J][4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int64_t k = k0 + 4 * (tid + 256 * u);
#pragma unroll
    for (int j = 0; j < J; ++j)
      w[j][u] = k < K ? *reinterpret_cast<const float4*>(W + (int64_t)j * K + k) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int b0 = blockIdx.y * kRowsPerWg;
  for (int r = 0; r < kRowsPerWg; ++r) {
    const int b = b0 + r;
    if (b >= B) break;  // workgroup-uniform
    float4 x[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t k = k0 + 4 * (tid + 256 * u);
      x[u] = k < K ? *reinterpret_cast<const float4*>(X + (int64_t)b * K + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float s[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
      float a = 0.f;
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        a = fmaf(x[u].x, w[j][u].x, a);
        a = fmaf(x[u].y, w[j][u].y, a);
        a = fmaf(x[u].z, w[j][u].z, a);
        a = fmaf(x[u].w, w[j][u].w, a);
      }
#pragma unroll
      for (int m = 32; m > 0; m >>= 1) a += __shfl_xor(a, m, 64);
      s[j] = a;
    }
    __syncthreads();  // the previous row's combine has read `red`
    if (lane == 0) {
#pragma unroll
      for (int j = 0; j < J; ++j) red[wv][j] = s[j];
    }
    __syncthreads();
    if (tid < J) part[((int64_t)blockIdx.x * B + b) * J + tid] = ((red[0][tid] + red[1][tid]) + red[2][tid]) + red[3][tid];
  }
}

__global__ void __launch_bounds__(256)
flat_head_reduce_k(const float* __restrict__ part, const float* __restrict__ bias, int32_t chunks, int32_t n, int32_t J,
                   float* __restrict__ out) {
  const int i = blockIdx.x * 256 + threadIdx.x;  // i = b * J + j
  if (i >= n) return;
  float s = bias ? bias[i % J] : 0.f;
  for (int c = 0; c < chunks; ++c) s += part[(int64_t)c * n + i];
  out[i] = s;
}

template <int J>
void launch_partial(const float* X, const float* W, int32_t B, int64_t K, float* part, int chunks, hipStream_t s) {
  hipLaunchKernelGGL((flat_head_partial_k<J>), dim3(chunks, (B + kRowsPerWg - 1) / kRowsPerWg), dim3(256), 0, s, X, W, B, K, part);
}

}  // namespace

extern "C" int64_t psf_flat_head_workspace(int32_t B, int64_t K, int32_t J) {
  if (B < 1 || K < 4 || (K & 3) || J < 1 || J > kMaxJ) return -1;
  return ((K + kChunk - 1) / kChunk) * (int64_t)B * J * (int64_t)sizeof(float);
}

extern "C" int psf_flat_head_f32(const float* X, const float* W, const float* bias, float* out, int32_t B, int64_t K,
                                 int32_t J, void* workspace, int64_t workspace_bytes, void* stream) {
  if (!X || !W || !out || !workspace) return psf_internal_fail(PSF_E_NULL, "psf_flat_head: X, W, out and workspace must be non-NULL");
  const int64_t need = psf_flat_head_workspace(B, K, J);
  if (need < 0) return psf_internal_fail(PSF_E_SHAPE, "psf_flat_head: need B >= 1, K a positive multiple of 4, 1 <= J <= 8");
  if (workspace_bytes < need) return psf_internal_fail(PSF_E_SHAPE, "psf_flat_head: workspace smaller than psf_flat_head_workspace(B, K, J)");
  if ((reinterpret_cast<uintptr_t>(X) & 15) || (reinterpret_cast<uintptr_t>(W) & 15))
    return psf_internal_fail(PSF_E_ALIGN, "psf_flat_head: X and W must be 16-byte aligned");
  const int64_t chunks64 = (K + kChunk - 1) / kChunk;
  if (chunks64 > 0x7fffffff) return psf_internal_fail(PSF_E_SHAPE, "psf_flat_head: K too large");
  const int chunks = (int)chunks64;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  float* part = reinterpret_cast<float*>(workspace);
  switch (J) {
    case 1: launch_partial<1>(X, W, B, K, part, chunks, s); break;
    case 2: launch_partial<2>(X, W, B, K, part, chunks, s); break;
    case 3: launch_partial<3>(X, W, B, K, part, chunks, s); break;
    case 4: launch_partial<4>(X, W, B, K, part, chunks, s); break;
    case 5: launch_partial<5>(X, W, B, K, part, chunks, s); break;
    case 6: launch_partial<6>(X, W, B, K, part, chunks, s); break;
    case 7: launch_partial<7>(X, W, B, K, part, chunks, s); break;
    default: launch_partial<8>(X, W, B, K, part, chunks, s); break;
  }
  const int n = B * J;
  hipLaunchKernelGGL(flat_head_reduce_k, dim3((n + 255) / 256), dim3(256), 0, s, part, bias, chunks, n, J, out);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? PSF_OK : psf_internal_fail((int)e, hipGetErrorString(e));
}
