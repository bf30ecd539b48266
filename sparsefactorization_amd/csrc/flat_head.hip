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

constexpr int kChunk = 4096;   // floats of K per workgroup at U = 4: 256 threads x 4 float4 (U = 1: 1024)
constexpr int kRowsPerWg = 8;  // batch rows per workgroup
constexpr int kMaxJ = 8;

// U float4 of K per thread: 4 (the first form) or 1 — with 4096-float chunks Temporal Order's head (K = 131072, B = 40) is 160
// workgroups on 256 CUs and read its 21 MB at 1 TB/s; the launcher takes U = 1 whenever U = 4 would give fewer than 1024.
template <int J, int U>
__global__ void __launch_bounds__(256)
flat_head_partial_k(const float* __restrict__ X, const float* __restrict__ W, int32_t B, int64_t K, float* __restrict__ part) {
  __shared__ float red[4][J];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int64_t k0 = (int64_t)blockIdx.x * (1024 * U);
  float4 w[J][U];
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int64_t k = k0 + 4 * (tid + 256 * u);
#pragma unroll
    for (int j = 0; j < J; ++j)
      w[j][u] = k < K ? *reinterpret_cast<const float4*>(W + (int64_t)j * K + k) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int b0 = blockIdx.y * kRowsPerWg;
  for (int r = 0; r < kRowsPerWg; ++r) {
    const int b = b0 + r;
    if (b >= B) break;  // workgroup-uniform
    float4 x[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t k = k0 + 4 * (tid + 256 * u);
      x[u] = k < K ? *reinterpret_cast<const float4*>(X + (int64_t)b * K + k) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float s[J];
#pragma unroll
    for (int j = 0; j < J; ++j) {
      float a = 0.f;
#pragma unroll
      for (int u = 0; u < U; ++u) {
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

// out[i] = bias + sum over the chunks of part[c][i]: one wave per output; lane l adds chunks l, l + 64, ... in ascending order,
// then the 64 lane sums are combined by a fixed butterfly (one thread per output walking all chunks took 32 us for the 128
// chunks of the genome head).
__global__ void __launch_bounds__(64)
flat_head_reduce_k(const float* __restrict__ part, const float* __restrict__ bias, int32_t chunks, int32_t n, int32_t J,
                   float* __restrict__ out) {
  const int i = blockIdx.x;  // i = b * J + j
  const int lane = threadIdx.x;
  float s = 0.f;
  for (int c = lane; c < chunks; c += 64) s += part[(int64_t)c * n + i];
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) s += __shfl_xor(s, m, 64);
  if (lane == 0) out[i] = s + (bias ? bias[i % J] : 0.f);
}

template <int J>
void launch_partial(const float* X, const float* W, int32_t B, int64_t K, float* part, int chunks, int u, hipStream_t s) {
  const dim3 grid(chunks, (B + kRowsPerWg - 1) / kRowsPerWg);
  if (u == 4) hipLaunchKernelGGL((flat_head_partial_k<J, 4>), grid, dim3(256), 0, s, X, W, B, K, part);
  else hipLaunchKernelGGL((flat_head_partial_k<J, 1>), grid, dim3(256), 0, s, X, W, B, K, part);
}

// chunk size of a launch: 4096 floats, or 1024 when that would leave fewer than 1024 workgroups
int head_unroll(int32_t B, int64_t K) {
  const int64_t wgs4 = ((K + kChunk - 1) / kChunk) * ((B + kRowsPerWg - 1) / kRowsPerWg);
  return wgs4 >= 1024 ? 4 : 1;
}

// ---- backward ----------------------------------------------------------------------------------------------------
//   dW[j, i] = sum_b dY[b, j] * X[b, i]        dX[b, i] = sum_j dY[b, j] * W[j, i]
// As library GEMMs these are [J x B] x [B x K] and [B x J] x [J x K] with K = N * C in the hundreds of thousands and J, B
// tiny: hipBLASLt takes 272 us for the first one at the genome shape (K = 524288, B = 16, J = 2;
// profiles/r03ap_family_step_kernels.log) for what is one read of X. Here a thread owns four consecutive i: it reads its
// float4 of every batch row once, accumulates the J rows of dW in registers (b ascending: a fixed order, no atomics) and
// writes the same row's dX from the J weight float4s it holds. dY ([B, J], <= 4 KB) sits in LDS.
constexpr int kBwdMaxB = 1024;
constexpr int kBwdMaxJ = 16;  // (the forward kernel keeps J weight chunks in registers: J <= 8; the backward 2 J float4)

constexpr int kBwdThreads = 64;  // one wave per workgroup: K / 256 workgroups (512 at K = 131072) keep every CU busy
constexpr int kBwdRows = 8;      // batch rows whose loads are issued together (the sums still run in ascending row order)

template <int J, bool DX, bool DW>
__global__ void __launch_bounds__(kBwdThreads)
flat_head_bwd_k(const float* __restrict__ dY, const float* __restrict__ X, const float* __restrict__ W, float* __restrict__ dX,
                float* __restrict__ dW, int32_t B, int64_t K) {
  extern __shared__ float sdy[];  // [B][J]
  for (int i = threadIdx.x; i < B * J; i += kBwdThreads) sdy[i] = dY[i];
  __syncthreads();
  const int64_t k = 4 * ((int64_t)blockIdx.x * kBwdThreads + threadIdx.x);
  if (k >= K) return;
  float4 w[J], acc[J];
#pragma unroll
  for (int j = 0; j < J; ++j) {
    if (DX) w[j] = *reinterpret_cast<const float4*>(W + (int64_t)j * K + k);
    acc[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (int b0 = 0; b0 < B; b0 += kBwdRows) {
    float4 x[kBwdRows];
    if (DW) {
#pragma unroll
      for (int u = 0; u < kBwdRows; ++u) {
        const int b = b0 + u < B ? b0 + u : B - 1;  // clamped: a finished row is read again, not used
        x[u] = *reinterpret_cast<const float4*>(X + (int64_t)b * K + k);
      }
    }
#pragma unroll
    for (int u = 0; u < kBwdRows; ++u) {
      const int b = b0 + u;
      if (b < B) {
        const float* g = sdy + b * J;
        if (DW) {
#pragma unroll
          for (int j = 0; j < J; ++j) {
            acc[j].x = fmaf(g[j], x[u].x, acc[j].x);
            acc[j].y = fmaf(g[j], x[u].y, acc[j].y);
            acc[j].z = fmaf(g[j], x[u].z, acc[j].z);
            acc[j].w = fmaf(g[j], x[u].w, acc[j].w);
          }
        }
        if (DX) {
          float4 d = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
          for (int j = 0; j < J; ++j) {
            d.x = fmaf(g[j], w[j].x, d.x);
            d.y = fmaf(g[j], w[j].y, d.y);
            d.z = fmaf(g[j], w[j].z, d.z);
            d.w = fmaf(g[j], w[j].w, d.w);
          }
          *reinterpret_cast<float4*>(dX + (int64_t)b * K + k) = d;
        }
      }
    }
  }
  if (DW) {
#pragma unroll
    for (int j = 0; j < J; ++j) *reinterpret_cast<float4*>(dW + (int64_t)j * K + k) = acc[j];
  }
}

template <int J>
void launch_bwd(const float* dY, const float* X, const float* W, float* dX, float* dW, int32_t B, int64_t K, hipStream_t s) {
  const unsigned blocks = (unsigned)((K / 4 + kBwdThreads - 1) / kBwdThreads);
  const size_t lds = (size_t)B * J * sizeof(float);
  if (dX && dW) hipLaunchKernelGGL((flat_head_bwd_k<J, true, true>), dim3(blocks), dim3(kBwdThreads), lds, s, dY, X, W, dX, dW, B, K);
  else if (dX) hipLaunchKernelGGL((flat_head_bwd_k<J, true, false>), dim3(blocks), dim3(kBwdThreads), lds, s, dY, X, W, dX, dW, B, K);
  else hipLaunchKernelGGL((flat_head_bwd_k<J, false, true>), dim3(blocks), dim3(kBwdThreads), lds, s, dY, X, W, dX, dW, B, K);
}

}  // namespace

extern "C" int psf_flat_head_bwd_f32(const float* dY, const float* X, const float* W, float* dX, float* dW, int32_t B,
                                     int64_t K, int32_t J, void* stream) {
  if (!dY || (!dX && !dW)) return psf_internal_fail(PSF_E_NULL, "psf_flat_head_bwd: dY and at least one of dX, dW must be non-NULL");
  if ((dW && !X) || (dX && !W)) return psf_internal_fail(PSF_E_NULL, "psf_flat_head_bwd: dW needs X, dX needs W");
  if (B < 1 || B > kBwdMaxB || K < 4 || (K & 3) || J < 1 || J > kBwdMaxJ || K / 4 / kBwdThreads > 0x7ffffffe)
    return psf_internal_fail(PSF_E_SHAPE, "psf_flat_head_bwd: need 1 <= B <= 1024, K a positive multiple of 4, 1 <= J <= 16");
  const uintptr_t al = (dW ? reinterpret_cast<uintptr_t>(X) | reinterpret_cast<uintptr_t>(dW) : 0) |
                       (dX ? reinterpret_cast<uintptr_t>(W) | reinterpret_cast<uintptr_t>(dX) : 0);
  if (al & 15) return psf_internal_fail(PSF_E_ALIGN, "psf_flat_head_bwd: X, W, dX and dW must be 16-byte aligned");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  switch (J) {
    case 1: launch_bwd<1>(dY, X, W, dX, dW, B, K, s); break;
    case 2: launch_bwd<2>(dY, X, W, dX, dW, B, K, s); break;
    case 3: launch_bwd<3>(dY, X, W, dX, dW, B, K, s); break;
    case 4: launch_bwd<4>(dY, X, W, dX, dW, B, K, s); break;
    case 5: launch_bwd<5>(dY, X, W, dX, dW, B, K, s); break;
    case 6: launch_bwd<6>(dY, X, W, dX, dW, B, K, s); break;
    case 7: launch_bwd<7>(dY, X, W, dX, dW, B, K, s); break;
    case 8: launch_bwd<8>(dY, X, W, dX, dW, B, K, s); break;
    case 9: launch_bwd<9>(dY, X, W, dX, dW, B, K, s); break;
    case 10: launch_bwd<10>(dY, X, W, dX, dW, B, K, s); break;
    case 11: launch_bwd<11>(dY, X, W, dX, dW, B, K, s); break;
    case 12: launch_bwd<12>(dY, X, W, dX, dW, B, K, s); break;
    case 13: launch_bwd<13>(dY, X, W, dX, dW, B, K, s); break;
    case 14: launch_bwd<14>(dY, X, W, dX, dW, B, K, s); break;
    case 15: launch_bwd<15>(dY, X, W, dX, dW, B, K, s); break;
    default: launch_bwd<16>(dY, X, W, dX, dW, B, K, s); break;
  }
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? PSF_OK : psf_internal_fail((int)e, hipGetErrorString(e));
}

extern "C" int64_t psf_flat_head_workspace(int32_t B, int64_t K, int32_t J) {
  if (B < 1 || K < 4 || (K & 3) || J < 1 || J > kMaxJ) return -1;
  return ((K + 1023) / 1024) * (int64_t)B * J * (int64_t)sizeof(float);  // sized for the smaller chunk
}

extern "C" int psf_flat_head_f32(const float* X, const float* W, const float* bias, float* out, int32_t B, int64_t K,
                                 int32_t J, void* workspace, int64_t workspace_bytes, void* stream) {
  if (!X || !W || !out || !workspace) return psf_internal_fail(PSF_E_NULL, "psf_flat_head: X, W, out and workspace must be non-NULL");
  const int64_t need = psf_flat_head_workspace(B, K, J);
  if (need < 0) return psf_internal_fail(PSF_E_SHAPE, "psf_flat_head: need B >= 1, K a positive multiple of 4, 1 <= J <= 8");
  if (workspace_bytes < need) return psf_internal_fail(PSF_E_SHAPE, "psf_flat_head: workspace smaller than psf_flat_head_workspace(B, K, J)");
  if ((reinterpret_cast<uintptr_t>(X) & 15) || (reinterpret_cast<uintptr_t>(W) & 15))
    return psf_internal_fail(PSF_E_ALIGN, "psf_flat_head: X and W must be 16-byte aligned");
  const int u = head_unroll(B, K);
  const int64_t chunks64 = (K + 1024 * u - 1) / (1024 * u);
  if (chunks64 > 0x7fffffff) return psf_internal_fail(PSF_E_SHAPE, "psf_flat_head: K too large");
  const int chunks = (int)chunks64;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  float* part = reinterpret_cast<float*>(workspace);
  switch (J) {
    case 1: launch_partial<1>(X, W, B, K, part, chunks, u, s); break;
    case 2: launch_partial<2>(X, W, B, K, part, chunks, u, s); break;
    case 3: launch_partial<3>(X, W, B, K, part, chunks, u, s); break;
    case 4: launch_partial<4>(X, W, B, K, part, chunks, u, s); break;
    case 5: launch_partial<5>(X, W, B, K, part, chunks, u, s); break;
    case 6: launch_partial<6>(X, W, B, K, part, chunks, u, s); break;
    case 7: launch_partial<7>(X, W, B, K, part, chunks, u, s); break;
    default: launch_partial<8>(X, W, B, K, part, chunks, u, s); break;
  }
  const int n = B * J;
  hipLaunchKernelGGL(flat_head_reduce_k, dim3(n), dim3(64), 0, s, part, bias, chunks, n, J, out);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? PSF_OK : psf_internal_fail((int)e, hipGetErrorString(e));
}
