// embed.hip — token embedding lookup fused with the positional-embedding add.
//
//   out[t, :] = table[idx[t], :] (+ pos[t mod N, :])            t < T = B*N tokens
//
// PSFNet's first two lines (SyntheticExperiments/psf.py:152-163 `embedding(data)` ... `+ pos_embedding(positions)`,
// LRA/psf.py:203-214): through PyTorch a gather kernel (142 us for the 84 MB of Temporal Order at N = 16384,
// B = 64: a 6-row table looked up a million times) followed by a broadcast add that re-reads and re-writes the
// same 84 MB. Here one pass: 16 bytes per thread, the index and table row come from cache, the output is written
// once. HBM-bound on the output (4*E bytes per token).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/psf_chord.h"

extern "C" int psf_internal_fail(int code, const char* message);

namespace {

__global__ void __launch_bounds__(256)
embed_tokens_k(const int64_t* __restrict__ idx, const float* __restrict__ table, const float* __restrict__ pos,
               float* __restrict__ out, int64_t T, int64_t N, int32_t V, int32_t E4) {
  const int64_t total = T * E4;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t t = i / E4;
    const int e4 = (int)(i - t * E4);
    int64_t v = idx[t];
    v = v < 0 ? 0 : (v >= V ? V - 1 : v);  // out-of-range ids are the caller's error (checked on the host side of the
                                           // Python wrapper in debug runs); never read outside the table
    float4 r = reinterpret_cast<const float4*>(table)[v * E4 + e4];
    if (pos) {
      const float4 p = reinterpret_cast<const float4*>(pos)[(t % N) * E4 + e4];
      r.x += p.x;
      r.y += p.y;
      r.z += p.z;
      r.w += p.w;
    }
    reinterpret_cast<float4*>(out)[i] = r;
  }
}

}  // namespace

extern "C" int psf_embed_tokens_f32(const int64_t* idx, const float* table, const float* pos, float* out, int64_t T,
                                    int64_t N, int32_t V, int32_t E, void* stream) {
  if (!idx || !table || !out) return psf_internal_fail(PSF_E_NULL, "psf_embed_tokens: idx, table and out must be non-NULL");
  if (T < 0 || N < 1 || V < 1 || E < 4 || (E & 3)) return psf_internal_fail(PSF_E_SHAPE, "psf_embed_tokens: need T >= 0, N >= 1, V >= 1, E a positive multiple of 4");
  if ((reinterpret_cast<uintptr_t>(table) & 15) || (reinterpret_cast<uintptr_t>(out) & 15) ||
      (pos && (reinterpret_cast<uintptr_t>(pos) & 15)) || (reinterpret_cast<uintptr_t>(idx) & 7))
    return psf_internal_fail(PSF_E_ALIGN, "psf_embed_tokens: table, pos and out must be 16-byte aligned, idx 8-byte aligned");
  if (T == 0) return PSF_OK;
  const int64_t total = T * (E / 4);
  const int64_t blocks = (total + 255) / 256;
  const int grid = (int)(blocks < 8192 ? blocks : 8192);
  hipLaunchKernelGGL(embed_tokens_k, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), idx, table, pos, out, T, N,
                     V, E / 4);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? PSF_OK : psf_internal_fail((int)e, hipGetErrorString(e));
}

// ------------------------------------------------------------------------------------------------------
// Rows of a narrow affine input layer: out[t, :] = x[t, 0..K) W^T + b, K <= 3 inputs per position
// (init_linear of the synthetic PSFNet, SyntheticExperiments/psf.py:153-154: Linear(2, 32) on [value, marker]).
// ------------------------------------------------------------------------------------------------------
// As a library GEMM the K = 2 product takes 80 us for 1 M positions; it is a 134 MB write. A thread computes four features of
// one position: x_0 w_e0, then fused adds of x_1 w_e1 and x_2 w_e2, then one rounded add of the bias — the arithmetic of the
// mixer kernels' AFFINE recipe (csrc/fwd_mlp_step.h: data_row8), so the two agree bit for bit.
namespace {

__global__ void __launch_bounds__(256)
affine_rows_k(const float* __restrict__ x, const float* __restrict__ W, const float* __restrict__ bias, float* __restrict__ out,
              int64_t T, int32_t K, int32_t E4) {
  // a thread keeps ONE group of four features (its weights and biases in registers) and walks the positions
  const int per = 256 / E4;  // positions per workgroup per pass (E4 <= 256: host-checked)
  const int tid = threadIdx.x, e4 = tid % E4, tl = tid / E4;
  if (tl >= per) return;
  float w0[4], w1[4], w2[4], bb[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float* w = W + (int64_t)(4 * e4 + j) * K;
    w0[j] = w[0], w1[j] = K > 1 ? w[1] : 0.f, w2[j] = K > 2 ? w[2] : 0.f;
    bb[j] = bias ? bias[4 * e4 + j] : 0.f;
  }
  for (int64_t t = (int64_t)blockIdx.x * per + tl; t < T; t += (int64_t)gridDim.x * per) {
    const float x0 = x[t * K], x1 = K > 1 ? x[t * K + 1] : 0.f, x2 = K > 2 ? x[t * K + 2] : 0.f;
    float r[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) r[j] = __fadd_rn(fmaf(x2, w2[j], fmaf(x1, w1[j], __fmul_rn(x0, w0[j]))), bb[j]);
    reinterpret_cast<float4*>(out)[t * E4 + e4] = make_float4(r[0], r[1], r[2], r[3]);
  }
}

}  // namespace

extern "C" int psf_affine_rows_f32(const float* x, const float* W, const float* bias, float* out, int64_t T, int32_t K, int32_t E,
                                   void* stream) {
  if (!x || !W || !out) return psf_internal_fail(PSF_E_NULL, "psf_affine_rows: x, W and out must be non-NULL");
  if (T < 0 || K < 1 || K > 3 || E < 4 || (E & 3) || E > 1024)
    return psf_internal_fail(PSF_E_SHAPE, "psf_affine_rows: need T >= 0, 1 <= K <= 3, E a multiple of 4 in 4..1024");
  if (reinterpret_cast<uintptr_t>(out) & 15) return psf_internal_fail(PSF_E_ALIGN, "psf_affine_rows: out must be 16-byte aligned");
  if (T == 0) return PSF_OK;
  const int per = 256 / (E / 4);
  const int64_t blocks = (T + per - 1) / per;
  const int grid = (int)(blocks < 8192 ? blocks : 8192);
  hipLaunchKernelGGL(affine_rows_k, dim3(grid), dim3(256), 0, reinterpret_cast<hipStream_t>(stream), x, W, bias, out, T, K, E / 4);
  const hipError_t e = hipGetLastError();
  return e == hipSuccess ? PSF_OK : psf_internal_fail((int)e, hipGetErrorString(e));
}

// ------------------------------------------------------------------------------------------------------
// Gradient of the table:  dTable[v, :] = sum over the tokens t with idx[t] == v of dOut[t, :]
// ------------------------------------------------------------------------------------------------------
// PyTorch's embedding_dense_backward sorts the indices and partitions them (rocprim) with sizes read back to the
// host: 0.1-0.26 ms per step on the LRA models, and not capturable in a HIP graph (a replay faulted in round 1,
// profiles/r01_graph_step_lab.log). Here ONE WAVE per (token slice, 64-column chunk) walks its tokens in order and
// accumulates into a private LDS table [V][64]: every table element has a single owner lane, so the sum over a
// slice is a fixed sequence of additions — no atomics, no sort, deterministic — and the slices are summed in a fixed
// order by a second kernel. Rows narrower than 64 floats put 64/E' tokens side by side in sub-tables (E' = E
// rounded up to a power of two), combined at the end.
namespace {

constexpr int kEmbSlicesMax = 2048;
constexpr int kEmbTokensPerSlice = 128;

__global__ void __launch_bounds__(64)
embed_bwd_partial_k(const int64_t* __restrict__ idx, const float* __restrict__ dOut, int64_t T, int32_t V, int32_t E,
                    int32_t ep_log2, int64_t tokens_per_slice, float* __restrict__ part) {
  extern __shared__ float tab[];  // [subs][V][ep], subs * ep == 64
  const int lane = threadIdx.x;
  const int ep = 1 << ep_log2, subs = 64 >> ep_log2;
  const int sub = lane >> ep_log2, col = lane & (ep - 1);
  const int c = blockIdx.y * 64 + col;
  const bool active = c < E;
  for (int i = lane; i < V * 64; i += 64) tab[i] = 0.f;
  const int64_t t0 = (int64_t)blockIdx.x * tokens_per_slice;
  const int64_t t1 = t0 + tokens_per_slice < T ? t0 + tokens_per_slice : T;
  float* mine = tab + (size_t)sub * V * ep + col;
  constexpr int U = 8;  // tokens in flight per sub-table (one wave per workgroup: loads in flight hide the latency)
  for (int64_t t = t0 + sub; t < t1; t += (int64_t)U * subs) {
    int v[U];
    float g[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t tu = t + (int64_t)u * subs;
      const bool ok = tu < t1;
      int64_t vv = ok ? idx[tu] : 0;
      vv = vv < 0 ? 0 : (vv >= V ? V - 1 : vv);
      v[u] = (int)vv;
      g[u] = (ok && active) ? dOut[tu * E + c] : 0.f;
    }
#pragma unroll
    for (int u = 0; u < U; ++u) mine[v[u] * ep] += g[u];  // in token order; a single wave's LDS accesses stay ordered
    // (ds_add_f32 without return instead of this read-add-write — same order, same bits — is SLOWER: 32 -> 110 us at the
    //  Temporal-Order vocabulary of 6, where consecutive adds hit the same addresses; 35 -> 38 us at 225 rows)
  }
  // sub-tables -> one table, in sub order; this slice's partial result
  float* out = part + (size_t)blockIdx.x * V * E;
  for (int i = lane; i < V * ep; i += 64) {
    const int vrow = i >> ep_log2, cc = i & (ep - 1);
    float s = tab[vrow * ep + cc];
    for (int q = 1; q < subs; ++q) s += tab[((size_t)q * V + vrow) * ep + cc];
    const int cg = blockIdx.y * 64 + cc;
    if (cg < E) out[(size_t)vrow * E + cg] = s;
  }
}

// dTable[i] = sum over the slices of part[q][i]. A workgroup owns 32 consecutive elements; its 32 slice lanes each add
// every 32nd slice in ascending order, then the 32 sub-sums are added in lane order: a fixed association. (With one
// thread per element the 192 outputs of Temporal Order took 235 us for 1024 dependent loads each.)
constexpr int kEmbRedLanes = 32;
__global__ void __launch_bounds__(32 * kEmbRedLanes)
embed_bwd_reduce_k(const float* __restrict__ part, int32_t slices, int64_t n, float* __restrict__ dTable) {
  __shared__ float sub[kEmbRedLanes][32];
  const int e = threadIdx.x & 31, sl = threadIdx.x >> 5;
  const int64_t i = (int64_t)blockIdx.x * 32 + e;
  float s = 0.f;
  if (i < n)
    for (int q = sl; q < slices; q += kEmbRedLanes) s += part[(size_t)q * n + i];
  sub[sl][e] = s;
  __syncthreads();
  if (sl == 0 && i < n) {
    float tot = sub[0][e];
#pragma unroll
    for (int q = 1; q < kEmbRedLanes; ++q) tot += sub[q][e];
    dTable[i] = tot;
  }
}

int emb_slices(int64_t T) {
  const int64_t s = (T + kEmbTokensPerSlice - 1) / kEmbTokensPerSlice;
  return (int)(s < 1 ? 1 : (s > kEmbSlicesMax ? kEmbSlicesMax : s));
}

}  // namespace

extern "C" int64_t psf_embed_tokens_bwd_workspace(int64_t T, int32_t V, int32_t E) {
  if (T < 1 || V < 1 || V > 512 || E < 1 || E > 4096) return -1;
  return (int64_t)emb_slices(T) * V * E * (int64_t)sizeof(float);
}

extern "C" int psf_embed_tokens_bwd_f32(const int64_t* idx, const float* dOut, int64_t T, int32_t V, int32_t E,
                                        float* dTable, void* workspace, int64_t workspace_bytes, void* stream) {
  if (!idx || !dOut || !dTable || !workspace) return psf_internal_fail(PSF_E_NULL, "psf_embed_tokens_bwd: NULL argument");
  const int64_t need = psf_embed_tokens_bwd_workspace(T, V, E);
  if (need < 0) return psf_internal_fail(PSF_E_SHAPE, "psf_embed_tokens_bwd: need T >= 1, 1 <= V <= 512, 1 <= E <= 4096");
  if (workspace_bytes < need) return psf_internal_fail(PSF_E_SHAPE, "psf_embed_tokens_bwd: workspace smaller than psf_embed_tokens_bwd_workspace(T, V, E)");
  const int slices = emb_slices(T);
  const int64_t tps = (T + slices - 1) / slices;
  int ep_log2 = 6;  // columns per sub-table: E rounded up to a power of two, at most 64
  while (ep_log2 > 0 && (1 << (ep_log2 - 1)) >= E) --ep_log2;
  const int chunks = (E + 63) / 64;
  const size_t lds = (size_t)V * 64 * sizeof(float);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipError_t e = hipSuccess;
  if (lds > 48 * 1024) e = hipFuncSetAttribute((const void*)embed_bwd_partial_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return psf_internal_fail((int)e, hipGetErrorString(e));
  float* part = reinterpret_cast<float*>(workspace);
  hipLaunchKernelGGL(embed_bwd_partial_k, dim3(slices, chunks), dim3(64), lds, s, idx, dOut, T, V, E, ep_log2, tps, part);
  const int64_t n = (int64_t)V * E;
  hipLaunchKernelGGL(embed_bwd_reduce_k, dim3((unsigned)((n + 31) / 32)), dim3(32 * kEmbRedLanes), 0, s, part, slices, n, dTable);
  e = hipGetLastError();
  return e == hipSuccess ? PSF_OK : psf_internal_fail((int)e, hipGetErrorString(e));
}
