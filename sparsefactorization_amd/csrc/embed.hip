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
