// linear_wgrad.hip — weight / bias gradient of the token-wise Linear layers that PRODUCE the chain's operands.
//
//   dWt[j,i] = sum_t dY[t,j] * X[t,i]      db[j] = sum_t dY[t,j]        t < T tokens (T = B*N, ~1e6),  i < m, j < n
//
// These are the layers of MLPBlock (SyntheticExperiments/psf.py:35-60): W_m = fs[m](data), V = g(data), with
// m, n in {2, 8, 15, 32, 128}. As a GEMM this is [n x T] * [T x m]: a reduction over a million rows into a tile
// of a few hundred numbers. Library GEMMs choose output-tiled kernels for it (measured r01: 0.8-1.3 ms per
// layer, 84 % of a training step, profiles/r01_train_step_profile.log); the job is memory-bound: read X and
// dY once (2 x 84 MB at E = h = 32, B = 40, N = 16384 => ~30 us at HBM speed).
//
// Kernel 1 (linear_wgrad_partial_k): the T rows are cut into one contiguous slab per wave; a wave streams its
// slab two rows at a time straight into the f32 matrix core, v_mfma_f32_32x32x2_f32 (exact f32 FMA chain, no
// reduced precision): A operand = dY rows (lane l holds dY[t + (l>>5)][j0 + (l&31)]), B operand = X rows, so
// every operand load is one coalesced 2-row burst and needs no LDS and no shuffle. Up to 4x4 accumulator tiles
// (128 x 128 outputs) per wave. The bias gradient is the running sum of the A operand.
// Kernel 2 (linear_wgrad_reduce_k): adds the per-wave partial tiles in a fixed order — no float atomics, so the
// result is bit-reproducible from run to run.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdio>

#include "../../include/psf_chord.h"

extern "C" int psf_internal_fail(int code, const char* message);  // psf_chord.hip: sets psf_last_error()

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kWaveRowsUnroll = 8;  // MFMA steps (2 rows each) whose loads are issued together
constexpr int kMaxTiles = 4;        // per dimension: up to 128 outputs

// C/D layout of v_mfma_f32_32x32x2_f32 (cdna_hip_programming.md §3): col = lane & 31,
// row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
__device__ __forceinline__ int cd_row(int reg, int lane) { return (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5); }

template <int TJ, int TI>
__global__ void __launch_bounds__(256)
linear_wgrad_partial_k(const float* __restrict__ X, int64_t ldx, const float* __restrict__ dY, int64_t ldy, int64_t T, int m, int n,
                       float* __restrict__ part, float* __restrict__ bpart, int64_t rows_per_wave, int nwaves) {
  const int lane = threadIdx.x & 63;
  const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int c = lane & 31, kk = lane >> 5;
  int64_t t_begin = (int64_t)w * rows_per_wave;  // waves past the last slab get an empty one (they still
  if (t_begin > T) t_begin = T;                  // take part in the workgroup reduction)
  int64_t t_end = t_begin + rows_per_wave;
  if (t_end > T) t_end = T;
  (void)nwaves;

  // per-lane column indices, clamped so every load is in bounds; out-of-range columns are zeroed by `jm` / `im`
  int jc[TJ], ic[TI];
  float jm[TJ], im[TI];
#pragma unroll
  for (int a = 0; a < TJ; ++a) {
    const int j = a * 32 + c;
    jc[a] = j < n ? j : n - 1;
    jm[a] = j < n ? 1.f : 0.f;
  }
#pragma unroll
  for (int b = 0; b < TI; ++b) {
    const int i = b * 32 + c;
    ic[b] = i < m ? i : m - 1;
    im[b] = i < m ? 1.f : 0.f;
  }

  f32x16 acc[TJ][TI];
  float bacc[TJ];
#pragma unroll
  for (int a = 0; a < TJ; ++a) {
    bacc[a] = 0.f;
#pragma unroll
    for (int b = 0; b < TI; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
  }

  constexpr int U = kWaveRowsUnroll;
  int64_t t = t_begin;
  // main loop: U steps of 2 rows, no row predicate
  for (; t + 2 * U <= t_end; t += 2 * U) {
    float av[U][TJ], bv[U][TI];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t row = t + 2 * u + kk;
#pragma unroll
      for (int a = 0; a < TJ; ++a) av[u][a] = dY[row * ldy + jc[a]];
#pragma unroll
      for (int b = 0; b < TI; ++b) bv[u][b] = X[row * ldx + ic[b]];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
#pragma unroll
      for (int a = 0; a < TJ; ++a) {
        const float aa = av[u][a] * jm[a];
        bacc[a] += aa;
#pragma unroll
        for (int b = 0; b < TI; ++b)
          acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(aa, bv[u][b] * im[b], acc[a][b], 0, 0, 0);
      }
    }
  }
  // tail: one step at a time, rows past the slab contribute zeros
  for (; t < t_end; t += 2) {
    const int64_t row = t + kk;
    const bool ok = row < t_end;
    const int64_t rc = ok ? row : t_end - 1;
#pragma unroll
    for (int a = 0; a < TJ; ++a) {
      const float aa = ok ? dY[rc * ldy + jc[a]] * jm[a] : 0.f;
      bacc[a] += aa;
#pragma unroll
      for (int b = 0; b < TI; ++b) {
        const float bb = ok ? X[rc * ldx + ic[b]] * im[b] : 0.f;
        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(aa, bb, acc[a][b], 0, 0, 0);
      }
    }
  }

  // combine the workgroup's four slabs in LDS (fixed order: wave 0 + 1 + 2 + 3), then ONE partial tile per
  // workgroup: part[wg][j][i] (TJ*32 x TI*32), bpart[wg][j]
  constexpr int MJ = TJ * 32, MI = TI * 32;
  __shared__ float red[3][16 * 64];
  __shared__ float redb[3][kMaxTiles * 32];
  const int wv = threadIdx.x >> 6;
#pragma unroll
  for (int a = 0; a < TJ; ++a) {
#pragma unroll
    for (int b = 0; b < TI; ++b) {
      if (wv > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wv - 1][r * 64 + lane] = acc[a][b][r];
      }
      __syncthreads();
      if (wv == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r)
          acc[a][b][r] = ((acc[a][b][r] + red[0][r * 64 + lane]) + red[1][r * 64 + lane]) + red[2][r * 64 + lane];
      }
      __syncthreads();
    }
  }
  float bs[TJ];
#pragma unroll
  for (int a = 0; a < TJ; ++a) {
    bs[a] = bacc[a] + __shfl_xor(bacc[a], 32, 64);
    if (wv > 0 && kk == 0) redb[wv - 1][a * 32 + c] = bs[a];
  }
  __syncthreads();
  if (wv == 0) {
    float* __restrict__ pw = part + (int64_t)blockIdx.x * MJ * MI;
#pragma unroll
    for (int a = 0; a < TJ; ++a)
#pragma unroll
      for (int b = 0; b < TI; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) pw[(a * 32 + cd_row(r, lane)) * MI + b * 32 + c] = acc[a][b][r];
    if (kk == 0) {
#pragma unroll
      for (int a = 0; a < TJ; ++a)
        bpart[(int64_t)blockIdx.x * MJ + a * 32 + c] =
            ((bs[a] + redb[0][a * 32 + c]) + redb[1][a * 32 + c]) + redb[2][a * 32 + c];
    }
  }
}

// out[j*m + i] = sum_w part[w][j][i]; db[j] = sum_w bpart[w][j]. Fixed order (bit-reproducible): a workgroup
// owns kRedOuts outputs; lane group g adds the partials w = g, g + kRedGroups, ... (independent loads, all in
// flight), then the kRedGroups group sums are added in ascending g.
constexpr int kRedOuts = 8, kRedGroups = 32;
static_assert(kRedOuts * kRedGroups == 256, "one 256-thread workgroup");

__global__ void __launch_bounds__(256)
linear_wgrad_reduce_k(const float* __restrict__ part, const float* __restrict__ bpart, int nwaves, int MJ, int MI,
                      int m, int n, float* __restrict__ dWt, float* __restrict__ db) {
  __shared__ float sums[kRedGroups][kRedOuts];
  const int ol = threadIdx.x & (kRedOuts - 1), grp = threadIdx.x >> 3;
  const int e = blockIdx.x * kRedOuts + ol;
  const int n_out = n * m + (db != nullptr ? n : 0);
  const float* p = part;
  int64_t stride = 0;
  if (e < n * m) {
    const int j = e / m, i = e - j * m;
    p = part + (int64_t)j * MI + i;
    stride = (int64_t)MJ * MI;
  } else if (e < n_out) {
    p = bpart + (e - n * m);
    stride = MJ;
  }
  float s = 0.f;
  if (e < n_out)
    for (int w = grp; w < nwaves; w += kRedGroups) s += p[(int64_t)w * stride];
  sums[grp][ol] = s;
  __syncthreads();
  if (grp == 0 && e < n_out) {
    float tot = 0.f;
#pragma unroll
    for (int g = 0; g < kRedGroups; ++g) tot += sums[g][ol];
    if (e < n * m) dWt[e] = tot;
    else db[e - n * m] = tot;
  }
}

struct Plan {
  int tj, ti, nwaves, nparts;
  int64_t rows_per_wave, part_floats, bpart_floats;
};

bool make_plan(int64_t T, int m, int n, Plan* p) {
  if (T < 1 || m < 1 || n < 1) return false;
  p->tj = (n + 31) / 32;
  p->ti = (m + 31) / 32;
  if (p->tj > kMaxTiles || p->ti > kMaxTiles) return false;
  const int64_t tile_floats = (int64_t)p->tj * p->ti * 1024;
  int64_t nw = 8192;                                   // 32 waves per CU: slabs overlap each other's load bursts
  const int64_t budget = ((int64_t)32 << 20) / 4;      // <= 32 MiB of partial tiles (one per 4 waves)
  if (nw / 4 * tile_floats > budget) nw = 4 * (budget / tile_floats);
  int64_t rpw = (T + nw - 1) / nw;
  rpw = (rpw + 1) & ~(int64_t)1;                       // even: a step consumes two rows
  if (rpw < 2 * kWaveRowsUnroll) rpw = 2 * kWaveRowsUnroll;
  nw = (T + rpw - 1) / rpw;
  p->nwaves = (int)nw;
  p->nparts = (int)((nw + 3) / 4);  // one partial tile per 4-wave workgroup
  p->rows_per_wave = rpw;
  p->part_floats = p->nparts * tile_floats;
  p->bpart_floats = (int64_t)p->nparts * p->tj * 32;
  return true;
}

template <int TJ>
hipError_t launch_ti(const Plan& p, const float* X, int64_t ldx, const float* dY, int64_t ldy, int64_t T, int m, int n,
                     float* part, float* bpart, hipStream_t s) {
  const dim3 grid((p.nwaves + 3) / 4), block(256);
  switch (p.ti) {
    case 1: hipLaunchKernelGGL((linear_wgrad_partial_k<TJ, 1>), grid, block, 0, s, X, ldx, dY, ldy, T, m, n, part, bpart, p.rows_per_wave, p.nwaves); break;
    case 2: hipLaunchKernelGGL((linear_wgrad_partial_k<TJ, 2>), grid, block, 0, s, X, ldx, dY, ldy, T, m, n, part, bpart, p.rows_per_wave, p.nwaves); break;
    case 3: hipLaunchKernelGGL((linear_wgrad_partial_k<TJ, 3>), grid, block, 0, s, X, ldx, dY, ldy, T, m, n, part, bpart, p.rows_per_wave, p.nwaves); break;
    case 4: hipLaunchKernelGGL((linear_wgrad_partial_k<TJ, 4>), grid, block, 0, s, X, ldx, dY, ldy, T, m, n, part, bpart, p.rows_per_wave, p.nwaves); break;
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace

extern "C" {

int64_t psf_linear_wgrad_workspace(int64_t T, int32_t m, int32_t n) {
  Plan p;
  if (!make_plan(T, m, n, &p)) return -1;
  return (p.part_floats + p.bpart_floats) * (int64_t)sizeof(float);
}

int psf_linear_wgrad_f32(const float* X, const float* dY, int64_t T, int32_t m, int32_t n, float* dWt, float* db,
                         void* workspace, int64_t workspace_bytes, void* stream) {
  return psf_linear_wgrad_strided_f32(X, m, dY, n, T, m, n, dWt, db, workspace, workspace_bytes, stream);
}

int psf_linear_wgrad_strided_f32(const float* X, int64_t ldx, const float* dY, int64_t ldy, int64_t T, int32_t m, int32_t n,
                                 float* dWt, float* db, void* workspace, int64_t workspace_bytes, void* stream) {
  if (!X || !dY || !dWt || !workspace) return psf_internal_fail(PSF_E_NULL, "psf_linear_wgrad: X, dY, dWt and workspace must be non-NULL");
  Plan p;
  if (!make_plan(T, m, n, &p)) return psf_internal_fail(PSF_E_SHAPE, "psf_linear_wgrad: need T >= 1 and 1 <= m, n <= 128");
  if (ldx < m || ldy < n) return psf_internal_fail(PSF_E_SHAPE, "psf_linear_wgrad: row strides must be >= the row lengths");
  if (workspace_bytes < (p.part_floats + p.bpart_floats) * (int64_t)sizeof(float))
    return psf_internal_fail(PSF_E_SHAPE, "psf_linear_wgrad: workspace smaller than psf_linear_wgrad_workspace(T, m, n)");
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  float* part = reinterpret_cast<float*>(workspace);
  float* bpart = part + p.part_floats;
  hipError_t e;
  switch (p.tj) {
    case 1: e = launch_ti<1>(p, X, ldx, dY, ldy, T, m, n, part, bpart, s); break;
    case 2: e = launch_ti<2>(p, X, ldx, dY, ldy, T, m, n, part, bpart, s); break;
    case 3: e = launch_ti<3>(p, X, ldx, dY, ldy, T, m, n, part, bpart, s); break;
    case 4: e = launch_ti<4>(p, X, ldx, dY, ldy, T, m, n, part, bpart, s); break;
    default: return psf_internal_fail(PSF_E_SHAPE, "psf_linear_wgrad: unsupported tile count");
  }
  if (e != hipSuccess) return psf_internal_fail((int)e, hipGetErrorString(e));
  const int outs = n * m + n;
  hipLaunchKernelGGL(linear_wgrad_reduce_k, dim3((outs + kRedOuts - 1) / kRedOuts), dim3(256), 0, s, part, bpart, p.nparts,
                     p.tj * 32, p.ti * 32, (int)m, (int)n, dWt, db);
  e = hipGetLastError();
  return e == hipSuccess ? PSF_OK : psf_internal_fail((int)e, hipGetErrorString(e));
}

}  // extern "C"
