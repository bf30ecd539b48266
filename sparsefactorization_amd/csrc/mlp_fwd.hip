// mlp_fwd.hip — fused forward of the token-wise MLPs that PRODUCE the chain's operands (inference path).
//
//   Y_k[t,:] = GELU(X[t,:] · A_k^T + a_k) · B_k^T + b_k        k < K MLPs sharing the input X [T, E]
//
// In PSFNet these are g and fs[0..M) (MLPBlock = Linear, GELU, Linear — SyntheticExperiments/psf.py:35-60,
// 110-126, called at :165,175): K = M+1 two-layer MLPs, all applied to the same `data`, producing V [T,C] and the
// M link-weight tensors W_m [T,L]. Through PyTorch that is 2K GEMMs + K GELUs that each stream the [T, h]
// hidden activations through HBM: 3.0 of the 4.05 ms of a forward at Adding N=16384, B=64, against 0.39 ms for
// the whole chain (profiles/r01_e2e_forward_split.log).
//
// Here a wave owns tiles of 32 tokens and keeps X's MFMA operand for them in registers for ALL K MLPs, so X is
// read from HBM once. Per MLP and tile, on the f32 matrix core (v_mfma_f32_32x32x2_f32, exact f32):
//   H^T[32 hidden x 32 tok] = A_k tile · X^T   (A from LDS, padded rows: conflict-free; X^T = the register operand)
//   GELU on the accumulator registers
//   Y^T[O x 32 tok]        += B_k tile · H^T   — the accumulator tile IS the next B operand: lane l holds row
//       (r&3)+8(r>>2)+4(l>>5) of H^T for token l&31 in register r, so feeding register r as the B operand of k-step
//       r pairs rows {row(r,0), row(r,1)}; the A operand takes B_k[o][that row] from LDS. No LDS round trip, no
//       shuffle (cdna_hip_programming.md §3 "An accumulator tile as the next MFMA's operand").
//   The Y^T tile is transposed through a per-wave LDS scratch and written as one contiguous 32*O-float burst.
// The hidden layer never exists in memory.
//
// Weights: a tiny prep kernel packs every MLP's zero-padded LDS image [A | a | B | b] into the caller's workspace
// once per call; the main kernel streams image k+1 into the second LDS buffer by LDS-DMA while MLP k computes
// (one barrier per MLP). The first version staged weights with per-MLP load -> ds_write loops between two
// barriers: an ablation (profiles/r01_mlp_ablation.log) put that at 0.55 of 1.10 ms.
//
// Limits: E <= 64 (even), h <= 128, O <= 32, K <= 32 per call; anything else stays on PyTorch (ListOps E = 512).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/psf_chord.h"
#include "mlp_fwd_x3.h"

#include <atomic>

extern "C" int psf_internal_fail(int code, const char* message);
extern std::atomic<int> psf_g_mlp_variant;  // psf_chord.hip: tuning knob "mlp_variant"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;

constexpr int kMaxMlps = 32;
constexpr int kMaxHidden = 128;

struct MlpDesc {
  const float* A;  // [h, E]
  const float* a;  // [h]
  const float* B;  // [O, h]
  const float* b;  // [O]
  float* Y;        // [T, O]
  int32_t h, O;
};

struct MlpArgs {
  MlpDesc d[kMaxMlps];
  const float* X;
  float* images;   // workspace: K packed LDS images of img_floats floats each
  int64_t T;
  int32_t E, K;
  int32_t hp_max;      // max over MLPs of h rounded up to 32
  int32_t img_floats;  // floats per image (multiple of 4)
};

// Image layout (floats), identical for every MLP of a call (sized by hp_max):
//   sA [hp_max][EP+1] | sa [hp_max] | sB [32][hp_max+1] | sb [32] | pad to a multiple of 4
__host__ __device__ inline int img_off_sa(int ep, int hp) { return hp * (ep + 1); }
__host__ __device__ inline int img_off_sB(int ep, int hp) { return img_off_sa(ep, hp) + hp; }
__host__ __device__ inline int img_off_sb(int ep, int hp) { return img_off_sB(ep, hp) + 32 * (hp + 1); }
__host__ __device__ inline int img_size(int ep, int hp) { return (img_off_sb(ep, hp) + 32 + 3) & ~3; }

__device__ __forceinline__ int cd_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

// erf by Abramowitz & Stegun 7.1.26: |error| <= 1.5e-7. One v_rcp, one v_exp, five FMAs (the device library's
// erff is ~50 VALU instructions).
__device__ __forceinline__ float erf_as(float x) {
  const float ax = fabsf(x);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = __expf(-ax * ax);
  return copysignf(fmaf(-p * t, e, 1.0f), x);
}

// GELU(x) = x * Phi(x), erf form as torch.nn.GELU(); absolute error <= 0.75e-7 * |x|
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752440f)); }

// One workgroup per MLP: write its zero-padded image.
__global__ void __launch_bounds__(256) mlp_pack_k(const MlpArgs a, int ep) {
  const MlpDesc d = a.d[blockIdx.x];
  const int hp = a.hp_max, E = a.E;
  float* img = a.images + (int64_t)blockIdx.x * a.img_floats;
  for (int i = threadIdx.x; i < a.img_floats; i += 256) {
    float v = 0.f;
    if (i < img_off_sa(ep, hp)) {
      const int r = i / (ep + 1), col = i - r * (ep + 1);
      if (r < d.h && col < E) v = d.A[r * E + col];
    } else if (i < img_off_sB(ep, hp)) {
      const int r = i - img_off_sa(ep, hp);
      if (r < d.h) v = d.a[r];
    } else if (i < img_off_sb(ep, hp)) {
      const int j = i - img_off_sB(ep, hp);
      const int o = j / (hp + 1), col = j - o * (hp + 1);
      if (o < d.O && col < d.h) v = d.B[o * d.h + col];
    } else {
      const int o = i - img_off_sb(ep, hp);
      if (o < d.O) v = d.b[o];
    }
    img[i] = v;
  }
}

// One MLP on one 32-token tile: H^T = A·X^T (+a), GELU, Y^T = B·H^T (+b), transposed through the wave's scratch and
// written as one contiguous burst. `img` is the MLP's LDS image, xr the wave's register operand of the tile.
template <int EP>
__device__ __forceinline__ void mlp_tile(const float* img, int hp, const float (&xr)[EP / 2], float* sw, float* __restrict__ yt,
                                         int n_el, int O, int lane) {
  constexpr int EH = EP / 2;  // k-steps of GEMM1
  constexpr int SA = EP + 1;  // padded row stride of the A_k image (floats)
  const int SB = hp + 1;      // padded row stride of the B_k image
  const int c = lane & 31, half = lane >> 5;
  const float* sA = img;
  const float* sa = img + img_off_sa(EP, hp);
  const float* sB = img + img_off_sB(EP, hp);
  const float* sb = img + img_off_sb(EP, hp);
  f32x16 acc2;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc2[r] = sb[cd_row(r, half)];
  for (int ht = 0; ht < hp; ht += 32) {
    f32x16 acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc1[r] = sa[ht + cd_row(r, half)];
    const float* arow = sA + (ht + c) * SA + half;
#pragma unroll
    for (int kk = 0; kk < EH; ++kk) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(arow[2 * kk], xr[kk], acc1, 0, 0, 0);
    const float* brow = sB + c * SB + ht;
    // Order pinned with sched_barrier: GELU(r) [VALU, ~14 instructions] ; MFMA(r) ; GELU(r+1) ; ... so that the
    // VALU work of register r+1 issues while the matrix core runs MFMA r (64 cycles, asynchronous). Left to
    // itself hipcc evaluates all 16 GELUs and then the 16 MFMAs, and the two pipes never overlap in a wave.
    float bw[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) bw[r] = brow[cd_row(r, half)];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float gv = gelu_erf(acc1[r]);
      __builtin_amdgcn_sched_barrier(0);
      acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(bw[r], gv, acc2, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // Y^T tile -> scratch[tok][o] (stride 33) -> one contiguous burst of n_el = rows*O floats. (Wave-private
  // scratch: the wave's own LDS writes are ordered before its reads, no barrier needed.)
#pragma unroll
  for (int r = 0; r < 16; ++r) sw[c * 33 + cd_row(r, half)] = acc2[r];
  // element e = lane + 64 i is (tok, o) = divmod(e, O): one division, then stepped by divmod(64, O)
  const int q64 = 64 / O, r64 = 64 - q64 * O;
  int tok = lane / O, o = lane - tok * O;
  for (int e = lane; e < n_el; e += 64) {
    yt[e] = sw[tok * 33 + o];
    tok += q64;
    o += r64;
    if (o >= O) {
      o -= O;
      ++tok;
    }
  }
}

// X operand of one tile: lane holds X[t0 + c][2*kk + half]. A tile's 32 rows are one contiguous 32*E-float
// burst: it is read with 16-byte loads into the wave's padded LDS scratch and the operand layout is read back from
// there (strided 4-byte loads straight from global cost 16 instructions x 32 cache lines per tile and thrashed the
// L1: 64 KB of rows per workgroup against a 32 KB cache).
template <int EP>
__device__ __forceinline__ void load_x_tile(const float* __restrict__ X, int64_t T, int E, int64_t t0, float* sw,
                                            float (&xr)[EP / 2], int lane) {
  constexpr int SA = EP + 1;
  const int c = lane & 31, half = lane >> 5;
  const int64_t rows_left = T - t0;
  const int nflt = (int)(rows_left >= 32 ? 32 : (rows_left > 0 ? rows_left : 0)) * E;  // floats of this tile in X
  const float* xt = X + t0 * E;
  for (int f = 4 * lane; f < 32 * E; f += 256) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (f < nflt) v = *reinterpret_cast<const float4*>(xt + f);  // E % 4 == 0: a float4 never straddles rows
    const int tok = f / E, col = f - tok * E;
    float* s = sw + tok * SA + col;
    s[0] = v.x;
    s[1] = v.y;
    s[2] = v.z;
    s[3] = v.w;
  }
#pragma unroll
  for (int kk = 0; kk < EP / 2; ++kk) {
    const int col = 2 * kk + half;
    xr[kk] = col < E ? sw[c * SA + col] : 0.f;
  }
}

// Streaming variant. EP: E padded to 32 or 64 (columns >= E are zero on both operands). TPW: token tiles per wave.
// Two image buffers: image k+1 streams in by LDS-DMA while MLP k computes; one barrier per MLP.
template <int EP, int TPW>
__global__ void __launch_bounds__(256)
mlp_fwd_k(const MlpArgs a) {
  constexpr int SA = EP + 1;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int hp = a.hp_max;
  const int img = a.img_floats;
  float* scratch = lds + 2 * img;  // [4 waves][32][EP+1]

  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  float* sw = scratch + wv * 32 * SA;  // per-wave scratch (X tile staging; the Y^T transpose uses stride 33)
  const int64_t tiles = (a.T + 31) / 32;
  const int64_t tiles_per_block = 4 * TPW;
  const int img_vecs = img >> 2;

  // flat LDS-DMA copy of image k into buffer (k & 1)
  auto stage = [&](int k) {
    const float* src = a.images + (int64_t)k * img;
    float* dst = lds + (k & 1) * img;
    for (int v0 = 0; v0 < img_vecs; v0 += 256) {
      const int v = v0 + tid;
      if (v < img_vecs)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 4 * v),
                                         (__attribute__((address_space(3))) void*)(dst + 4 * (v0 + (tid & ~63))), 16, 0, 0);
    }
  };

  for (int64_t blk = blockIdx.x; blk * tiles_per_block < tiles; blk += gridDim.x) {
    float xr[TPW][EP / 2];
    int64_t t0[TPW];
#pragma unroll
    for (int tp = 0; tp < TPW; ++tp) {
      t0[tp] = (blk * tiles_per_block + wv * TPW + tp) * 32;
      load_x_tile<EP>(a.X, a.T, a.E, t0[tp], sw, xr[tp], lane);
    }
    __syncthreads();  // the previous block's last MLP is done with both image buffers
    stage(0);

    for (int k = 0; k < a.K; ++k) {
      __syncthreads();  // image k has landed (hipcc drains vmcnt before the barrier); MLP k-1 is finished
      if (k + 1 < a.K) stage(k + 1);  // flies during this MLP's arithmetic
      const float* image = lds + (k & 1) * img;
      const int O = a.d[k].O;
      float* __restrict__ Yk = a.d[k].Y;
#pragma unroll
      for (int tp = 0; tp < TPW; ++tp) {
        if (t0[tp] >= a.T) continue;  // wave-uniform
        const int64_t rem = a.T - t0[tp];
        mlp_tile<EP>(image, hp, xr[tp], sw, Yk + t0[tp] * O, (int)(rem < 32 ? rem : 32) * O, O, lane);
      }
    }
  }
}

// Resident variant, for calls whose K images all fit in LDS next to the scratch (the h <= 32 networks: Adding /
// Order / CIFAR widths): every image is staged ONCE per workgroup, then the 8 waves run free — no barrier in the
// main loop, so waves drift apart and one wave's GELU / store phases overlap another's MFMA phases. (In the
// streaming variant each per-MLP barrier re-synchronises the waves, and the vmcnt(0) hipcc puts before it — needed
// for the LDS-DMA — also drains that MLP's Y stores.)
template <int EP, int TPW>
__global__ void __launch_bounds__(512)
mlp_fwd_resident_k(const MlpArgs a) {
  constexpr int SA = EP + 1;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int hp = a.hp_max;
  const int img = a.img_floats;
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  float* sw = lds + a.K * img + wv * 32 * SA;

  const int tot_vecs = (a.K * img) >> 2;
  for (int v0 = 0; v0 < tot_vecs; v0 += 512) {
    const int v = v0 + tid;
    if (v < tot_vecs)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a.images + 4 * v),
                                       (__attribute__((address_space(3))) void*)(lds + 4 * (v0 + (tid & ~63))), 16, 0, 0);
  }
  __syncthreads();  // the only barrier: all images have landed

  const int64_t groups = ((a.T + 31) / 32 + TPW - 1) / TPW;  // a group = TPW consecutive tiles, owned by one wave
  const int64_t nwaves = (int64_t)gridDim.x * 8;
  for (int64_t g = (int64_t)blockIdx.x * 8 + wv; g < groups; g += nwaves) {
    float xr[TPW][EP / 2];
    int64_t t0[TPW];
#pragma unroll
    for (int tp = 0; tp < TPW; ++tp) {
      t0[tp] = (g * TPW + tp) * 32;
      load_x_tile<EP>(a.X, a.T, a.E, t0[tp], sw, xr[tp], lane);
    }
    for (int k = 0; k < a.K; ++k) {
      const float* image = lds + k * img;
      const int O = a.d[k].O;
      float* __restrict__ Yk = a.d[k].Y;
#pragma unroll
      for (int tp = 0; tp < TPW; ++tp) {
        if (t0[tp] >= a.T) continue;  // wave-uniform
        const int64_t rem = a.T - t0[tp];
        mlp_tile<EP>(image, hp, xr[tp], sw, Yk + t0[tp] * O, (int)(rem < 32 ? rem : 32) * O, O, lane);
      }
    }
  }
}

constexpr size_t kLdsMax = 160 * 1024;

struct Plan {
  int ep, tpw, hp_max, img_floats;
  size_t lds_bytes;           // streaming variant: two image buffers + 4 wave scratches
  size_t lds_resident_bytes;  // resident variant: K images + 8 wave scratches (0: does not fit)
};

bool make_plan(int32_t E, int32_t K, const int32_t* h, const int32_t* O, Plan* p) {
  if (E < 4 || E > 64 || (E & 3) || K < 1 || K > kMaxMlps || !h || !O) return false;
  p->hp_max = 32;
  for (int k = 0; k < K; ++k) {
    if (h[k] < 1 || h[k] > kMaxHidden || O[k] < 1 || O[k] > 32) return false;
    const int hp = (h[k] + 31) & ~31;
    if (hp > p->hp_max) p->hp_max = hp;
  }
  p->ep = E <= 32 ? 32 : 64;
  p->tpw = p->ep == 32 ? 4 : 2;
  p->img_floats = img_size(p->ep, p->hp_max);
  p->lds_bytes = sizeof(float) * (2 * (size_t)p->img_floats + 4 * 32 * (size_t)(p->ep + 1));
  p->lds_resident_bytes = sizeof(float) * ((size_t)K * p->img_floats + 8 * 32 * (size_t)(p->ep + 1));
  if (p->lds_resident_bytes > kLdsMax) p->lds_resident_bytes = 0;
  return true;
}

}  // namespace

extern "C" {

int64_t psf_mlp_fwd_workspace(int32_t E, int32_t K, const int32_t* h, const int32_t* O) {
  Plan p;
  if (!make_plan(E, K, h, O, &p)) return -1;
  const int64_t f32_bytes = (int64_t)K * p.img_floats * (int64_t)sizeof(float);
  const int64_t x3_bytes = psf_x3_mlp_fwd_workspace(E, K, h, O);  // -1: that variant does not cover these sizes
  return x3_bytes > f32_bytes ? x3_bytes : f32_bytes;
}

int psf_mlp_fwd_f32(const float* X, int64_t T, int32_t E, int32_t K, const float* const* A, const float* const* a,
                    const float* const* B, const float* const* b, const int32_t* h, const int32_t* O,
                    float* const* Y, void* workspace, int64_t workspace_bytes, void* stream) {
  if (!X || !A || !a || !B || !b || !h || !O || !Y || !workspace)
    return psf_internal_fail(PSF_E_NULL, "psf_mlp_fwd: NULL argument");
  Plan p;
  if (T < 1 || !make_plan(E, K, h, O, &p))
    return psf_internal_fail(PSF_E_SHAPE, "psf_mlp_fwd: need T >= 1, E in {4,8,...,64}, 1 <= K <= 32, 1 <= h <= 128, 1 <= O <= 32");
  if ((reinterpret_cast<uintptr_t>(X) & 15) != 0) return psf_internal_fail(PSF_E_ALIGN, "psf_mlp_fwd: X must be 16-byte aligned");
  if (workspace_bytes < psf_mlp_fwd_workspace(E, K, h, O) || (reinterpret_cast<uintptr_t>(workspace) & 15) != 0)
    return psf_internal_fail(PSF_E_SHAPE, "psf_mlp_fwd: workspace too small (psf_mlp_fwd_workspace) or not 16-byte aligned");
  for (int k = 0; k < K; ++k)
    if (!A[k] || !a[k] || !B[k] || !b[k] || !Y[k]) return psf_internal_fail(PSF_E_NULL, "psf_mlp_fwd: NULL layer pointer");
  const int variant = psf_g_mlp_variant.load();
  // Split-bf16 variant (mlp_fwd_x3.hip): the same result to f32 accuracy on the bf16 matrix pipe, which overlaps
  // with the VALU work; default wherever it applies (E <= 32).
  const bool x3_ok = psf_x3_mlp_fwd_workspace(E, K, h, O) >= 0;
  if (variant == 3 && !x3_ok) return psf_internal_fail(PSF_E_TUNING, "psf_mlp_fwd: mlp_variant=3 needs E <= 32");
  if (x3_ok && (variant == 0 || variant == 3)) {
    const hipError_t e3 = psf_x3_mlp_fwd_launch(X, T, E, K, A, a, B, b, h, O, Y, workspace, reinterpret_cast<hipStream_t>(stream));
    return e3 == hipSuccess ? PSF_OK : psf_internal_fail((int)e3, hipGetErrorString(e3));
  }
  MlpArgs args;
  args.X = X;
  args.images = reinterpret_cast<float*>(workspace);
  args.T = T;
  args.E = E;
  args.K = K;
  args.hp_max = p.hp_max;
  args.img_floats = p.img_floats;
  for (int k = 0; k < K; ++k) args.d[k] = MlpDesc{A[k], a[k], B[k], b[k], Y[k], h[k], O[k]};
  for (int k = K; k < kMaxMlps; ++k) args.d[k] = MlpDesc{nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0};
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(mlp_pack_k, dim3(K), dim3(256), 0, s, args, p.ep);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return psf_internal_fail((int)e, hipGetErrorString(e));

  const int64_t tiles = (T + 31) / 32;
  if (variant == 2 && !p.lds_resident_bytes)
    return psf_internal_fail(PSF_E_TUNING, "psf_mlp_fwd: mlp_variant=2 but the K weight images do not fit in LDS");
  // Tiles per wave: as many as the registers allow (4 at EP = 32, 2 at EP = 64) when T is large; fewer for short
  // inputs so that the launch still has >= 2 workgroups per CU (Pathfinder B=64: 2048 tiles were 128 workgroups).
  int tpw = p.tpw;
  while (tpw > 1 && (tiles + 4 * tpw - 1) / (4 * tpw) < 512) tpw >>= 1;
  const int64_t groups = (tiles + p.tpw - 1) / p.tpw;
  const bool resident = p.lds_resident_bytes && variant != 1 && (variant == 2 || groups >= 2 * 8 * 256);
  if (resident) {
    // persistent: one 8-wave workgroup per CU (LDS-bound), waves stride over groups of TPW tiles
    const int grid = (int)((groups + 7) / 8 < 256 ? (groups + 7) / 8 : 256);
    const int lds = (int)p.lds_resident_bytes;
    if (p.ep == 32) {
      e = hipFuncSetAttribute((const void*)mlp_fwd_resident_k<32, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      if (e == hipSuccess) hipLaunchKernelGGL((mlp_fwd_resident_k<32, 4>), dim3(grid), dim3(512), lds, s, args);
    } else {
      e = hipFuncSetAttribute((const void*)mlp_fwd_resident_k<64, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      if (e == hipSuccess) hipLaunchKernelGGL((mlp_fwd_resident_k<64, 2>), dim3(grid), dim3(512), lds, s, args);
    }
  } else {
    const int64_t blocks_needed = (tiles + 4 * tpw - 1) / (4 * tpw);
    const int grid = (int)(blocks_needed < 2048 ? blocks_needed : 2048);
    auto launch = [&](auto kernel) {
      if (p.lds_bytes > 48 * 1024)
        e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds_bytes);
      if (e == hipSuccess) hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), p.lds_bytes, s, args);
    };
    if (p.ep == 32) {
      if (tpw == 4) launch(mlp_fwd_k<32, 4>);
      else if (tpw == 2) launch(mlp_fwd_k<32, 2>);
      else launch(mlp_fwd_k<32, 1>);
    } else {
      if (tpw == 2) launch(mlp_fwd_k<64, 2>);
      else launch(mlp_fwd_k<64, 1>);
    }
  }
  if (e == hipSuccess) e = hipGetLastError();
  return e == hipSuccess ? PSF_OK : psf_internal_fail((int)e, hipGetErrorString(e));
}

}  // extern "C"
