// mlp_bwd.hip — fused backward of the producer MLPs (training path of mlp_fwd.hip).
//
//   Y_k = GELU(X · A_k^T + a_k) · B_k^T + b_k,  k < K      (MLPBlock, SyntheticExperiments/psf.py:35-60; PSFNet
//   applies g and fs[0..M) to the same `data`, psf.py:165,175)
//
// Given dY_k, one launch produces dX = sum_k dHpre_k · A_k and per-wave partial sums of dA_k, da_k, dB_k, db_k; a
// two-stage fixed-order reduction (no float atomics: bit-reproducible) finishes the weight gradients. Nothing of
// size [T, h] ever exists in memory: the hidden layer is RECOMPUTED from X (16 MFMAs per 32-token tile), which
// is cheaper than the 2 x [T, K*h] floats autograd would save and re-read (2.35 GB at Order N=16384, B=40).
// Through PyTorch this backward is 4K GEMMs + K GELU-backward kernels + K-1 accumulations of dX: 4.9 ms of the
// 8.3 ms training step (profiles/r01_train_step_profile_after.log).
//
// A hidden layer wider than 32 is processed as independent 32-row "units" (u = (k, ht)): Hpre, dHpost, dHpre, dA
// and dB rows of a unit depend on no other unit; only dX (summed over all units) and db (taken from the ht = 0
// unit) cross units.
//
// Per wave: TPW tiles of 32 tokens; X's operand for the contraction over e (xr) is held in registers, the tiles
// themselves stay in LDS for the contraction over tokens (xT); the dX^T accumulators persist over all units. Per unit and tile, on
// v_mfma_f32_32x32x2_f32 (D[i][j] += A[i][k] B[k][j]; A-operand lane l = (i = l&31, k = l>>5), B-operand lane
// l = (k = l>>5, j = l&31), result register r of lane l = D[(r&3) + 8(r>>2) + 4(l>>5)][l&31]):
//   1. Hpre^T  [j x tok] = A_u · X^T + a_u                              16 MFMAs   (A from the LDS image, xr)
//   2. dHpost^T[j x tok] = B_u^T · dY^T                                 O/2 MFMAs  (B^T from the image, dY tile via LDS)
//   3. VALU: Phi, phi from one rcp + one exp;  Hpost = x Phi;  G = dHpost (Phi + x phi)   (G = dHpre^T)
//   4. dB_u^T  [j x o]  += Hpost^T · dY    (contraction over tokens)    16 MFMAs   (Hpost^T re-laid through LDS)
//   5. dA_u    [j x e]  += G · X           (contraction over tokens)    16 MFMAs   (G re-laid through LDS, xT)
//   6. dX^T    [e x tok]+= A_u^T · G       (contraction over j)         16 MFMAs   (G's accumulator registers ARE
//        the B operand: register r pairs rows {row(r,0), row(r,1)}; the A operand takes A_u[that row][e] from LDS)
//   da_u, db_u: sums of the step-5 / step-4 operand registers (their k index is the token).
// After the wave's TPW tiles the unit's 34 accumulator registers are summed over the workgroup's four waves
// through LDS and flushed to the workgroup's slot of the partial buffer.
//
// Limits: E <= 32 (multiple of 4), h <= 128, O <= 32, K <= 32; anything else stays on autograd.
//
// Three kernels share that structure (knob mlp_bwd_variant = 1, 2, 3; 0 = auto -> 3):
//   mlp_bwd_k      all five GEMMs on v_mfma_f32_32x32x2_f32. That instruction occupies the vector ALU's datapath
//                  (profiles/r01_mfmalab.log: MFMA cycles and VALU cycles add), so a tile-unit costs 72 x 64 MFMA
//                  cycles PLUS ~2400 VALU cycles: 1.17 ms at Order N=16384, B=40.
//   mlp_bwd_x3_k   steps 1, 2, 6 on v_mfma_f32_32x32x16_bf16 with the exact three-way bf16 split
//                  of mlp_x3_common.h; the token contractions (steps 4, 5) stay on the f32 instruction, their operands
//                  re-laid through f32 LDS tiles: 1.03 ms.
//   mlp_bwd_x3p_k  (default) all five GEMMs on the bf16 pipe, every operand split ONCE into bf16 planes in LDS that serve
//                  both orientations (row reads and ds_read_b64_tr_b16, mlp_planes.h): 0.86 ms. See the comment above it.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "../../include/psf_chord.h"
#include "mlp_planes.h"
#include "mlp_x3_common.h"

extern "C" int psf_internal_fail(int code, const char* message);
extern std::atomic<int> psf_g_mlp_bwd_variant;  // psf_chord.hip: tuning knob "mlp_bwd_variant"

namespace {

using psf_x3::bf16x8;
using psf_x3::cd_row;
using psf_x3::f32x16;
using psf_x3::Frag3;
using psf_x3::mfma6;
using psf_x3::split_pack8;
using psf_x3::split_pack8_pk;

constexpr int kMaxMlps = 32;
constexpr int kMaxUnits = 128;
constexpr int kImg = 32 * 33 + 32 + 32 * 33;  // sA [32][33] | sa [32] | sB [32 o][33]  = 2144 floats
constexpr int kOffSa = 32 * 33;
constexpr int kOffSb = kOffSa + 32;
constexpr int kPart = 1024 + 1024 + 64 + 64;  // per (group, unit): dA [j][e] | dB^T [j][o] | da [half][j] | db [half][o]
constexpr int kSlices = 64;                   // stage-1 reduction slices
static_assert(kPart % 4 == 0, "the stage-1 reduction reads the partial sums as float4");

// split-bf16 unit image (bytes), every part already in MFMA operand order:
//   A terms   3 x [32 j][80 B: 32 e bf16 + pad]            step 1, A operand (8 consecutive e of row j per ds_read_b128)
//   sa        32 f32
//   A^T terms 3 x [2 s][2 half][32 e][8 bf16]               step 6, A operand: A[ht + rho(s, half, i)][e]
//   B^T terms 3 x [2 s][2 half][32 j][8 bf16]               step 2, A operand: B[o = 16 s + 8 half + i][ht + j]
// rho(s, half, i) = (i & 3) + 16 s + 8 (i >> 2) + 4 half: the hidden row that accumulator register 8 s + i of a lane of
// that half holds, so a lane's registers 8s..8s+7 ARE its B-operand fragment of k-step s (accumulator-as-operand).
constexpr int kXARow = 80;
constexpr int kXATerm = 32 * kXARow;                // 2560
constexpr int kXOffSa = 3 * kXATerm;                // 7680
constexpr int kXOffAT = kXOffSa + 128;              // 7808
constexpr int kXFragTerm = 2 * 2 * 32 * 16;         // 2048
constexpr int kXOffBT = kXOffAT + 3 * kXFragTerm;   // 13952
constexpr int kXImgBytes = kXOffBT + 3 * kXFragTerm;  // 20096
constexpr int kXImgVecs = kXImgBytes / 16;          // 1256
constexpr int kImgFloatsMax = kXImgBytes / 4 > kImg ? kXImgBytes / 4 : kImg;  // workspace slot per unit, either variant

struct BwdMlp {
  const float* A;   // [h, E]
  const float* a;   // [h]
  const float* B;   // [O, h]
  const float* dY;  // [T, O]
  float* dA;
  float* da;
  float* dB;
  float* db;
  int32_t h, O;
};

struct BwdArgs {
  BwdMlp m[kMaxMlps];
  // unit -> MLP | hidden block << 8. Dwords, not bytes: a byte table indexed by the (uniform) unit counter becomes a
  // global_load_ubyte + s_waitcnt vmcnt(0) — a full memory round trip that also drains every prefetch in flight — where
  // a dword table is one s_load_dword.
  uint32_t unit[kMaxUnits];
  const float* X;
  float* dX;        // [T, E] or nullptr
  float* images;    // U images of kImg floats
  float* partials;  // [G][U][kPart], one slot per workgroup
  float* stage1;    // [kSlices][U * kPart]
  int64_t T;
  int64_t G;        // workgroups = partial slots
  int32_t E, K, U;
};

// y = GELU(x) = x Phi(x) and dy/dx = Phi(x) + x phi(x) for a PAIR of values on packed f32 math (v_pk_fma_f32 /
// v_pk_mul_f32: two elements per instruction; here VALU cycles add to the f32-MFMA cycles). Phi by Abramowitz &
// Stegun 26.2.17 (|error| <= 7.5e-8): t = 1/(1 + 0.2316419 |x|), q = phi(x) (b1 t + ... + b5 t^5), Phi = x >= 0 ?
// 1 - q : q, with 1/sqrt(2 pi) folded into the b's; E = exp(-x^2/2) = phi(x) sqrt(2 pi) serves both results.
using f32x2 = __attribute__((ext_vector_type(2))) float;
// The same on single values. Packed f32 VALU (v_pk_fma_f32 ...) halves the instruction count but is slow beside MFMAs
// (MI355X_MICROARCH.md constants table: one v_pk_fma_f32 costs +22 cycles against two v_fma_f32 next to a bf16 MFMA): the
// kernels whose matrix work runs on the separate bf16 pipe while a sibling wave does this arithmetic use the scalar form
// (planes kernel 0.873 -> 0.844 ms, profiles/r02ah_nopk.log), the f32-MFMA kernel (shared datapath anyway) the packed one.
__device__ __forceinline__ void gelu_and_grad1(float x, float& y, float& dydx) {
  const float t = __builtin_amdgcn_rcpf(fmaf(fabsf(x), 0.2316419f, 1.0f));
  float p = fmaf(0.53070271f, t, -0.72657602f);
  p = fmaf(p, t, 0.71070687f);
  p = fmaf(p, t, -0.14224837f);
  p = fmaf(p, t, 0.12741479f);
  p = p * t;
  const float E = __builtin_amdgcn_exp2f((x * x) * -0.72134752044448170368f);
  const float dlt = copysignf(0.5f - p * E, x);  // 0.5 - q >= 0
  const float Phi = 0.5f + dlt;
  y = x * Phi;
  dydx = fmaf(x * 0.39894228040143267794f, E, Phi);
}
__device__ __forceinline__ void gelu_and_grad2(f32x2 x, f32x2& y, f32x2& dydx) {
  f32x2 t;
  t.x = __builtin_amdgcn_rcpf(fmaf(fabsf(x.x), 0.2316419f, 1.0f));
  t.y = __builtin_amdgcn_rcpf(fmaf(fabsf(x.y), 0.2316419f, 1.0f));
  const f32x2 B5 = {0.53070271f, 0.53070271f}, B4 = {-0.72657602f, -0.72657602f}, B3 = {0.71070687f, 0.71070687f},
              B2 = {-0.14224837f, -0.14224837f}, B1 = {0.12741479f, 0.12741479f};
  f32x2 p = __builtin_elementwise_fma(B5, t, B4);
  p = __builtin_elementwise_fma(p, t, B3);
  p = __builtin_elementwise_fma(p, t, B2);
  p = __builtin_elementwise_fma(p, t, B1);
  p = p * t;
  const f32x2 c2 = {-0.72134752044448170368f, -0.72134752044448170368f};
  const f32x2 arg = (x * x) * c2;
  f32x2 E;
  E.x = __builtin_amdgcn_exp2f(arg.x);
  E.y = __builtin_amdgcn_exp2f(arg.y);
  const f32x2 half2 = {0.5f, 0.5f}, inv_sqrt_2pi = {0.39894228040143267794f, 0.39894228040143267794f};
  f32x2 dlt = half2 - p * E;  // 0.5 - q >= 0
  dlt.x = copysignf(dlt.x, x.x);
  dlt.y = copysignf(dlt.y, x.y);
  const f32x2 Phi = half2 + dlt;
  y = x * Phi;
  dydx = __builtin_elementwise_fma(x * inv_sqrt_2pi, E, Phi);
}

__global__ void __launch_bounds__(256) mlp_bwd_pack_k(const BwdArgs a) {
  const int u = blockIdx.x;
  const BwdMlp d = a.m[a.unit[u] & 0xff];
  const int ht = 32 * (int)(a.unit[u] >> 8), E = a.E;
  float* img = a.images + (int64_t)u * kImg;
  for (int i = threadIdx.x; i < kImg; i += 256) {
    float v = 0.f;
    if (i < kOffSa) {
      const int j = i / 33, e = i - j * 33;
      if (ht + j < d.h && e < E) v = d.A[(ht + j) * E + e];
    } else if (i < kOffSb) {
      const int j = i - kOffSa;
      if (ht + j < d.h) v = d.a[ht + j];
    } else {
      const int q = i - kOffSb, o = q / 33, j = q - o * 33;
      if (o < d.O && j < 32 && ht + j < d.h) v = d.B[o * d.h + ht + j];
    }
    img[i] = v;
  }
}

// Rows [t0, t0+32) of a row-major [T, W] array (W <= 32) into the wave's scratch S[tok][33]; rows >= T read as 0.
// The tile is one contiguous burst of 32*W floats. VEC4 needs W % 4 == 0 and a 16-byte-aligned base.
template <bool VEC4>
__device__ __forceinline__ void tile_to_scratch(const float* __restrict__ base, int64_t T, int W, int64_t t0, float* S, int lane) {
  const int64_t rows_left = T - t0;
  const int n = (int)(rows_left >= 32 ? 32 : (rows_left > 0 ? rows_left : 0)) * W;
  const float* src = base + t0 * W;
  if (VEC4) {
    for (int f = 4 * lane; f < 32 * W; f += 256) {
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (f < n) v = *reinterpret_cast<const float4*>(src + f);
      const int tok = f / W, col = f - tok * W;
      float* s = S + tok * 33 + col;
      s[0] = v.x;
      s[1] = v.y;
      s[2] = v.z;
      s[3] = v.w;
    }
  } else {
    const int q64 = 64 / W, r64 = 64 - q64 * W;
    int tok = lane / W, col = lane - tok * W;
    for (int f = lane; f < 32 * W; f += 64) {
      S[tok * 33 + col] = f < n ? src[f] : 0.f;
      tok += q64;
      col += r64;
      if (col >= W) {
        col -= W;
        ++tok;
      }
    }
  }
}

// launch_bounds(256, 2): two workgroups (8 waves) per CU need <= 256 registers per lane; hipcc then spills ~30
// registers, all but four scratch accesses of which sit in the per-block prologue / epilogue, not in the unit loop.
// Unbounded it takes 352 registers (one wave per SIMD, every LDS round trip exposed): 2.46 ms vs the bounded build
// at Order N=16384, B=40.
// NDY: registers of the dY prefetch = ceil(32 * max O / 64): 8 for O <= 16, else 16.
template <int TPW, int NDY>
__global__ void __launch_bounds__(256, 2)
mlp_bwd_k(const BwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int c = lane & 31, half = lane >> 5;
  // per wave: the TPW X tiles [tok][33] (B operand of step 5, read per use: keeping that orientation in registers
  // too cost 32 of them and pushed the unit loop into scratch spills), the dY tile [tok][SD], the re-layout tile
  constexpr int SD = NDY == 8 ? 17 : 33;  // odd strides: conflict-free by row and by column
  constexpr int kImgBufs = NDY == 8 ? 2 : 1;  // O > 16: one image buffer so that two workgroups still fit a CU
  constexpr int kWaveLds = TPW * 32 * 33 + 32 * SD + 32 * 33;
  float* SX = lds + kImgBufs * kImg + wv * kWaveLds;
  float* S1 = SX + TPW * 32 * 33;  // dY tile
  float* S2 = S1 + 32 * SD;        // re-layout of Hpost^T / G; dX^T at the end
  const int E = a.E, U = a.U;
  const int64_t tiles = (a.T + 31) / 32;
  const int64_t tiles_per_block = 4 * TPW;
  constexpr int img_vecs = kImg / 4;

  auto stage = [&](int u) {
    const float* src = a.images + (int64_t)u * kImg;
    float* dst = lds + (kImgBufs == 2 ? (u & 1) : 0) * kImg;
    for (int v0 = 0; v0 < img_vecs; v0 += 256) {
      const int v = v0 + tid;
      if (v < img_vecs)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 4 * v),
                                         (__attribute__((address_space(3))) void*)(dst + 4 * (v0 + (tid & ~63))), 16, 0, 0);
    }
  };

  for (int64_t blk = blockIdx.x; blk * tiles_per_block < tiles; blk += gridDim.x) {
    float xr[TPW][16];
    f32x16 dxa[TPW];
    int64_t t0[TPW];
#pragma unroll
    for (int tp = 0; tp < TPW; ++tp) {
      t0[tp] = (blk * tiles_per_block + wv * TPW + tp) * 32;
      float* sx = SX + tp * 32 * 33;
      tile_to_scratch<true>(a.X, a.T, E, t0[tp], sx, lane);
      if (E < 32)
        for (int i = lane; i < 32 * 32; i += 64)
          if ((i & 31) >= E) sx[(i >> 5) * 33 + (i & 31)] = 0.f;  // columns >= E read as zero
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) xr[tp][kk] = sx[c * 33 + 2 * kk + half];  // X[tok = c][e = 2kk+half]
#pragma unroll
      for (int r = 0; r < 16; ++r) dxa[tp][r] = 0.f;
    }
    float* part = a.partials + (blk * (int64_t)U) * kPart;  // one partial slot per workgroup
    // dY tiles are fetched one tile-unit ahead into registers (element lane + 64 i of the 32*O-float burst): with two
    // waves per SIMD a load consumed right after its issue exposes the whole memory latency once per tile-unit.
    float dyn[NDY];
    auto dy_fetch = [&](int u2, int64_t t02) {
      const BwdMlp& d2 = a.m[a.unit[u2] & 0xff];
      const int W = d2.O;
      const int64_t rows_left = a.T - t02;
      const int n = (int)(rows_left >= 32 ? 32 : (rows_left > 0 ? rows_left : 0)) * W;
      const float* src = d2.dY + t02 * W;
#pragma unroll
      for (int i = 0; i < NDY; ++i) {
        const int f = lane + 64 * i;
        dyn[i] = f < n ? src[f] : 0.f;
      }
    };
    auto dy_commit = [&](int W) {  // registers -> S1[tok][33]
      const int q64 = 64 / W, r64 = 64 - q64 * W;
      int tok = lane / W, col = lane - tok * W;
#pragma unroll
      for (int i = 0; i < NDY; ++i) {
        if (lane + 64 * i < 32 * W) S1[tok * SD + col] = dyn[i];
        tok += q64;
        col += r64;
        if (col >= W) {
          col -= W;
          ++tok;
        }
      }
    };
    dy_fetch(0, t0[0]);
    if (kImgBufs == 2) {
      __syncthreads();  // the previous block's last unit is done with both image buffers
      stage(0);
    }

    for (int u = 0; u < U; ++u) {
      __syncthreads();  // image u has landed (hipcc drains vmcnt before the barrier); unit u-1 is finished
      if (kImgBufs == 2) {
        if (u + 1 < U) stage(u + 1);
      } else {
        stage(u);
        __syncthreads();
      }
      const float* sA = lds + (kImgBufs == 2 ? (u & 1) : 0) * kImg;
      const float* sa = sA + kOffSa;
      const float* sB = sA + kOffSb;
      const BwdMlp& d = a.m[a.unit[u] & 0xff];
      const int O = d.O;

      f32x16 dA, dBT;
#pragma unroll
      for (int r = 0; r < 16; ++r) dA[r] = dBT[r] = 0.f;
      float da = 0.f, db = 0.f;

#pragma unroll
      for (int tp = 0; tp < TPW; ++tp) {
        dy_commit(O);
        if (tp + 1 < TPW)
          dy_fetch(u, t0[tp + 1 < TPW ? tp + 1 : 0]);
        else if (u + 1 < U)
          dy_fetch(u + 1, t0[0]);
        if (t0[tp] >= a.T) continue;  // wave-uniform
        // 1. Hpre^T
        f32x16 acc1, acc3;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          acc1[r] = sa[cd_row(r, half)];
          acc3[r] = 0.f;
        }
        const float* arow = sA + c * 33 + half;
#pragma unroll
        for (int kk = 0; kk < 16; ++kk) acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(arow[2 * kk], xr[tp][kk], acc1, 0, 0, 0);
        // 2. dHpost^T: A operand B[o = 2kk+half][j = c], B operand dY[tok = c][o = 2kk+half]
#pragma unroll 1
        for (int kk = 0; 2 * kk < O; ++kk) {
          const int o = 2 * kk + half;
          const float dyv = o < O ? S1[c * SD + o] : 0.f;
          acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(sB[o * 33 + c], dyv, acc3, 0, 0, 0);
        }
        // 3. GELU and its derivative; Hpost^T -> S2[j][tok]
        float g[16];
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          f32x2 y, dy;
          gelu_and_grad2(f32x2{acc1[r], acc1[r + 1]}, y, dy);
          S2[cd_row(r, half) * 33 + c] = y.x;
          S2[cd_row(r + 1, half) * 33 + c] = y.y;
          g[r] = acc3[r] * dy.x;  // G = dHpre^T
          g[r + 1] = acc3[r + 1] * dy.y;
        }
        // 4. dB^T += Hpost^T[j = c][tok = 2kk+half] · dY[tok = 2kk+half][o = c]
        {
          float dbs = 0.f;
#pragma unroll
          for (int kk = 0; kk < 16; ++kk) {
            const int q = 2 * kk + half;
            const float dyt = c < O ? S1[q * SD + c] : 0.f;
            dbs += dyt;
            dBT = __builtin_amdgcn_mfma_f32_32x32x2f32(S2[c * 33 + q], dyt, dBT, 0, 0, 0);
          }
          db += dbs;
        }
        // 5. dA += G[j = c][tok = 2kk+half] · X[tok = 2kk+half][e = c]   (G re-laid through S2)
#pragma unroll
        for (int r = 0; r < 16; ++r) S2[cd_row(r, half) * 33 + c] = g[r];
        {
          float das = 0.f;
#pragma unroll
          for (int kk = 0; kk < 16; ++kk) {
            const float gt = S2[c * 33 + 2 * kk + half];
            das += gt;
            dA = __builtin_amdgcn_mfma_f32_32x32x2f32(gt, SX[tp * 32 * 33 + (2 * kk + half) * 33 + c], dA, 0, 0, 0);
          }
          da += das;
        }
        // 6. dX^T += A_u^T · G: k-step r pairs hidden rows {row(r,0), row(r,1)}
        if (a.dX) {
#pragma unroll
          for (int r = 0; r < 16; ++r)
            dxa[tp] = __builtin_amdgcn_mfma_f32_32x32x2f32(sA[cd_row(r, half) * 33 + c], g[r], dxa[tp], 0, 0, 0);
        }
      }
      // Combine the four waves' partial sums through LDS (fixed order w = 0..3) and flush once per workgroup: a
      // quarter of the partial-buffer traffic of per-wave flushes (1.3 GB at Order N=16384, B=40, whose fixed-order
      // reduction alone took 0.38 ms).
      float* pu = part + (int64_t)u * kPart;
      const float* wave0 = lds + kImgBufs * kImg + TPW * 32 * 33;  // wave 0's S1; wave w's is + w * kWaveLds
      auto sum4 = [&](int off) {  // off: float offset from a wave's S1, 16-byte aligned
        float4 acc = *reinterpret_cast<const float4*>(wave0 + off);
#pragma unroll
        for (int w = 1; w < 4; ++w) {
          const float4 v = *reinterpret_cast<const float4*>(wave0 + w * kWaveLds + off);
          acc.x += v.x;
          acc.y += v.y;
          acc.z += v.z;
          acc.w += v.w;
        }
        return acc;
      };
#pragma unroll
      for (int r = 0; r < 16; ++r) S2[cd_row(r, half) * 32 + c] = dA[r];
      S1[lane] = da;
      S1[64 + lane] = db;
      __syncthreads();
      *reinterpret_cast<float4*>(pu + wv * 256 + 4 * lane) = sum4(32 * SD + wv * 256 + 4 * lane);
      if (wv == 0 && lane < 32) *reinterpret_cast<float4*>(pu + 2048 + 4 * lane) = sum4(4 * lane);  // da [2][32] | db [2][32]
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 16; ++r) S2[cd_row(r, half) * 32 + c] = dBT[r];
      __syncthreads();
      *reinterpret_cast<float4*>(pu + 1024 + wv * 256 + 4 * lane) = sum4(32 * SD + wv * 256 + 4 * lane);
    }
    __syncthreads();  // all combine reads of S2 are done before the dX epilogue reuses it

    if (a.dX) {
#pragma unroll
      for (int tp = 0; tp < TPW; ++tp) {
        if (t0[tp] >= a.T) continue;
        // dX^T tile -> S2[tok][e] -> one contiguous burst
#pragma unroll
        for (int r = 0; r < 16; ++r) S2[c * 33 + cd_row(r, half)] = dxa[tp][r];
        const int64_t rem = a.T - t0[tp];
        const int n_el = (int)(rem < 32 ? rem : 32) * E;
        float* xt = a.dX + t0[tp] * E;
        const int q64 = 64 / E, r64 = 64 - q64 * E;
        int tok = lane / E, e = lane - tok * E;
        for (int f = lane; f < n_el; f += 64) {
          xt[f] = S2[tok * 33 + e];
          tok += q64;
          e += r64;
          if (e >= E) {
            e -= E;
            ++tok;
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// split-bf16 variant
// ------------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) mlp_bwd_x3_pack_k(const BwdArgs a) {
  using psf_x3::bf16_bits;
  using psf_x3::split3;
  const int u = blockIdx.x;
  const BwdMlp d = a.m[a.unit[u] & 0xff];
  const int ht = 32 * (int)(a.unit[u] >> 8), E = a.E;
  unsigned char* img = reinterpret_cast<unsigned char*>(a.images) + (size_t)u * kXImgBytes;
  uint16_t* img16 = reinterpret_cast<uint16_t*>(img);
  float* img32 = reinterpret_cast<float*>(img);
  for (int i = threadIdx.x; i < 32 * 40; i += 256) {  // A terms [j][e] (the pad columns are written as zeros)
    const int j = i / 40, e = i - j * 40;
    const float v = (e < E && ht + j < d.h) ? d.A[(ht + j) * E + e] : 0.f;
    uint32_t t1, t2, t3;
    split3(v, t1, t2, t3);
    img16[(0 * kXATerm + j * kXARow) / 2 + e] = bf16_bits(t1);
    img16[(1 * kXATerm + j * kXARow) / 2 + e] = bf16_bits(t2);
    img16[(2 * kXATerm + j * kXARow) / 2 + e] = bf16_bits(t3);
  }
  for (int j = threadIdx.x; j < 32; j += 256) img32[kXOffSa / 4 + j] = ht + j < d.h ? d.a[ht + j] : 0.f;
  for (int q = threadIdx.x; q < 2 * 2 * 32 * 8; q += 256) {
    const int i = q & 7, col = (q >> 3) & 31, hf = (q >> 8) & 1, s = q >> 9;
    uint32_t t1, t2, t3;
    // A^T: [s][half][e = col][i] = A[ht + rho][e]
    const int rho = (i & 3) + 16 * s + 8 * (i >> 2) + 4 * hf;
    const float va = (col < E && ht + rho < d.h) ? d.A[(ht + rho) * E + col] : 0.f;
    split3(va, t1, t2, t3);
    img16[(kXOffAT + 0 * kXFragTerm) / 2 + q] = bf16_bits(t1);
    img16[(kXOffAT + 1 * kXFragTerm) / 2 + q] = bf16_bits(t2);
    img16[(kXOffAT + 2 * kXFragTerm) / 2 + q] = bf16_bits(t3);
    // B^T: [s][half][j = col][i] = B[o = 16 s + 8 half + i][ht + j]
    const int o = 16 * s + 8 * hf + i;
    const float vb = (o < d.O && ht + col < d.h) ? d.B[o * d.h + ht + col] : 0.f;
    split3(vb, t1, t2, t3);
    img16[(kXOffBT + 0 * kXFragTerm) / 2 + q] = bf16_bits(t1);
    img16[(kXOffBT + 1 * kXFragTerm) / 2 + q] = bf16_bits(t2);
    img16[(kXOffBT + 2 * kXFragTerm) / 2 + q] = bf16_bits(t3);
  }
}

__device__ __forceinline__ Frag3 load_frag3(const unsigned char* p, int term_stride) {
  Frag3 f;
  f.t1 = *reinterpret_cast<const bf16x8*>(p);
  f.t2 = *reinterpret_cast<const bf16x8*>(p + term_stride);
  f.t3 = *reinterpret_cast<const bf16x8*>(p + 2 * term_stride);
  return f;
}

// 512 threads: eight waves share one unit image. launch_bounds(512, 1): one workgroup (two waves per SIMD) per CU,
// <= 256 registers per lane. Steps 4 and 5 (the contractions over tokens) stay on the f32 instruction. (Builds with those
// two steps on bf16 through a second split per orientation, and with four waves per workgroup and two workgroups per CU,
// were measured and removed: 1.90 ms spilled / no gain)
template <int TPW, int NDY>
__global__ void __launch_bounds__(512, 1)
mlp_bwd_x3_k(const BwdArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int c = lane & 31, half = lane >> 5;
  constexpr int SD = NDY == 8 ? 17 : 33;      // odd strides: conflict-free by row and by column
  constexpr int NW = 8;
  constexpr int kImgBufs = NDY == 8 ? 2 : 1;  // O > 16 (wider dY tiles): one image buffer (LDS)
  constexpr int NS2 = NDY == 8 ? 1 : 2;       // k-steps of step 2 (16 outputs each)
  constexpr int kWaveLds = TPW * 32 * 33 + 32 * SD + 32 * 33;
  float* lds_f = reinterpret_cast<float*>(lds_raw + kImgBufs * kXImgBytes);
  float* SX = lds_f + wv * kWaveLds;  // the wave's X tiles [tok][33] (B operand of step 5)
  float* S1 = SX + TPW * 32 * 33;     // dY tile [tok][SD]
  float* S2 = S1 + 32 * SD;           // re-layout of Hpost^T / G; dX^T at the end
  const int E = a.E, U = a.U;
  const int64_t tiles = (a.T + 31) / 32;
  const int64_t tiles_per_block = NW * TPW;
  const unsigned char* images = reinterpret_cast<const unsigned char*>(a.images);

  auto stage = [&](int u) {
    const unsigned char* src = images + (size_t)u * kXImgBytes;
    unsigned char* dst = lds_raw + (kImgBufs == 2 ? (u & 1) : 0) * kXImgBytes;
    for (int v0 = 0; v0 < kXImgVecs; v0 += 64 * NW) {
      const int v = v0 + tid;
      if (v < kXImgVecs)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 16 * v),
                                         (__attribute__((address_space(3))) void*)(dst + 16 * (v0 + (tid & ~63))), 16, 0, 0);
    }
  };

  for (int64_t blk = blockIdx.x; blk * tiles_per_block < tiles; blk += gridDim.x) {
    Frag3 xf[TPW][2];  // X fragments: k-step s covers e = 16 s + 8 half + (0..7) of the lane's token (B operand, step 1)
    f32x16 dxa[TPW];
    int64_t t0[TPW];
#pragma unroll
    for (int tp = 0; tp < TPW; ++tp) {
      t0[tp] = (blk * tiles_per_block + wv * TPW + tp) * 32;
      float* sx = SX + tp * 32 * 33;
      tile_to_scratch<true>(a.X, a.T, E, t0[tp], sx, lane);
      if (E < 32)
        for (int i = lane; i < 32 * 32; i += 64)
          if ((i & 31) >= E) sx[(i >> 5) * 33 + (i & 31)] = 0.f;  // columns >= E read as zero
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = sx[c * 33 + 16 * s + 8 * half + i];
        xf[tp][s] = split_pack8(v);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) dxa[tp][r] = 0.f;
    }
    float* part = a.partials + (blk * (int64_t)U) * kPart;  // one partial slot per workgroup
    float dyn[NDY];
    auto dy_fetch = [&](int u2, int64_t t02) {
      const BwdMlp& d2 = a.m[a.unit[u2] & 0xff];
      const int W = d2.O;
      const int64_t rows_left = a.T - t02;
      const int n = (int)(rows_left >= 32 ? 32 : (rows_left > 0 ? rows_left : 0)) * W;
      const float* src = d2.dY + t02 * W;
#pragma unroll
      for (int i = 0; i < NDY; ++i) {
        const int f = lane + 64 * i;
        dyn[i] = f < n ? src[f] : 0.f;
      }
    };
    auto dy_commit = [&](int W) {  // registers -> S1[tok][SD]
      const int q64 = 64 / W, r64 = 64 - q64 * W;
      int tok = lane / W, col = lane - tok * W;
#pragma unroll
      for (int i = 0; i < NDY; ++i) {
        if (lane + 64 * i < 32 * W) S1[tok * SD + col] = dyn[i];
        tok += q64;
        col += r64;
        if (col >= W) {
          col -= W;
          ++tok;
        }
      }
    };
    dy_fetch(0, t0[0]);
    if (kImgBufs == 2) {
      __syncthreads();  // the previous block's last unit is done with both image buffers
      stage(0);
    }

    for (int u = 0; u < U; ++u) {
      __syncthreads();  // image u has landed (hipcc drains vmcnt before the barrier); unit u-1 is finished
      if (kImgBufs == 2) {
        if (u + 1 < U) stage(u + 1);
      } else {
        stage(u);
        __syncthreads();
      }
      const unsigned char* img = lds_raw + (kImgBufs == 2 ? (u & 1) : 0) * kXImgBytes;
      const float* sa = reinterpret_cast<const float*>(img + kXOffSa);
      const BwdMlp& d = a.m[a.unit[u] & 0xff];
      const int O = d.O;

      f32x16 dA, dBT;
#pragma unroll
      for (int r = 0; r < 16; ++r) dA[r] = dBT[r] = 0.f;
      float da = 0.f, db = 0.f;

#pragma unroll
      for (int tp = 0; tp < TPW; ++tp) {
        dy_commit(O);
        if (tp + 1 < TPW)
          dy_fetch(u, t0[tp + 1 < TPW ? tp + 1 : 0]);
        else if (u + 1 < U)
          dy_fetch(u + 1, t0[0]);
        // wave-uniform. (Computing past-the-end tiles on zeros instead, so that the wave's two tiles form ONE basic block
        // the scheduler can interleave, was tried: hipcc then overlaps both tiles' live ranges and spills, 1.05 -> 1.51 ms.)
        if (t0[tp] >= a.T) continue;
        // 1. Hpre^T = A_u X^T + a_u on the bf16 pipe
        f32x16 acc1, acc3;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          acc1[r] = sa[cd_row(r, half)];
          acc3[r] = 0.f;
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
          acc1 = mfma6(load_frag3(img + c * kXARow + 32 * s + 16 * half, kXATerm), xf[tp][s], acc1);
        // 2. dHpost^T = B_u^T dY^T: B operand = dY[tok = c][o = 16 s + 8 half + i], split here
#pragma unroll
        for (int s = 0; s < NS2; ++s) {
          float dv[8];
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int o = 16 * s + 8 * half + i;
            dv[i] = o < O ? S1[c * SD + o] : 0.f;
          }
          acc3 = mfma6(load_frag3(img + kXOffBT + ((s * 2 + half) * 32 + c) * 16, kXFragTerm), split_pack8_pk(dv), acc3);
        }
        // 3. GELU and its derivative; Hpost^T -> S2[j][tok]
        float g[16];
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          f32x2 y, dy;
          gelu_and_grad2(f32x2{acc1[r], acc1[r + 1]}, y, dy);
          S2[cd_row(r, half) * 33 + c] = y.x;
          S2[cd_row(r + 1, half) * 33 + c] = y.y;
          g[r] = acc3[r] * dy.x;  // G = dHpre^T
          g[r + 1] = acc3[r + 1] * dy.y;
        }
        {
          // 4. dB^T += Hpost^T[j = c][tok = 2kk+half] · dY[tok = 2kk+half][o = c] on the f32 instruction. (Reading the
          //    operands of four MFMAs ahead of them was tried: at two tiles per wave the 16 extra live registers go to
          //    scratch, 1.05 -> 1.52 ms.)
          {
            float dbs = 0.f;
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
              const int q = 2 * kk + half;
              const float dyt = c < O ? S1[q * SD + c] : 0.f;
              dbs += dyt;
              dBT = __builtin_amdgcn_mfma_f32_32x32x2f32(S2[c * 33 + q], dyt, dBT, 0, 0, 0);
            }
            db += dbs;
          }
          // 5. dA += G[j = c][tok = 2kk+half] · X[tok = 2kk+half][e = c]   (G re-laid through S2)
#pragma unroll
          for (int r = 0; r < 16; ++r) S2[cd_row(r, half) * 33 + c] = g[r];
          {
            float das = 0.f;
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
              const float gt = S2[c * 33 + 2 * kk + half];
              das += gt;
              dA = __builtin_amdgcn_mfma_f32_32x32x2f32(gt, SX[tp * 32 * 33 + (2 * kk + half) * 33 + c], dA, 0, 0, 0);
            }
            da += das;
          }
        }
        // 6. dX^T += A_u^T · G on the bf16 pipe: the lane's registers g[8s..8s+7] are its B fragment of k-step s
        if (a.dX) {
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            float gv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) gv[i] = g[8 * s + i];
            dxa[tp] = mfma6(load_frag3(img + kXOffAT + ((s * 2 + half) * 32 + c) * 16, kXFragTerm), split_pack8_pk(gv), dxa[tp]);
          }
        }
      }
      // Combine the eight waves' partial sums through LDS (fixed order w = 0..7) and flush once per workgroup.
      float* pu = part + (int64_t)u * kPart;
      const float* wave0 = lds_f + TPW * 32 * 33;  // wave 0's S1; wave w's is + w * kWaveLds
      auto sum8 = [&](int off) {  // off: float offset from a wave's S1, 16-byte aligned
        float4 acc = *reinterpret_cast<const float4*>(wave0 + off);
#pragma unroll
        for (int w = 1; w < NW; ++w) {
          const float4 v = *reinterpret_cast<const float4*>(wave0 + w * kWaveLds + off);
          acc.x += v.x;
          acc.y += v.y;
          acc.z += v.z;
          acc.w += v.w;
        }
        return acc;
      };
#pragma unroll
      for (int r = 0; r < 16; ++r) S2[cd_row(r, half) * 32 + c] = dA[r];
      S1[lane] = da;
      S1[64 + lane] = db;
      __syncthreads();
      if (wv < 4) *reinterpret_cast<float4*>(pu + wv * 256 + 4 * lane) = sum8(32 * SD + wv * 256 + 4 * lane);
      if (wv == NW - 4 && lane < 32) *reinterpret_cast<float4*>(pu + 2048 + 4 * lane) = sum8(4 * lane);  // da [2][32] | db [2][32]
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 16; ++r) S2[cd_row(r, half) * 32 + c] = dBT[r];
      __syncthreads();
      if (wv < 4) *reinterpret_cast<float4*>(pu + 1024 + wv * 256 + 4 * lane) = sum8(32 * SD + wv * 256 + 4 * lane);
    }
    __syncthreads();  // all combine reads of S2 are done before the dX epilogue reuses it

    if (a.dX) {
#pragma unroll
      for (int tp = 0; tp < TPW; ++tp) {
        if (t0[tp] >= a.T) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) S2[c * 33 + cd_row(r, half)] = dxa[tp][r];
        const int64_t rem = a.T - t0[tp];
        const int n_el = (int)(rem < 32 ? rem : 32) * E;
        float* xt = a.dX + t0[tp] * E;
        const int q64 = 64 / E, r64 = 64 - q64 * E;
        int tok = lane / E, e = lane - tok * E;
        for (int f = lane; f < n_el; f += 64) {
          xt[f] = S2[tok * 33 + e];
          tok += q64;
          e += r64;
          if (e >= E) {
            e -= E;
            ++tok;
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// split-bf16 variant on dual-use LDS planes (mlp_planes.h): ALL five GEMMs on the bf16 pipe, every operand split ONCE
// ------------------------------------------------------------------------------------------------------------------
// What mlp_bwd_x3_k pays for the two contractions over tokens (steps 4, 5) is a second split of dY, G and X in the
// transposed orientation — or, as built, the f32 instruction that shares the VALU's datapath. Here every activation is
// split once, stored as bf16 planes, and the orientation an MFMA needs is chosen by the READ: ds_read_b128 along a row,
// ds_read_b64_tr_b16 down a column (the hardware transposes 4 x 16 blocks on the way to the registers).
//   X tile      split at block start into three planes [tok][e] per tile: row read = step-1 B operand, transposed read =
//               step-5 B operand. No f32 copy of X in LDS, no X fragments held in registers.
//   dY tile     each lane loads the 8 (16) outputs of ITS token straight from global memory = its step-2 B fragment; the
//               same split terms go through a [tok][o] scratch plane and come back transposed as the step-4 B operand.
//   Hpost^T, G  accumulator layout -> split16 -> one ds_write_b64 per four registers into a [tok][j] scratch plane ->
//               transposed read = step-4 / step-5 A operand. G's packed terms ARE the step-6 B fragments.
//   A_u         one [j][e] plane per term: row read = step-1 A operand, transposed read in accumulator order = step-6 A
//               operand (the A^T copy of the x3 image is gone: 12.4 KB per unit image instead of 20 KB).
// A wave has TWO 2 KB scratch planes next to its X planes; the terms go through them two at a time and are read back
// transposed into registers (a plane is rewritten as soon as its reads are issued: LDS operations of one wave execute in
// order), so two tiles per wave x eight waves x two image buffers fit 160 KB.
// da_u = G 1: three more MFMAs per k-step on G's transposed terms against a fragment of ones (every column of the result
// is da_u; the matrix pipe has the slack, the VALU does not). db: per-lane sums of the dY registers, reduced over the 32
// token lanes once per unit (DPP), in the ht = 0 unit only.
constexpr int kPOffSa = 3 * psf_x3::kPlaneBytes;       // 6144: sa in accumulator-register order [half][16]
constexpr int kPOffBT = kPOffSa + 128;                 // 6272: B^T terms as in the x3 image
constexpr int kPImgBytes = kPOffBT + 3 * kXFragTerm;   // 12416
constexpr int kPImgVecs = kPImgBytes / 16;             // 776
constexpr int kPScrBytes = 2 * psf_x3::kPlaneBytes + 512;  // two scratch planes; as f32: combine tile [32][32] | da [2][32] | db [2][32]
static_assert(kPImgBytes <= kXImgBytes, "the workspace slot per unit is sized for the x3 image");

__global__ void __launch_bounds__(256) mlp_bwd_x3p_pack_k(const BwdArgs a) {
  using psf_x3::bf16_bits;
  using psf_x3::plane_off;
  using psf_x3::split3;
  const int u = blockIdx.x;
  const BwdMlp d = a.m[a.unit[u] & 0xff];
  const int ht = 32 * (int)(a.unit[u] >> 8), E = a.E;
  unsigned char* img = reinterpret_cast<unsigned char*>(a.images) + (size_t)u * kXImgBytes;
  uint16_t* img16 = reinterpret_cast<uint16_t*>(img);
  float* img32 = reinterpret_cast<float*>(img);
  for (int i = threadIdx.x; i < 32 * 32; i += 256) {  // A planes [j][e], swizzled
    const int j = i >> 5, e = i & 31;
    const float v = (e < E && ht + j < d.h) ? d.A[(ht + j) * E + e] : 0.f;
    uint32_t t1, t2, t3;
    split3(v, t1, t2, t3);
    const int at = (plane_off(j, e >> 3) >> 1) + (e & 7);
    img16[at] = bf16_bits(t1);
    img16[psf_x3::kPlaneBytes / 2 + at] = bf16_bits(t2);
    img16[psf_x3::kPlaneBytes + at] = bf16_bits(t3);
  }
  for (int q = threadIdx.x; q < 32; q += 256) {  // sa[half][r] = a[ht + cd_row(r, half)]
    const int j = ht + cd_row(q & 15, q >> 4);
    img32[kPOffSa / 4 + q] = j < d.h ? d.a[j] : 0.f;
  }
  for (int q = threadIdx.x; q < 2 * 2 * 32 * 8; q += 256) {  // B^T: [s][half][j = col][i] = B[o = 16 s + 8 half + i][ht + j]
    const int i = q & 7, col = (q >> 3) & 31, hf = (q >> 8) & 1, s = q >> 9;
    const int o = 16 * s + 8 * hf + i;
    const float vb = (o < d.O && ht + col < d.h) ? d.B[o * d.h + ht + col] : 0.f;
    uint32_t t1, t2, t3;
    split3(vb, t1, t2, t3);
    img16[(kPOffBT + 0 * kXFragTerm) / 2 + q] = bf16_bits(t1);
    img16[(kPOffBT + 1 * kXFragTerm) / 2 + q] = bf16_bits(t2);
    img16[(kPOffBT + 2 * kXFragTerm) / 2 + q] = bf16_bits(t3);
  }
}

template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// sum over the 32 lanes of the lane's half of the wave; every lane ends up with the total
__device__ __forceinline__ float half_sum(float v) {
  v = dpp_add<0xB1>(v);   // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);   // quad_perm [2,3,0,1]
  v = dpp_add<0x141>(v);  // row_half_mirror
  v = dpp_add<0x140>(v);  // row_mirror
  return v + __shfl_xor(v, 16, 64);
}

#ifdef PSF_X3P_TRACE  // profiles/x3plab.hip: shader-clock timestamps of one unit of one workgroup, per wave
__device__ unsigned long long psf_x3p_trace[8][32];
#define PSF_TRACE(slot)                                                        \
  do {                                                                         \
    if (trace_on) {                                                            \
      const unsigned long long t_ = clock64();                                 \
      if (lane == 0) psf_x3p_trace[wv][(slot)] = t_;                           \
    }                                                                          \
  } while (0)
#else
#define PSF_TRACE(slot) do { } while (0)
#endif

// 512 threads, one workgroup per CU (two waves per SIMD, <= 256 registers), two image buffers.
// NDY: dY values per lane and tile = outputs of one token that one half of the wave covers: 8 (O <= 16) or 16.
template <int TPW, int NDY>
__global__ void __launch_bounds__(512, 1)
mlp_bwd_x3p_k(const BwdArgs a) {
  using namespace psf_x3;
  constexpr int NW = 8;  // (twelve waves of one tile each, three per SIMD at 168 registers: 0.94 vs 0.86 ms, profiles/r02ae)
  constexpr int NS2 = NDY / 8;  // k-steps of the contraction over outputs (step 2)
  constexpr int kWaveBytes = TPW * 3 * kPlaneBytes + kPScrBytes;
  // static LDS: 2 images + 8 waves x (TPW x 3 X planes + scratch) = 160,000 bytes at TPW = 2
  __shared__ __attribute__((aligned(16))) unsigned char img_lds[2 * kPImgBytes];
  __shared__ __attribute__((aligned(16))) unsigned char wave_lds[NW * kWaveBytes];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 31, half = lane >> 5;
  unsigned char* XP = wave_lds + wv * kWaveBytes;  // the wave's X planes [tile][term]
  unsigned char* HP = XP + TPW * 3 * kPlaneBytes;                  // scratch plane: Hpost^T / G, one term at a time
  unsigned char* YP = HP + kPlaneBytes;                            // scratch plane: dY, one term at a time
  float* SCR = reinterpret_cast<float*>(HP);                       // the same bytes for the cross-wave combine
  const PlaneLane L = plane_lane(lane);
  const int E = a.E, U = a.U;
  const int64_t tiles = (a.T + 31) / 32;
  const int64_t tiles_per_block = NW * TPW;
  const unsigned char* images = reinterpret_cast<const unsigned char*>(a.images);

  // Image u by LDS-DMA: 16 bytes per lane, global -> LDS without passing through registers. Issued as inline assembly,
  // not through __builtin_amdgcn_global_load_lds: hipcc makes every later read of LDS that may alias a DMA in flight wait
  // for vmcnt(0) (its alias rule for LDS-DMA is all-or-nothing), and that wait also drains the dY prefetch issued a few
  // instructions earlier — a full memory round trip per unit at the first image read, measured (profiles/r02ac_x3plab.log).
  // Hidden from the compiler, the DMA costs one explicit s_waitcnt vmcnt(0) in front of the barrier that publishes the image.
  auto stage = [&](int u) {
    const unsigned char* src = images + (size_t)u * kXImgBytes;
    unsigned char* dst = img_lds + (u & 1) * kPImgBytes;
    for (int v0 = 0; v0 < kPImgVecs; v0 += 64 * NW) {
      const int v = v0 + tid;
      const uint32_t lds_at = __builtin_amdgcn_readfirstlane(
          (uint32_t)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)(dst + 16 * (v0 + (tid & ~63)))));
      uint32_t m0_saved;  // M0 = LDS address of the wave's 1 KB destination; the lane's 16 bytes go to M0 + 16 lane
      if (v < kPImgVecs)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(m0_saved) : "s"(lds_at), "v"(src + 16 * v) : "memory");
    }
  };

  for (int64_t blk = blockIdx.x; blk * tiles_per_block < tiles; blk += gridDim.x) {
    f32x16 dxa[TPW];
    int64_t t0[TPW];
#pragma unroll
    for (int tp = 0; tp < TPW; ++tp) {
      t0[tp] = (blk * tiles_per_block + wv * TPW + tp) * 32;
      const int64_t tok = t0[tp] + c;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int e0 = 16 * s + 8 * half;
        float4 lo = make_float4(0.f, 0.f, 0.f, 0.f), hi = lo;
        if (tok < a.T && e0 < E) lo = *reinterpret_cast<const float4*>(a.X + tok * E + e0);
        if (tok < a.T && e0 + 4 < E) hi = *reinterpret_cast<const float4*>(a.X + tok * E + e0 + 4);
        const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        const Frag3 f = split_pack8(v);
        unsigned char* xp = XP + tp * 3 * kPlaneBytes + L.row[s];
        *reinterpret_cast<bf16x8*>(xp) = f.t1;
        *reinterpret_cast<bf16x8*>(xp + kPlaneBytes) = f.t2;
        *reinterpret_cast<bf16x8*>(xp + 2 * kPlaneBytes) = f.t3;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) dxa[tp][r] = 0.f;
    }
    float* part = a.partials + (blk * (int64_t)U) * kPart;  // one partial slot per workgroup
    float dyn[NDY];
    // The lane's token, outputs 16 s + 8 half + i. Loads only (no use of the values here: anything that touches a loaded
    // register waits for it, and eight dependent round trips are what a select per load costs): rows past T and outputs
    // past O read an in-bounds neighbour instead. Outputs >= O meet zero columns of B^T in step 2 and land in columns of
    // dB^T / db that are never read; rows >= T are zeroed by dy_take in the one partial tile.
    auto dy_fetch = [&](int u2, int64_t t02) {
      const BwdMlp& d2 = a.m[a.unit[u2] & 0xff];
      const int W = d2.O;
      const int64_t tok = t02 + c;
      const float* src = d2.dY + (tok < a.T ? tok : 0) * W;
#pragma unroll
      for (int s = 0; s < NS2; ++s)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int o = 16 * s + 8 * half + i;
          dyn[8 * s + i] = src[o < W ? o : 0];
        }
    };
    dy_fetch(0, t0[0]);
    __syncthreads();  // the previous block's last unit is done with both image buffers
    stage(0);

    for (int u = 0; u < U; ++u) {
#ifdef PSF_X3P_TRACE
      const bool trace_on = blockIdx.x == 300 && u == 5;
#endif
      PSF_TRACE(0);
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's part of image u has landed
      __syncthreads();  // image u is complete; unit u-1's combine is finished
      PSF_TRACE(1);
      if (u + 1 < U) stage(u + 1);
      const unsigned char* img = img_lds + (u & 1) * kPImgBytes;
      const bool first_block_of_mlp = (a.unit[u] >> 8) == 0;

      f32x16 dA, dBT, dav;  // dav[r] = sum over tokens of G[cd_row(r, half)][tok], the same in every column
      float dbp[NDY];
#pragma unroll
      for (int r = 0; r < 16; ++r) dA[r] = dBT[r] = dav[r] = 0.f;
#pragma unroll
      for (int i = 0; i < NDY; ++i) dbp[i] = 0.f;
#pragma unroll
      for (int tp = 0; tp < TPW; ++tp) {
        PSF_TRACE(2 + 8 * tp);
        const unsigned char* xp = XP + tp * 3 * kPlaneBytes;
        // 1. Hpre^T = A_u X^T + a_u : both operands by row reads. In front of the first use of the dY registers: these twelve
        // MFMAs need the image and the X planes only, and cover what is left of the dY loads' (and, behind the partial-sum
        // stores of the previous unit, the stores') round trip.
        f32x16 acc1;
        {
          const float4* sa4 = reinterpret_cast<const float4*>(img + kPOffSa + 64 * half);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4 v = sa4[q];
            acc1[4 * q] = v.x;
            acc1[4 * q + 1] = v.y;
            acc1[4 * q + 2] = v.z;
            acc1[4 * q + 3] = v.w;
          }
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const Frag3 wa{row_frag(img, L, s), row_frag(img + kPlaneBytes, L, s), row_frag(img + 2 * kPlaneBytes, L, s)};
          const Frag3 xb{row_frag(xp, L, s), row_frag(xp + kPlaneBytes, L, s), row_frag(xp + 2 * kPlaneBytes, L, s)};
          acc1 = mfma6(wa, xb, acc1);
        }
        asm volatile("" : "+v"(acc1) : : "memory");
        // 0. this tile's dY values -> split terms (the lane's step-2 B fragments); start the next tile-unit's loads
        if (t0[tp] + 32 > a.T) {  // wave-uniform: the partial tile (and tiles past the end)
#pragma unroll
          for (int i = 0; i < NDY; ++i) dyn[i] = t0[tp] + c < a.T ? dyn[i] : 0.f;
        }
        Frag3 dy3[NS2];
#pragma unroll
        for (int s = 0; s < NS2; ++s) {
          const float v[8] = {dyn[8 * s], dyn[8 * s + 1], dyn[8 * s + 2], dyn[8 * s + 3],
                              dyn[8 * s + 4], dyn[8 * s + 5], dyn[8 * s + 6], dyn[8 * s + 7]};
          dy3[s] = split_pack8(v);
        }
        if (first_block_of_mlp) {
#pragma unroll
          for (int i = 0; i < NDY; ++i) dbp[i] += dyn[i];
        }
        // Pin the order "consume the old values, THEN issue the next loads". hipcc hoists the loads above the split, and the
        // wait in front of the split is then vmcnt(0) (partial-sum stores are still pending and make the counter's order
        // unknown to it), which drains the loads it has just issued: one memory round trip per tile-unit, measured.
#pragma unroll
        for (int s = 0; s < NS2; ++s)
          asm volatile("" : "+v"(dy3[s].t1), "+v"(dy3[s].t2), "+v"(dy3[s].t3) : : "memory");
#pragma unroll
        for (int i = 0; i < NDY; i += 4) asm volatile("" : "+v"(dbp[i]), "+v"(dbp[i + 1]), "+v"(dbp[i + 2]), "+v"(dbp[i + 3]) : : "memory");
        if (tp + 1 < TPW)
          dy_fetch(u, t0[tp + 1 < TPW ? tp + 1 : 0]);
        else if (u + 1 < U)
          dy_fetch(u + 1, t0[0]);
        if (t0[tp] >= a.T) continue;  // wave-uniform
        PSF_TRACE(3 + 8 * tp);
        // the same terms through the scratch planes [tok][o] and back transposed: the step-4 B operand. Issued first: the
        // round trips overlap with the MFMAs of steps 1 and 2.
        Frag3 yb[2];
        {
          const bf16x8 zero = __builtin_bit_cast(bf16x8, make_uint4(0u, 0u, 0u, 0u));  // outputs >= 16 of a narrow dY
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            *reinterpret_cast<bf16x8*>(HP + L.row[s]) = s < NS2 ? dy3[s < NS2 ? s : 0].t1 : zero;
            *reinterpret_cast<bf16x8*>(YP + L.row[s]) = s < NS2 ? dy3[s < NS2 ? s : 0].t2 : zero;
          }
          asm volatile("" ::: "memory");
#pragma unroll
          for (int s = 0; s < 2; ++s) yb[s].t1 = tr_frag(HP, L, s), yb[s].t2 = tr_frag(YP, L, s);
          asm volatile("" ::: "memory");
#pragma unroll
          for (int s = 0; s < 2; ++s) *reinterpret_cast<bf16x8*>(HP + L.row[s]) = s < NS2 ? dy3[s < NS2 ? s : 0].t3 : zero;
          asm volatile("" ::: "memory");
#pragma unroll
          for (int s = 0; s < 2; ++s) yb[s].t3 = tr_frag(HP, L, s);
          asm volatile("" ::: "memory");
        }

        f32x16 acc3;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc3[r] = 0.f;
        // 2. dHpost^T = B_u^T dY^T : B operand = the lane's own dY fragment
#pragma unroll
        for (int s = 0; s < NS2; ++s)
          acc3 = mfma6(load_frag3(img + kPOffBT + ((s * 2 + half) * 32 + c) * 16, kXFragTerm), dy3[s], acc3);
        // 3. GELU and its derivative
        PSF_TRACE(4 + 8 * tp);
        // Priority by phase: a wave in its vector phase (GELU, splits) yields to the SIMD's other wave whenever that one is in
        // a matrix phase, whose instructions need one issue slot in eight; raised again in front of the matrix steps below.
        // With step 1 ahead of the dY wait: 3 % (profiles/r03ah_x3p_experiments.log).
        __builtin_amdgcn_s_setprio(0);
        float y[16], g[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float dd;
          gelu_and_grad1(acc1[r], y[r], dd);
          g[r] = acc3[r] * dd;  // G = dHpre^T
        }
        // 4. dB^T[j][o] += Hpost^T[j][tok] dY[tok][o]        (both operands by transposed reads)
        // 5. dA[j][e] += G[j][tok] X[tok][e],  da[j] += G[j][tok] 1
        // 6. dX^T[e][tok] += A_u^T[e][j] G[j][tok]              (A^T by transposed reads of the image, G from registers)
        // The six terms of Hpost^T and G go through the two scratch planes in three rounds (LDS operations of one wave
        // execute in order, so a plane is rewritten as soon as its reads are issued). G is split while round 1 is in
        // flight; step 6, which needs no transposed activation, covers round 2, and step 4 covers round 3.
        PSF_TRACE(5 + 8 * tp);
        Frag3 ha[2], ga[2];
        const Split16 ys = split16(y);
        store_acc_plane(HP, L, ys, 0);
        store_acc_plane(YP, L, ys, 1);
        asm volatile("" ::: "memory");
        const Split16 gs = split16(g);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 2; ++s) ha[s].t1 = tr_frag(HP, L, s), ha[s].t2 = tr_frag(YP, L, s);
        asm volatile("" ::: "memory");
        store_acc_plane(HP, L, ys, 2);
        store_acc_plane(YP, L, gs, 0);
        asm volatile("" ::: "memory");
        if (a.dX) {
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            const Frag3 at{tr_frag_acc(img, L, s), tr_frag_acc(img + kPlaneBytes, L, s), tr_frag_acc(img + 2 * kPlaneBytes, L, s)};
            const Frag3 gb{acc_frag(gs, 0, s), acc_frag(gs, 1, s), acc_frag(gs, 2, s)};
            dxa[tp] = mfma6(at, gb, dxa[tp]);
          }
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int s = 0; s < 2; ++s) ha[s].t3 = tr_frag(HP, L, s), ga[s].t1 = tr_frag(YP, L, s);
        asm volatile("" ::: "memory");
        store_acc_plane(HP, L, gs, 1);
        store_acc_plane(YP, L, gs, 2);
        asm volatile("" ::: "memory");
        PSF_TRACE(6 + 8 * tp);
#pragma unroll
        for (int s = 0; s < 2; ++s) dBT = mfma6(ha[s], yb[s], dBT);
#pragma unroll
        for (int s = 0; s < 2; ++s) ga[s].t2 = tr_frag(HP, L, s), ga[s].t3 = tr_frag(YP, L, s);
        asm volatile("" ::: "memory");
        {
          const bf16x8 ones = __builtin_bit_cast(bf16x8, make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u));
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            const Frag3 xb{tr_frag(xp, L, s), tr_frag(xp + kPlaneBytes, L, s), tr_frag(xp + 2 * kPlaneBytes, L, s)};
            dA = mfma6(ga[s], xb, dA);
            dav = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[s].t3, ones, dav, 0, 0, 0);
            dav = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[s].t2, ones, dav, 0, 0, 0);
            dav = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[s].t1, ones, dav, 0, 0, 0);
          }
        }
      }
      // db: the per-lane sums over the 32 token lanes of each half (da came off the matrix pipe)
      PSF_TRACE(18);
      if (first_block_of_mlp) {
#pragma unroll
        for (int i = 0; i < NDY; ++i) dbp[i] = half_sum(dbp[i]);
      }
      // Combine the eight waves' partial sums through LDS (fixed order w = 0..7) and flush once per workgroup.
      float* pu = part + (int64_t)u * kPart;
      const float* wave0 = reinterpret_cast<const float*>(wave_lds + TPW * 3 * kPlaneBytes);  // wave 0's SCR
      auto sum8 = [&](int off) {  // off: float offset into a wave's SCR, 16-byte aligned
        float4 acc = *reinterpret_cast<const float4*>(wave0 + off);
#pragma unroll
        for (int w = 1; w < NW; ++w) {
          const float4 v = *reinterpret_cast<const float4*>(wave0 + w * (kWaveBytes / 4) + off);
          acc.x += v.x;
          acc.y += v.y;
          acc.z += v.z;
          acc.w += v.w;
        }
        return acc;
      };
#pragma unroll
      for (int r = 0; r < 16; ++r) SCR[cd_row(r, half) * 32 + c] = dA[r];
      SCR[1024 + lane] = 0.f;  // da [2][32]: the whole sum goes to row 0
      SCR[1088 + lane] = 0.f;  // db [2][32]
      if (c == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) SCR[1024 + cd_row(r, half)] = dav[r];
#pragma unroll
        for (int s = 0; s < NS2; ++s)
#pragma unroll
          for (int i = 0; i < 8; ++i) SCR[1088 + 16 * s + 8 * half + i] = dbp[8 * s + i];
      }
      PSF_TRACE(19);
      __syncthreads();
      PSF_TRACE(20);
      if (wv < 4) *reinterpret_cast<float4*>(pu + wv * 256 + 4 * lane) = sum8(wv * 256 + 4 * lane);
      if (wv == 4 && lane < 32) *reinterpret_cast<float4*>(pu + 2048 + 4 * lane) = sum8(1024 + 4 * lane);
      PSF_TRACE(21);
      __syncthreads();
      PSF_TRACE(22);
#pragma unroll
      for (int r = 0; r < 16; ++r) SCR[cd_row(r, half) * 32 + c] = dBT[r];
      __syncthreads();
      PSF_TRACE(23);
      if (wv < 4) *reinterpret_cast<float4*>(pu + 1024 + wv * 256 + 4 * lane) = sum8(wv * 256 + 4 * lane);
      PSF_TRACE(24);
    }

    if (a.dX) {  // the lane holds dX^T[e = 8 g + 4 half + (0..3)][tok = c] in registers 4 g .. 4 g + 3
#pragma unroll
      for (int tp = 0; tp < TPW; ++tp) {
        const int64_t tok = t0[tp] + c;
        if (tok < a.T) {
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            const int e0 = 8 * gq + 4 * half;
            if (e0 < E)
              *reinterpret_cast<float4*>(a.dX + tok * E + e0) =
                  make_float4(dxa[tp][4 * gq], dxa[tp][4 * gq + 1], dxa[tp][4 * gq + 2], dxa[tp][4 * gq + 3]);
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// the same arithmetic with the SIMD's two waves in different ROLES (knob mlp_bwd_variant = 4)
// ------------------------------------------------------------------------------------------------------------------
// In mlp_bwd_x3p_k every wave runs the whole chain of a tile — matrix steps 1-2, the GELU and the splits on the vector ALU,
// matrix steps 4-6 — and the SIMD's second wave runs the same chain: each wave's phases wait for each other, both waves
// meet four workgroup barriers per unit, and the matrix pipe and the vector ALU co-execute for an eighth of the time
// (profiles/r02ad_mlp_bwd_pmc.json). Here waves 0-3 are PRODUCERS (steps 1, 2, the GELU, the splits, step 6: everything
// that needs the token on the lane) and waves 4-7 CONSUMERS (the two contractions over tokens, steps 4 and 5, da and db:
// matrix instructions on transposed LDS reads only, and all weight-gradient sums); wave w and w + 4 share SIMD w
// (MI355X_MICROARCH.md: a workgroup's waves go to the SIMDs cyclically) and form a PAIR that owns TPW tiles of the block.
//   * The producer hands each tile-unit over as three pieces through single LDS slots of the pair: the dY terms (D slot),
//     the Hpost^T terms and the G terms (S slot, one after the other). Hand-over is by sequence counters in LDS that the
//     partner polls (s_sleep between polls): no workgroup barrier anywhere after the start of the kernel. LDS operations
//     of one wave execute in order, so "data, then counter" needs no wait on either side.
//   * The consumers own dA, dB^T, da, db of the unit: the cross-wave combine is among four waves, through the pair's slots
//     (free at that point), while the producers are already in the next unit's first tile; its arrival and done counters are
//     LDS atomics.
//   * The consumers also stage the unit images (LDS-DMA); a producer starts unit g when all four quarters of image g have
//     landed, a consumer overwrites a buffer when all four producers are done with the unit that used it.
//   * db = 1^T dY comes off the matrix pipe like da = G 1 (a fragment of ones as the A operand), so the producer has no
//     per-lane sums and no DPP reduction.
// All sums keep a fixed order: bit-reproducible. Every wave passes every counter of every (block, unit, tile) exactly once —
// tiles past the end of the input run as zero tiles — so no wave can wait for a partner that has left.
constexpr int kPsPairs = 4;
enum PsFlag : int { kFDReady = 0, kFDFree = 4, kFSReady = 8, kFSFree = 12, kFImgReady = 16, kFUnitDone = 20, kFCombArrive = 24, kFCombDone = 25, kFCount = 32 };

// The counters are addressed as LDS (address space 3): through a generic pointer a volatile access is a FLAT load with
// s_waitcnt vmcnt(0), which drains the dY prefetch on every poll (first build: 1.2 ms).
using ps_flag_t = volatile __attribute__((address_space(3))) uint32_t;
#ifdef PSF_PS_GUARD  // lab builds: a wait that never ends gives up, marks the launch and lets every later wait fall through
__device__ unsigned int psf_ps_stuck;
__device__ __forceinline__ void ps_wait_ge(const ps_flag_t* f, uint32_t target) {
  asm volatile("" ::: "memory");
  int spins = 0;
  while ((int32_t)(__builtin_amdgcn_readfirstlane(*f) - target) < 0) {
    __builtin_amdgcn_s_sleep(1);
    if (*(volatile unsigned int*)&psf_ps_stuck != 0u) break;
    if (++spins > (1 << 18)) {
      psf_ps_stuck = 1u + ((unsigned int)(uintptr_t)f >> 2) % 64u;
      break;
    }
  }
  asm volatile("" ::: "memory");
}
#else
__device__ __forceinline__ void ps_wait_ge(const ps_flag_t* f, uint32_t target) {
  asm volatile("" ::: "memory");
  while ((int32_t)(__builtin_amdgcn_readfirstlane(*f) - target) < 0) __builtin_amdgcn_s_sleep(1);
  asm volatile("" ::: "memory");
}
#endif
__device__ __forceinline__ void ps_wait4_ge(const ps_flag_t* f, uint32_t target) {
#pragma unroll
  for (int i = 0; i < kPsPairs; ++i) ps_wait_ge(f + i, target);
}
__device__ __forceinline__ void ps_signal(ps_flag_t* f, uint32_t value) {
  asm volatile("" ::: "memory");
  *f = value;
  asm volatile("" ::: "memory");
}

// Keeps a computation in the basic block it is written in: hipcc sinks pure arithmetic into the block of its first use,
// which for values handed over after a poll loop is the far side of the loop — out of the matrix instructions' shadow.
__device__ __forceinline__ void ps_pin(psf_x3::Split16& x) {
#pragma unroll
  for (int t = 0; t < 3; ++t)
    asm volatile("" : "+v"(x.d[t][0]), "+v"(x.d[t][1]), "+v"(x.d[t][2]), "+v"(x.d[t][3]), "+v"(x.d[t][4]), "+v"(x.d[t][5]),
                      "+v"(x.d[t][6]), "+v"(x.d[t][7]));
}
__device__ __forceinline__ void ps_pin(float (&x)[16]) {
#pragma unroll
  for (int t = 0; t < 16; t += 8)
    asm volatile("" : "+v"(x[t]), "+v"(x[t + 1]), "+v"(x[t + 2]), "+v"(x[t + 3]), "+v"(x[t + 4]), "+v"(x[t + 5]), "+v"(x[t + 6]),
                      "+v"(x[t + 7]));
}

template <int TPW, int NDY>
__global__ void __launch_bounds__(512, 1)
mlp_bwd_ps_k(const BwdArgs a) {
  using namespace psf_x3;
  constexpr int NS2 = NDY / 8;
  constexpr int kDyPlane = NDY == 8 ? 1024 : kPlaneBytes;  // [32 tok][16 o] in 32-byte rows, or a full swizzled plane
  constexpr int kXBytes = TPW * 3 * kPlaneBytes;
  constexpr int kPairBytes = kXBytes + 3 * kPlaneBytes + 3 * kDyPlane;
  static_assert(3 * kPlaneBytes + 3 * kDyPlane >= (2048 + 64) * 4, "the pair's slots hold a consumer's sums for the combine");
  static_assert(2 * kPImgBytes + kPsPairs * kPairBytes + kFCount * 4 <= 160 * 1024, "LDS");
  __shared__ __attribute__((aligned(16))) unsigned char img_lds[2 * kPImgBytes];
  __shared__ __attribute__((aligned(16))) unsigned char pair_lds[kPsPairs * kPairBytes];
  __shared__ uint32_t flag_lds[kFCount];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 31, half = lane >> 5;
  const int pr = wv & 3;
  const bool producer = wv < 4;
  unsigned char* XP = pair_lds + pr * kPairBytes;  // the pair's X planes [tile][term]
  unsigned char* SP = XP + kXBytes;                // S slot: three planes [tok][j], Hpost^T terms or G terms
  unsigned char* DP = SP + 3 * kPlaneBytes;        // D slot: three dY planes [tok][o]
  ps_flag_t* F = (ps_flag_t*)flag_lds;
  const PlaneLane L = plane_lane(lane);
  const int E = a.E, U = a.U;
  const int64_t tiles = (a.T + 31) / 32;
  const int64_t tiles_per_block = kPsPairs * TPW;
  const unsigned char* images = reinterpret_cast<const unsigned char*>(a.images);
  if (tid < kFCount) flag_lds[tid] = 0u;
  __syncthreads();

  uint32_t n = 0;  // tile-units of this pair so far (= pieces of the D slot; the S slot has had 2 n)
  uint32_t g = 0;  // units of this workgroup so far: image g lives in buffer g & 1
  for (int64_t blk = blockIdx.x; blk * tiles_per_block < tiles; blk += gridDim.x) {
    const bool last_block = (blk + gridDim.x) * tiles_per_block >= tiles;
    const int64_t tile0 = blk * tiles_per_block + pr * TPW;  // the pair's first tile
    if (producer) {
      // ---------------------------------------------------------------------------------------------- producer
      // Software pipeline over the pair's U x TPW tile-units (i = u TPW + tp): while the vector ALU runs the GELU and the
      // splits of tile-unit i, the matrix pipe runs steps 1 and 2 of tile-unit i + 1. A dependent MFMA blocks the wave's
      // in-order issue until its predecessor retires, so the two only overlap when they alternate in the instruction
      // stream (sched_group_barrier below).
      const int NI = U * TPW;
      float dyn[NDY];
      // The lane's token, outputs 16 s + 8 half + (0..7). Loads only, no use of the values here (see mlp_bwd_x3p_k): rows past
      // T and outputs past O read an in-bounds neighbour instead. The unit's dY pointer and width are fetched from the
      // kernel arguments one step ahead (dy_unit): two dependent scalar loads in front of the address arithmetic cost a
      // tile-unit ~400 clocks.
      const float* dy_base = nullptr;
      int dy_w = 0;
      auto dy_unit = [&](int u2) {
        const BwdMlp& d2 = a.m[a.unit[u2] & 0xff];
        dy_base = d2.dY;
        dy_w = d2.O;
      };
      auto dy_fetch = [&](int tp2) {
        const int W = dy_w;
        const int64_t tok = (tile0 + tp2) * 32 + c;
        const float* src = dy_base + (tok < a.T ? tok : 0) * W;
#pragma unroll
        for (int s = 0; s < NS2; ++s)
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int o = 16 * s + 8 * half + i;
            dyn[8 * s + i] = src[o < W ? o : 0];
          }
      };
      auto dy_split = [&](int i2, Frag3 (&out)[NS2]) {  // consumes dyn
        const int tp2 = i2 % TPW;
        const int64_t t02 = (tile0 + tp2) * 32;
        if (t02 + 32 > a.T) {  // wave-uniform: the partial tile and tiles past the end
#pragma unroll
          for (int i = 0; i < NDY; ++i) dyn[i] = t02 + c < a.T ? dyn[i] : 0.f;
        }
#pragma unroll
        for (int s = 0; s < NS2; ++s) {
          const float v[8] = {dyn[8 * s], dyn[8 * s + 1], dyn[8 * s + 2], dyn[8 * s + 3],
                              dyn[8 * s + 4], dyn[8 * s + 5], dyn[8 * s + 6], dyn[8 * s + 7]};
          out[s] = split_pack8(v);
        }
#pragma unroll
        for (int s = 0; s < NS2; ++s) asm volatile("" : "+v"(out[s].t1), "+v"(out[s].t2), "+v"(out[s].t3) : : "memory");
      };
      auto bias_acc = [&](const unsigned char* img) {
        f32x16 acc;
        const float4* sa4 = reinterpret_cast<const float4*>(img + kPOffSa + 64 * half);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 v = sa4[q];
          acc[4 * q] = v.x;
          acc[4 * q + 1] = v.y;
          acc[4 * q + 2] = v.z;
          acc[4 * q + 3] = v.w;
        }
        return acc;
      };
      dy_unit(0);
      dy_fetch(0);
      ps_wait_ge(F + kFSFree + pr, 2u * n);  // the consumer has read the previous block's X planes for the last time
#pragma unroll
      for (int tp = 0; tp < TPW; ++tp) {
        const int64_t tok = (tile0 + tp) * 32 + c;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const int e0 = 16 * s + 8 * half;
          float4 lo = make_float4(0.f, 0.f, 0.f, 0.f), hi = lo;
          if (tok < a.T && e0 < E) lo = *reinterpret_cast<const float4*>(a.X + tok * E + e0);
          if (tok < a.T && e0 + 4 < E) hi = *reinterpret_cast<const float4*>(a.X + tok * E + e0 + 4);
          const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
          const Frag3 f = split_pack8(v);
          unsigned char* xp = XP + tp * 3 * kPlaneBytes + L.row[s];
          *reinterpret_cast<bf16x8*>(xp) = f.t1;
          *reinterpret_cast<bf16x8*>(xp + kPlaneBytes) = f.t2;
          *reinterpret_cast<bf16x8*>(xp + 2 * kPlaneBytes) = f.t3;
        }
      }
      // prologue: steps 1 and 2 of tile-unit 0
      f32x16 acc1, acc3;
      Frag3 dy3[NS2];
      ps_wait4_ge(F + kFImgReady, g + 1u);  // all four quarters of image g have landed
      {
        const unsigned char* img = img_lds + (g & 1u) * kPImgBytes;
        acc1 = bias_acc(img);
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const Frag3 wa{row_frag(img, L, s), row_frag(img + kPlaneBytes, L, s), row_frag(img + 2 * kPlaneBytes, L, s)};
          const Frag3 xb{row_frag(XP, L, s), row_frag(XP + kPlaneBytes, L, s), row_frag(XP + 2 * kPlaneBytes, L, s)};
          acc1 = mfma6(wa, xb, acc1);
        }
        dy_split(0, dy3);
        if (TPW == 1 && U > 1) dy_unit(1);
        dy_fetch(TPW > 1 ? 1 : 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) acc3[r] = 0.f;
#pragma unroll
        for (int s = 0; s < NS2; ++s)
          acc3 = mfma6(load_frag3(img + kPOffBT + ((s * 2 + half) * 32 + c) * 16, kXFragTerm), dy3[s], acc3);
        if (TPW == 1) ps_signal(F + kFUnitDone + pr, g + 1u);
      }
      int un = 0, tpn = 0;  // unit and tile of tile-unit i + 1 (clamped to the last one)
      int u2 = TPW > 1 ? 0 : (U > 1 ? 1 : 0), tp2 = TPW > 1 ? 1 : 0;  // ... of tile-unit i + 2, whose dY is fetched in step i
      for (int i = 0; i < NI; ++i, ++n) {
#ifdef PSF_X3P_TRACE
        const bool trace_on = blockIdx.x == 300 && i / TPW == 5 && i % TPW < 2;
        const int tslot = 8 * (i % TPW);
#endif
        PSF_TRACE(tslot + 0);
        if (i + 2 < NI) {
          if (++tp2 == TPW) {
            tp2 = 0;
            dy_unit(++u2);  // scalar loads now, used in A2
          }
        }
        const bool next = i + 1 < NI;
        if (next) {
          if (++tpn == TPW) tpn = 0, ++un;
          if (tpn == 0) ps_wait4_ge(F + kFImgReady, g + (uint32_t)un + 1u);  // the next unit's image has landed
        }
        const unsigned char* imgn = img_lds + ((g + (uint32_t)un) & 1u) * kPImgBytes;
        const unsigned char* xpn = XP + tpn * 3 * kPlaneBytes;
        PSF_TRACE(tslot + 1);
        // A1: GELU and its derivative of tile-unit i, split of Hpost^T  ||  step 1 of tile-unit i + 1
        f32x16 acc1n = bias_acc(imgn);
        Frag3 wa[2], xb[2];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          wa[s] = Frag3{row_frag(imgn, L, s), row_frag(imgn + kPlaneBytes, L, s), row_frag(imgn + 2 * kPlaneBytes, L, s)};
          xb[s] = Frag3{row_frag(xpn, L, s), row_frag(xpn + kPlaneBytes, L, s), row_frag(xpn + 2 * kPlaneBytes, L, s)};
        }
        // Source order IS the issue order here (sched_barrier after every step): the GELU of one pair of values (or the
        // split of two pairs), then one matrix instruction, which runs while the next step's vector work issues.
        // The two counters the hand-over needs are read NOW and looked at after the block: a poll costs an LDS round trip
        // even when the partner is long done.
        const uint32_t d_free_early = F[kFDFree + pr], s_free_early = F[kFSFree + pr];
        float y[16], gg[16];
        Split16 ys;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 12; ++m) {
          if (m < 8) {
            float d0, d1;
            gelu_and_grad1(acc1[2 * m], y[2 * m], d0);
            gelu_and_grad1(acc1[2 * m + 1], y[2 * m + 1], d1);
            gg[2 * m] = acc3[2 * m] * d0;  // G = dHpre^T
            gg[2 * m + 1] = acc3[2 * m + 1] * d1;
          } else {
            split16_pair(y[4 * (m - 8)], y[4 * (m - 8) + 1], ys, 2 * (m - 8));
            split16_pair(y[4 * (m - 8) + 2], y[4 * (m - 8) + 3], ys, 2 * (m - 8) + 1);
          }
          __builtin_amdgcn_sched_barrier(0);
          if (m < 6) acc1n = mfma6_term(wa[0], xb[0], acc1n, m);
          else acc1n = mfma6_term(wa[1], xb[1], acc1n, m - 6);
          __builtin_amdgcn_sched_barrier(0);
        }
        ps_pin(ys);
        ps_pin(gg);
        asm volatile("" : "+v"(acc1n));
        PSF_TRACE(tslot + 2);
        // hand-over 1: the dY terms of tile-unit i (split one iteration ago)
        if ((int32_t)(__builtin_amdgcn_readfirstlane(d_free_early) - n) < 0) ps_wait_ge(F + kFDFree + pr, n);
        if (NDY == 8) {
          unsigned char* dp = DP + 32 * c + 16 * half;
          *reinterpret_cast<bf16x8*>(dp) = dy3[0].t1;
          *reinterpret_cast<bf16x8*>(dp + kDyPlane) = dy3[0].t2;
          *reinterpret_cast<bf16x8*>(dp + 2 * kDyPlane) = dy3[0].t3;
        } else {
#pragma unroll
          for (int s = 0; s < NS2; ++s) {
            *reinterpret_cast<bf16x8*>(DP + L.row[s]) = dy3[s].t1;
            *reinterpret_cast<bf16x8*>(DP + kDyPlane + L.row[s]) = dy3[s].t2;
            *reinterpret_cast<bf16x8*>(DP + 2 * kDyPlane + L.row[s]) = dy3[s].t3;
          }
        }
        ps_signal(F + kFDReady + pr, n + 1u);
        PSF_TRACE(tslot + 3);
        // hand-over 2: Hpost^T
        if ((int32_t)(__builtin_amdgcn_readfirstlane(s_free_early) - 2u * n) < 0) ps_wait_ge(F + kFSFree + pr, 2u * n);
        store_acc_plane(SP, L, ys, 0);
        store_acc_plane(SP + kPlaneBytes, L, ys, 1);
        store_acc_plane(SP + 2 * kPlaneBytes, L, ys, 2);
        ps_signal(F + kFSReady + pr, 2u * n + 1u);
        PSF_TRACE(tslot + 4);
        // A2: split of the next tile-unit's dY and of G  ||  step 2 of tile-unit i + 1
        const uint32_t s_free_early2 = F[kFSFree + pr];
        Frag3 bt[NS2];
#pragma unroll
        for (int s = 0; s < NS2; ++s) bt[s] = load_frag3(imgn + kPOffBT + ((s * 2 + half) * 32 + c) * 16, kXFragTerm);
        if (i % TPW == 1) PSF_TRACE(16);
        dy_split(i + 1 < NI ? i + 1 : NI - 1, dy3);
        if (i % TPW == 1) PSF_TRACE(17);
        dy_fetch(tp2);
        if (i % TPW == 1) PSF_TRACE(18);
        f32x16 acc3n;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc3n[r] = 0.f;
        Split16 gs;
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 8; ++m) {  // one matrix instruction of step 2, then one pair of G split
          if (m < 6 * NS2 && m < 6) acc3n = mfma6_term(bt[0], dy3[0], acc3n, m);
          else if (NS2 == 2 && m < 8) acc3n = mfma6_term(bt[NS2 - 1], dy3[NS2 - 1], acc3n, m - 6);
          __builtin_amdgcn_sched_barrier(0);
          split16_pair(gg[2 * m], gg[2 * m + 1], gs, m);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (NS2 == 2) {
#pragma unroll
          for (int k = 2; k < 6; ++k) acc3n = mfma6_term(bt[NS2 - 1], dy3[NS2 - 1], acc3n, k);
        }
        ps_pin(gs);
        asm volatile("" : "+v"(acc3n));
        PSF_TRACE(tslot + 5);
        if (next && tpn == TPW - 1) ps_signal(F + kFUnitDone + pr, g + (uint32_t)un + 1u);  // image reads of that unit: all issued
        // hand-over 3: G
        if ((int32_t)(__builtin_amdgcn_readfirstlane(s_free_early2) - (2u * n + 1u)) < 0) ps_wait_ge(F + kFSFree + pr, 2u * n + 1u);
        PSF_TRACE(tslot + 6);
        store_acc_plane(SP, L, gs, 0);
        store_acc_plane(SP + kPlaneBytes, L, gs, 1);
        store_acc_plane(SP + 2 * kPlaneBytes, L, gs, 2);
        ps_signal(F + kFSReady + pr, 2u * n + 2u);
        acc1 = acc1n;
        acc3 = acc3n;
      }
      g += (uint32_t)U;
    } else {
      // ---------------------------------------------------------------------------------------------- consumer
      // quarter pr of image `unit` -> buffer `buf` by LDS-DMA (inline assembly: see mlp_bwd_x3p_k)
      auto stage = [&](int unit, uint32_t buf) {
        const unsigned char* src = images + (size_t)unit * kXImgBytes;
        unsigned char* dst = img_lds + buf * kPImgBytes;
        for (int v0 = 0; v0 < kPImgVecs; v0 += 64 * kPsPairs) {
          const int v = v0 + 64 * pr + lane;
          const uint32_t lds_at = __builtin_amdgcn_readfirstlane(
              (uint32_t)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)(dst + 16 * (v0 + 64 * pr))));
          uint32_t m0_saved;
          if (v < kPImgVecs)
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(m0_saved) : "s"(lds_at), "v"(src + 16 * v) : "memory");
        }
      };
      float* part = a.partials + (blk * (int64_t)U) * kPart;
      float* SCR = reinterpret_cast<float*>(SP);  // the pair's two slots as one f32 area for the combine
      const float* SCR0 = reinterpret_cast<const float*>(pair_lds + kXBytes);
      const bf16x8 ones = __builtin_bit_cast(bf16x8, make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u));
      int dtr[2];  // transposed read of a 32-byte-row dY plane: rows 8 half + 4 t + q, columns 4 p .. 4 p + 3 (NDY == 8)
#pragma unroll
      for (int t = 0; t < 2; ++t) dtr[t] = 32 * (8 * half + 4 * t + ((lane >> 2) & 3)) + 8 * (lane & 3);
      f32x16 dxa[TPW];
#pragma unroll
      for (int tp = 0; tp < TPW; ++tp)
#pragma unroll
        for (int r = 0; r < 16; ++r) dxa[tp][r] = 0.f;
      for (int u = 0; u < U; ++u, ++g) {
#ifdef PSF_X3P_TRACE
        const bool trace_on = blockIdx.x == 300 && u == 5;
#endif
        PSF_TRACE(30);
        if (g == 0) {
          stage(0, 0u);
          __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0)
          ps_signal(F + kFImgReady + pr, 1u);
        }
        const bool has_next = u + 1 < U || !last_block;
        if (has_next) {
          ps_wait4_ge(F + kFUnitDone, g);  // every producer is done with unit g - 1, whose image buffer this one takes
          stage(u + 1 < U ? u + 1 : 0, (g + 1u) & 1u);
        }
        const unsigned char* img = img_lds + (g & 1u) * kPImgBytes;
        const bool first_block_of_mlp = (a.unit[u] >> 8) == 0;
        f32x16 dA, dBT, dav, dbv;
#pragma unroll
        for (int r = 0; r < 16; ++r) dA[r] = dBT[r] = dav[r] = dbv[r] = 0.f;
#pragma unroll
        for (int tp = 0; tp < TPW; ++tp, ++n) {
          const unsigned char* xp = XP + tp * 3 * kPlaneBytes;
          Frag3 yb[2], ha[2], ga[2], gb[2];
          if (tp < 2) PSF_TRACE(8 * tp + 0);
          ps_wait_ge(F + kFDReady + pr, n + 1u);
          if (tp < 2) PSF_TRACE(8 * tp + 1);
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            if (NDY == 8) {
              yb[s].t1 = join8(tr_read(DP + dtr[0] + 512 * s), tr_read(DP + dtr[1] + 512 * s));
              yb[s].t2 = join8(tr_read(DP + kDyPlane + dtr[0] + 512 * s), tr_read(DP + kDyPlane + dtr[1] + 512 * s));
              yb[s].t3 = join8(tr_read(DP + 2 * kDyPlane + dtr[0] + 512 * s), tr_read(DP + 2 * kDyPlane + dtr[1] + 512 * s));
            } else {
              yb[s].t1 = tr_frag(DP, L, s), yb[s].t2 = tr_frag(DP + kDyPlane, L, s), yb[s].t3 = tr_frag(DP + 2 * kDyPlane, L, s);
            }
          }
          if (tp + 1 < TPW) ps_signal(F + kFDFree + pr, n + 1u);  // (a unit's last tile: after the combine)
          ps_wait_ge(F + kFSReady + pr, 2u * n + 1u);
#pragma unroll
          for (int s = 0; s < 2; ++s)
            ha[s].t1 = tr_frag(SP, L, s), ha[s].t2 = tr_frag(SP + kPlaneBytes, L, s), ha[s].t3 = tr_frag(SP + 2 * kPlaneBytes, L, s);
          ps_signal(F + kFSFree + pr, 2u * n + 1u);
          if (tp < 2) PSF_TRACE(8 * tp + 2);
          // 4. dB^T[j][o] += Hpost^T[j][tok] dY[tok][o];  db[o] += 1 dY[tok][o]
#pragma unroll
          for (int s = 0; s < 2; ++s) dBT = mfma6(ha[s], yb[s], dBT);
          if (first_block_of_mlp) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
              dbv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, yb[s].t3, dbv, 0, 0, 0);
              dbv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, yb[s].t2, dbv, 0, 0, 0);
              dbv = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ones, yb[s].t1, dbv, 0, 0, 0);
            }
          }
          if (tp == 0 && has_next) {
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's quarter of image g + 1 has landed
            ps_signal(F + kFImgReady + pr, g + 2u);
          }
          if (tp < 2) PSF_TRACE(8 * tp + 3);
          ps_wait_ge(F + kFSReady + pr, 2u * n + 2u);
          if (tp < 2) PSF_TRACE(8 * tp + 4);
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            ga[s].t1 = tr_frag(SP, L, s), ga[s].t2 = tr_frag(SP + kPlaneBytes, L, s), ga[s].t3 = tr_frag(SP + 2 * kPlaneBytes, L, s);
            gb[s].t1 = row_frag(SP, L, s), gb[s].t2 = row_frag(SP + kPlaneBytes, L, s), gb[s].t3 = row_frag(SP + 2 * kPlaneBytes, L, s);
          }
          if (tp + 1 < TPW) ps_signal(F + kFSFree + pr, 2u * n + 2u);
          // 5. dA[j][e] += G[j][tok] X[tok][e],  da[j] += G[j][tok] 1
          // 6. dX^T[e][tok] += A_u^T[e][j] G[j][tok]   (A^T by transposed reads of the image, G by row reads of its planes)
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            const Frag3 xb{tr_frag(xp, L, s), tr_frag(xp + kPlaneBytes, L, s), tr_frag(xp + 2 * kPlaneBytes, L, s)};
            dA = mfma6(ga[s], xb, dA);
            dav = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[s].t3, ones, dav, 0, 0, 0);
            dav = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[s].t2, ones, dav, 0, 0, 0);
            dav = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[s].t1, ones, dav, 0, 0, 0);
            if (a.dX) {
              const Frag3 at{tr_frag(img, L, s), tr_frag(img + kPlaneBytes, L, s), tr_frag(img + 2 * kPlaneBytes, L, s)};
              dxa[tp] = mfma6(at, gb[s], dxa[tp]);
            }
          }
        }
        // combine among the four consumers (fixed order 0..3) through the pairs' slots, one partial slot per workgroup
        PSF_TRACE(16);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int r = 0; r < 16; ++r) SCR[cd_row(r, half) * 32 + c] = dA[r];
#pragma unroll
        for (int r = 0; r < 16; ++r) SCR[1024 + cd_row(r, half) * 32 + c] = dBT[r];
        if (c == 0) {  // every column of dav is da
#pragma unroll
          for (int r = 0; r < 16; ++r) SCR[2048 + cd_row(r, half)] = dav[r];
        }
        if (half == 0) SCR[2080 + c] = dbv[0];  // every row of dbv is db
        asm volatile("" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add((__attribute__((address_space(3))) uint32_t*)flag_lds + kFCombArrive, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        PSF_TRACE(17);
        ps_wait_ge(F + kFCombArrive, 4u * (g + 1u));
        PSF_TRACE(18);
        float* pu = part + (int64_t)u * kPart;
        auto sum4 = [&](int off) {  // off: float offset into a consumer's area, 16-byte aligned
          float4 acc = *reinterpret_cast<const float4*>(SCR0 + off);
#pragma unroll
          for (int w = 1; w < kPsPairs; ++w) {
            const float4 v = *reinterpret_cast<const float4*>(SCR0 + w * (kPairBytes / 4) + off);
            acc.x += v.x;
            acc.y += v.y;
            acc.z += v.z;
            acc.w += v.w;
          }
          return acc;
        };
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const int at = 512 * pr + 256 * i + 4 * lane;
          *reinterpret_cast<float4*>(pu + at) = sum4(at);
        }
        if (lane < 8) {  // da [2][32] | db [2][32]: the whole sums in row 0
          const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
          if (pr == 0) *reinterpret_cast<float4*>(pu + 2048 + 4 * lane) = sum4(2048 + 4 * lane);
          else if (pr == 1) *reinterpret_cast<float4*>(pu + 2112 + 4 * lane) = sum4(2080 + 4 * lane);
          else if (pr == 2) *reinterpret_cast<float4*>(pu + 2080 + 4 * lane) = zero;
          else *reinterpret_cast<float4*>(pu + 2144 + 4 * lane) = zero;
        }
        asm volatile("" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add((__attribute__((address_space(3))) uint32_t*)flag_lds + kFCombDone, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        PSF_TRACE(19);
        ps_wait_ge(F + kFCombDone, 4u * (g + 1u));
        PSF_TRACE(20);
        ps_signal(F + kFDFree + pr, n);        // the slots go back to the producer
        ps_signal(F + kFSFree + pr, 2u * n);
      }
      if (a.dX) {  // the lane holds dX^T[e = 8 gq + 4 half + (0..3)][tok = c] in registers 4 gq .. 4 gq + 3
#pragma unroll
        for (int tp = 0; tp < TPW; ++tp) {
          const int64_t tok = (tile0 + tp) * 32 + c;
          if (tok < a.T) {
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
              const int e0 = 8 * gq + 4 * half;
              if (e0 < E)
                *reinterpret_cast<float4*>(a.dX + tok * E + e0) =
                    make_float4(dxa[tp][4 * gq], dxa[tp][4 * gq + 1], dxa[tp][4 * gq + 2], dxa[tp][4 * gq + 3]);
            }
          }
        }
      }
    }
  }
}

// stage 1: R1[s][i] = sum over the groups of slice s of P[g][i]   (i < U*kPart; fixed order: four interleaved running sums
// g = g0 + 0, 1, 2, 3 (mod 4), then ((s0 + s1) + s2) + s3). float4 per thread and four loads in flight per running sum: the
// rolled scalar loop it replaces had one dependent load in flight per thread (49 us for 167 MB at Order N=16384, B=40).
__global__ void __launch_bounds__(256) mlp_bwd_reduce1_k(const BwdArgs a) {
  const int64_t n4 = (int64_t)a.U * kPart / 4;  // kPart % 4 == 0
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const int s = blockIdx.y;
  const int64_t per = (a.G + kSlices - 1) / kSlices;
  const int64_t g0 = s * per, g1 = g0 + per < a.G ? g0 + per : a.G;
  const float4* __restrict__ P = reinterpret_cast<const float4*>(a.partials);
  float4 acc[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
  int64_t g = g0;
  for (; g + 4 <= g1; g += 4) {
    float4 v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = P[(g + q) * n4 + i];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      acc[q].x += v[q].x;
      acc[q].y += v[q].y;
      acc[q].z += v[q].z;
      acc[q].w += v[q].w;
    }
  }
  for (int q = 0; g < g1; ++g, ++q) {
    const float4 v = P[g * n4 + i];
    acc[q].x += v.x;
    acc[q].y += v.y;
    acc[q].z += v.z;
    acc[q].w += v.w;
  }
  float4 r;
  r.x = ((acc[0].x + acc[1].x) + acc[2].x) + acc[3].x;
  r.y = ((acc[0].y + acc[1].y) + acc[2].y) + acc[3].y;
  r.z = ((acc[0].z + acc[1].z) + acc[2].z) + acc[3].z;
  r.w = ((acc[0].w + acc[1].w) + acc[2].w) + acc[3].w;
  reinterpret_cast<float4*>(a.stage1)[s * n4 + i] = r;
}

// stage 2: sum the slices and scatter into the unpadded gradient tensors
__global__ void __launch_bounds__(256) mlp_bwd_reduce2_k(const BwdArgs a) {
  const int64_t n = (int64_t)a.U * kPart;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n) return;
  const int u = (int)(idx / kPart), i = (int)(idx - (int64_t)u * kPart);
  const BwdMlp& d = a.m[a.unit[u] & 0xff];
  const int ht = 32 * (int)(a.unit[u] >> 8), E = a.E;
  auto total = [&](int64_t at) {
    float acc = 0.f;
    for (int s = 0; s < kSlices; ++s) acc += a.stage1[s * n + at];
    return acc;
  };
  if (i < 1024) {
    const int j = i >> 5, e = i & 31;
    if (ht + j < d.h && e < E) d.dA[(ht + j) * E + e] = total(idx);
  } else if (i < 2048) {
    const int j = (i - 1024) >> 5, o = i & 31;
    if (ht + j < d.h && o < d.O) d.dB[o * d.h + ht + j] = total(idx);
  } else if (i < 2048 + 32) {
    const int j = i - 2048;
    if (ht + j < d.h) d.da[ht + j] = total(idx) + total(idx + 32);
  } else if (i >= 2112 && i < 2112 + 32) {
    const int o = i - 2112;
    if (ht == 0 && o < d.O) d.db[o] = total(idx) + total(idx + 32);
  }
}

struct Plan {
  int U;
  int tpw;  // tiles per wave: 2, or 1 for short inputs (keeps >= 2 workgroups per CU in flight)
  int tpw8;
  int64_t G, G8;
  uint8_t unit_k[kMaxUnits], unit_hb[kMaxUnits];
};

bool make_plan(int64_t T, int32_t E, int32_t K, const int32_t* h, const int32_t* O, Plan* p) {
  if (T < 1 || E < 4 || E > 32 || (E & 3) || K < 1 || K > kMaxMlps || !h || !O) return false;
  p->U = 0;
  for (int k = 0; k < K; ++k) {
    if (h[k] < 1 || h[k] > 128 || O[k] < 1 || O[k] > 32) return false;
    for (int hb = 0; hb * 32 < h[k]; ++hb) {
      p->unit_k[p->U] = (uint8_t)k;
      p->unit_hb[p->U] = (uint8_t)hb;
      ++p->U;
    }
  }
  const int64_t tiles = (T + 31) / 32;
  p->tpw = (tiles + 7) / 8 >= 512 ? 2 : 1;
  // one workgroup (= one partial slot) per waves*TPW tiles: 4 waves (f32 kernel) or 8 (split-bf16 kernel)
  p->G = (tiles + 4 * p->tpw - 1) / (4 * p->tpw);
  int max_o = 1;
  for (int k = 0; k < K; ++k) max_o = O[k] > max_o ? O[k] : max_o;
  // 512-thread kernels: outputs wider than 16 double the dY registers of a lane and two tiles per wave no longer fit 256
  // registers (70 spilled, 0.417 ms at 4 x O = 32, T = 655 k); one tile per wave fits: 0.397 ms.
  p->tpw8 = max_o > 16 ? 1 : p->tpw;
  p->G8 = (tiles + 8 * p->tpw8 - 1) / (8 * p->tpw8);
  return true;
}

// sized for either kernel variant (the knob may change between the workspace query and the call)
int64_t workspace_floats(const Plan& p) {
  const int64_t slots = p.G > p.G8 ? p.G : p.G8;
  return (int64_t)p.U * kImgFloatsMax + slots * p.U * kPart + (int64_t)kSlices * p.U * kPart;
}

}  // namespace

extern "C" {

int64_t psf_mlp_bwd_workspace(int64_t T, int32_t E, int32_t K, const int32_t* h, const int32_t* O) {
  Plan p;
  if (!make_plan(T, E, K, h, O, &p)) return -1;
  return workspace_floats(p) * (int64_t)sizeof(float);
}

int psf_mlp_bwd_f32(const float* X, int64_t T, int32_t E, int32_t K, const float* const* A, const float* const* a,
                    const float* const* B, const int32_t* h, const int32_t* O, const float* const* dY, float* dX,
                    float* const* dA, float* const* da, float* const* dB, float* const* db, void* workspace,
                    int64_t workspace_bytes, void* stream) {
  if (!X || !A || !a || !B || !h || !O || !dY || !dA || !da || !dB || !db || !workspace)
    return psf_internal_fail(PSF_E_NULL, "psf_mlp_bwd: NULL argument");
  Plan p;
  if (!make_plan(T, E, K, h, O, &p))
    return psf_internal_fail(PSF_E_SHAPE, "psf_mlp_bwd: need T >= 1, E in {4,8,...,32}, 1 <= K <= 32, 1 <= h <= 128, 1 <= O <= 32");
  if ((reinterpret_cast<uintptr_t>(X) & 15) != 0) return psf_internal_fail(PSF_E_ALIGN, "psf_mlp_bwd: X must be 16-byte aligned");
  if (workspace_bytes < workspace_floats(p) * (int64_t)sizeof(float) || (reinterpret_cast<uintptr_t>(workspace) & 15) != 0)
    return psf_internal_fail(PSF_E_SHAPE, "psf_mlp_bwd: workspace too small (psf_mlp_bwd_workspace) or not 16-byte aligned");
  BwdArgs args;
  for (int k = 0; k < kMaxMlps; ++k) args.m[k] = BwdMlp{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0};
  for (int k = 0; k < K; ++k) {
    if (!A[k] || !a[k] || !B[k] || !dY[k] || !dA[k] || !da[k] || !dB[k] || !db[k])
      return psf_internal_fail(PSF_E_NULL, "psf_mlp_bwd: NULL layer pointer");
    args.m[k] = BwdMlp{A[k], a[k], B[k], dY[k], dA[k], da[k], dB[k], db[k], h[k], O[k]};
  }
  for (int u = 0; u < kMaxUnits; ++u) {
    args.unit[u] = u < p.U ? ((uint32_t)p.unit_k[u] | ((uint32_t)p.unit_hb[u] << 8)) : 0u;
  }
  const int variant = psf_g_mlp_bwd_variant.load();
  const bool x3 = variant == 2;                      // the first split-bf16 form (steps 1, 2, 6 on bf16)
  int max_o = 1;
  for (int k = 0; k < K; ++k) max_o = O[k] > max_o ? O[k] : max_o;
  const bool roles = variant == 4;                   // producer / consumer waves (mlp_bwd_ps_k)
  const bool planes = variant == 0 || variant == 3 || roles;  // split-bf16 on dual-use LDS planes (the default)
  const int ps_tiles = kPsPairs * (max_o <= 16 ? 4 : 3);      // tiles per workgroup of mlp_bwd_ps_k
  if (planes && dX && (reinterpret_cast<uintptr_t>(dX) & 15) != 0)
    return psf_internal_fail(PSF_E_ALIGN, "psf_mlp_bwd: dX must be 16-byte aligned");
  float* ws = reinterpret_cast<float*>(workspace);
  args.X = X;
  args.dX = dX;
  args.images = ws;
  args.partials = ws + (int64_t)p.U * kImgFloatsMax;
  args.stage1 = args.partials + (p.G > p.G8 ? p.G : p.G8) * p.U * kPart;
  args.T = T;
  args.G = roles ? ((T + 31) / 32 + ps_tiles - 1) / ps_tiles : (x3 || planes) ? p.G8 : p.G;
  args.E = E;
  args.K = K;
  args.U = p.U;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  if (planes) hipLaunchKernelGGL(mlp_bwd_x3p_pack_k, dim3(p.U), dim3(256), 0, s, args);
  else if (x3) hipLaunchKernelGGL(mlp_bwd_x3_pack_k, dim3(p.U), dim3(256), 0, s, args);
  else hipLaunchKernelGGL(mlp_bwd_pack_k, dim3(p.U), dim3(256), 0, s, args);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return psf_internal_fail((int)e, hipGetErrorString(e));

  // every slot of the partial buffer is written exactly once: one workgroup per waves*TPW tiles
  const int64_t blocks = args.G;
  if (blocks > 0x7fffffff) return psf_internal_fail(PSF_E_SHAPE, "psf_mlp_bwd: T too large");
  auto launch = [&](auto kernel, int waves, size_t img_bytes, int tpw, int sd, int img_bufs) {
    const size_t lds_bytes = (size_t)img_bufs * img_bytes + sizeof(float) * (size_t)waves * (tpw * 32 * 33 + 32 * sd + 32 * 33);
    e = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes);
    if (e == hipSuccess) hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(64 * waves), lds_bytes, s, args);
  };
  auto launch_planes = [&](auto kernel, int waves) {  // static LDS: 2 images + 8 x (tpw x 3 planes + scratch) = 160,000 B at tpw = 2
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(64 * waves), 0, s, args);
  };
  if (roles) {
    if (max_o <= 16) launch_planes(mlp_bwd_ps_k<4, 8>, 8);
    else launch_planes(mlp_bwd_ps_k<3, 16>, 8);
  } else if (planes) {
    if (max_o <= 16) {
      if (p.tpw8 == 2) launch_planes(mlp_bwd_x3p_k<2, 8>, 8);
      else launch_planes(mlp_bwd_x3p_k<1, 8>, 8);
    } else {
      launch_planes(mlp_bwd_x3p_k<1, 16>, 8);  // make_plan: one tile per wave when an output is wider than 16
    }
  } else if (x3) {
    if (max_o <= 16) {
      if (p.tpw8 == 2) launch(mlp_bwd_x3_k<2, 8>, 8, kXImgBytes, 2, 17, 2);
      else launch(mlp_bwd_x3_k<1, 8>, 8, kXImgBytes, 1, 17, 2);
    } else {
      launch(mlp_bwd_x3_k<1, 16>, 8, kXImgBytes, 1, 33, 1);  // make_plan: one tile per wave when an output is wider than 16
    }
  } else if (max_o <= 16) {
    if (p.tpw == 2) launch(mlp_bwd_k<2, 8>, 4, kImg * sizeof(float), 2, 17, 2);
    else launch(mlp_bwd_k<1, 8>, 4, kImg * sizeof(float), 1, 17, 2);
  } else {
    if (p.tpw == 2) launch(mlp_bwd_k<2, 16>, 4, kImg * sizeof(float), 2, 33, 1);
    else launch(mlp_bwd_k<1, 16>, 4, kImg * sizeof(float), 1, 33, 1);
  }
  if (e != hipSuccess) return psf_internal_fail((int)e, hipGetErrorString(e));
  const int64_t n = (int64_t)p.U * kPart;
  hipLaunchKernelGGL(mlp_bwd_reduce1_k, dim3((unsigned)((n / 4 + 255) / 256), kSlices), dim3(256), 0, s, args);
  hipLaunchKernelGGL(mlp_bwd_reduce2_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, args);
  e = hipGetLastError();
  return e == hipSuccess ? PSF_OK : psf_internal_fail((int)e, hipGetErrorString(e));
}

}  // extern "C"
