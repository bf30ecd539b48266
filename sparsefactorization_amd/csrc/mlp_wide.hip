// mlp_wide.hip — the producer MLPs of PSFNet at the WIDE (LRA) sizes: E up to 512, hidden <= 128, outputs up to 128.
//
//   Y_k = GELU(X A_k^T + a_k) B_k^T + b_k,  k < K        (MLPBlock, LRA/psf.py:35-60; PSFNet applies g and fs[0..M) to the
//   same `data`, LRA/psf.py:214,227; reference ListOps config E = 512, h = 128, outputs 12 x 11 and 128,
//   LRA/psf_training_config.py:2-30)
//
// The narrow kernels (mlp_fwd_x3.hip, mlp_bwd.hip: E <= 32) keep every weight of a 32-row unit in LDS and never let a
// hidden activation reach memory. At E = 512 the first layers ARE the step: three GEMMs of 100 GFLOP each (forward,
// input gradient, weight gradient), which PyTorch runs as f32 hipBLASLt GEMMs at 87-140 TFLOP/s (60 % of the ListOps
// training step, profiles/r02as_lra_step_profile_listops.log). Here they run on the bf16 matrix pipe at f32 accuracy
// (x3_gemm.h: every operand split exactly into three bf16 terms, six product terms) and the M + 1 first layers are ONE
// stacked GEMM with J = sum of the (padded) hidden widths:
//
//   forward   split X -> col16 planes XP                                   x3_split_planes_k
//             Hpre^T[j][tok] = Wcat XP^T + acat   (NT GEMM, K = E)         x3_gemm_k<false>, fragment-order output HF
//             Y_k = GELU(Hpre_k) B_k^T + b_k                                wide_out_k   (accumulator-as-operand, as the
//                                                                           narrow forward's second GEMM)
//   backward  dHpost = dY_k B_k; G = dHpost .* GELU'(Hpre) -> planes GP;    wide_mid_k   (the narrow backward's steps 2-4
//             dB_k, da_k, db_k partial sums                                 with Hpre LOADED instead of recomputed)
//             dX = G Wcat            (NT GEMM, K = J)                       x3_gemm_k<false>
//             dAcat = G^T X          (TN GEMM over the tokens, split-K)     x3_gemm_k<true>  + fixed-order reduction
//
// What the forward leaves for the backward ("saved", caller-owned): XP (6 bytes per element of X) and HF (Hpre in the
// accumulator-fragment order every consumer wants: registers 4 q .. 4 q + 3 of lane l of the 32 x 32 tile (hidden unit,
// token tile) as one 16-byte vector at (((tile_t * U + unit) * 4 + q) * 64 + l) * 4 floats — every access of producer
// and consumers is a 1 KB dwordx4 burst).
// Nothing is recomputed: at these widths recomputing Hpre would be a fourth 100-GFLOP GEMM, keeping it is 4 J bytes
// per token of HBM traffic. All reductions run in a fixed order (no float atomics): bit-reproducible.
//
// Limits: E a multiple of 16, 16 <= E <= 1024; 1 <= h <= 128; 1 <= O <= 128; K <= 24; every plane below 2 GiB.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "../../include/psf_chord.h"
#include "mlp_planes.h"
#include "mlp_x3_common.h"
#include "x3_gemm.h"

extern "C" int psf_internal_fail(int code, const char* message);
extern std::atomic<int> psf_g_wide_fuse;  // psf_chord.hip: tuning knob "wide_fuse"

namespace {

using namespace psf_wide;

constexpr int kMaxWideMlps = 24;
constexpr int kMaxWideUnits = 96;      // 24 MLPs x 4 units of 32 hidden rows
constexpr int kOtMax = kOutTilesMax;    // output tiles of 32 per MLP (O <= 128)
constexpr int kPackBtStep = 1024;      // one term of one (unit, k-step of 16 outputs): [2 half][32 j][8] bf16
constexpr int kSplitsMax = 64;
constexpr int kSlotClass = 4 * 96;           // work-item table of wide_mid_k: three classes x (<= 4 sub-items per unit)
constexpr int kSlotBytes = 3 * kSlotClass * 4;

// ---- y = GELU(x) = x Phi(x), dy/dx = Phi(x) + x phi(x); Phi by Abramowitz & Stegun 26.2.17 (|error| <= 7.5e-8), on
// scalar f32 instructions (packed f32 VALU is slow beside the sibling wave's MFMAs: mlp_bwd.hip, gelu_and_grad1).
__device__ __forceinline__ void gelu_and_grad(float x, float& y, float& dydx) {
  const float t = __builtin_amdgcn_rcpf(fmaf(fabsf(x), 0.2316419f, 1.0f));
  float p = fmaf(0.53070271f, t, -0.72657602f);
  p = fmaf(p, t, 0.71070687f);
  p = fmaf(p, t, -0.14224837f);
  p = fmaf(p, t, 0.12741479f);
  p = p * t;
  const float E = __builtin_amdgcn_exp2f((x * x) * -0.72134752044448170368f);
  const float dlt = copysignf(0.5f - p * E, x);
  const float Phi = 0.5f + dlt;
  y = x * Phi;
  dydx = fmaf(x * 0.39894228040143267794f, E, Phi);
}
__device__ __forceinline__ Frag3 load_frag3(const unsigned char* p, int term_stride) {
  return Frag3{*reinterpret_cast<const bf16x8*>(p), *reinterpret_cast<const bf16x8*>(p + term_stride),
               *reinterpret_cast<const bf16x8*>(p + 2 * term_stride)};
}

// ---------------------------------------------------------------------------------------------------------------
// geometry shared by host and kernels
// ---------------------------------------------------------------------------------------------------------------
struct WideGeom {
  int64_t T, T_pad;       // tokens, padded to the GEMM tile (256)
  int32_t E, E_pad;       // input width, padded to 256 (only the dX / dA tiles see the padding)
  int32_t J, J_pad;       // stacked hidden rows (each MLP padded to a multiple of 32), padded to 256
  int32_t K, U;           // MLPs; 32-row units in J_pad
  int32_t unit_k[kMaxWideUnits];  // unit -> MLP (units past J: -1)
};

struct WideFwdMlp {
  const float* A;  // [h, E]
  const float* a;  // [h]
  const float* B;  // [O, h]
  const float* b;  // [O]
  float* Y;        // [T, O]
  int32_t h, O, joff, ot;  // joff: first stacked row; ot = ceil(O / 32)
};

struct WideFwdArgs {
  WideFwdMlp m[kMaxWideMlps];
  WideGeom g;
  unsigned char* w1p[3];   // Wcat planes: rows j (J_pad), column blocks e
  float* acat;             // [J_pad]
  unsigned char* pack2;    // [U][kOtMax][3 terms][kPack2Tile]
  float* b2;               // [K][kOtMax][2 half][16]: b_k[32 ot + cd_row(r, half)]
  const float* HF;         // Hpre fragments
  uint32_t skip_mask;      // MLPs whose second layer ran in the forward GEMM's epilogue (x3_gemm.h, FusedOut)
};

struct WideBwdMlp {
  const float* A;
  const float* B;
  const float* dY;  // [T, O]
  float* dA;
  float* da;
  float* dB;
  float* db;
  int32_t h, O, joff, ot;
};

struct WideBwdArgs {
  WideBwdMlp m[kMaxWideMlps];
  WideGeom g;
  unsigned char* w1tp[3];  // Wcat^T planes: rows e (E_pad), column blocks j (J / 16)
  unsigned char* packbt;   // [U][8 k-steps][3 terms][kPackBtStep]
  unsigned char* gp[3];    // G planes: rows tok (T_pad), column blocks j
  const float* HF;
  float* part;             // [groups][rec_total] partial sums of dB^T, da, db
  float* red;              // [kRedSlices][rec_total] first stage of their reduction
  const float* dapart;     // [splits][J_pad][E_pad] partial dAcat
  // float offset of a unit's first record in a group's slab; a unit has nsub(unit) records of rec_size(unit) floats:
  // [ot][1024] dB^T | [32] da | [ot][32] db (first unit of its MLP)
  uint32_t rec_off[kMaxWideUnits];
  // work items of wide_mid_k by unit class (0: O <= 16, 1: O <= 32, 2: wider), each class kSlotClass entries:
  // unit | sub << 8 | nsub << 16 (written by the pack kernel)
  uint32_t* slots;
  int32_t rec_total, groups, splits, n_slots[3];
};
__host__ __device__ inline int wide_class(int O) { return O <= 16 ? 0 : O <= 32 ? 1 : 2; }

// A unit of a wide-output MLP (more than 32 outputs: 2-4 output tiles) costs 2-3 x a narrow one per token tile in
// wide_mid_k; it is cut into nsub work items of 8 / nsub token tiles so that items cost about the same.
__host__ __device__ inline int wide_nsub(int ot) { return ot <= 1 ? 1 : ot == 2 ? 2 : 4; }
__host__ __device__ inline int wide_rec_size(int ot, bool first) { return ot * 1024 + 32 + (first ? ot * 32 : 0); }

// ---------------------------------------------------------------------------------------------------------------
// packing (once per call; all of it a few MB)
// ---------------------------------------------------------------------------------------------------------------
// forward: Wcat planes, acat, second-layer operand tiles, b2
__global__ void __launch_bounds__(256) wide_pack_fwd_k(const WideFwdArgs a) {
  const WideGeom& g = a.g;
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x, gsz = (int64_t)gridDim.x * 256;
  // (1) Wcat planes: one thread per (row j, column block eb): 16 values -> 3 x 32 bytes
  const int eb_n = g.E / 16;
  for (int64_t i = gid; i < (int64_t)g.J_pad * eb_n; i += gsz) {
    const int eb = (int)(i / g.J_pad), j = (int)(i - (int64_t)eb * g.J_pad);
    const int k = j < g.J ? g.unit_k[j >> 5] : -1;
    float v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = 0.f;
    if (k >= 0) {
      const int jj = j - a.m[k].joff;
      if (jj < a.m[k].h) {
        const float4* src = reinterpret_cast<const float4*>(a.m[k].A + (int64_t)jj * g.E + 16 * eb);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 x = src[q];
          v[4 * q] = x.x, v[4 * q + 1] = x.y, v[4 * q + 2] = x.z, v[4 * q + 3] = x.w;
        }
      }
    }
    const float lo[8] = {v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]};
    const float hi[8] = {v[8], v[9], v[10], v[11], v[12], v[13], v[14], v[15]};
    const Frag3 f0 = split_pack8(lo), f1 = split_pack8(hi);
    const int64_t at = ((int64_t)eb * g.J_pad + j) * 32;
    *reinterpret_cast<bf16x8*>(a.w1p[0] + at) = f0.t1;
    *reinterpret_cast<bf16x8*>(a.w1p[0] + at + 16) = f1.t1;
    *reinterpret_cast<bf16x8*>(a.w1p[1] + at) = f0.t2;
    *reinterpret_cast<bf16x8*>(a.w1p[1] + at + 16) = f1.t2;
    *reinterpret_cast<bf16x8*>(a.w1p[2] + at) = f0.t3;
    *reinterpret_cast<bf16x8*>(a.w1p[2] + at + 16) = f1.t3;
  }
  // (2) acat
  for (int64_t j = gid; j < g.J_pad; j += gsz) {
    const int k = j < g.J ? g.unit_k[j >> 5] : -1;
    float v = 0.f;
    if (k >= 0 && j - a.m[k].joff < a.m[k].h) v = a.m[k].a[j - a.m[k].joff];
    a.acat[j] = v;
  }
  // (3) second layer, A operand of Y^T[o][tok] = sum_j B[o][j] Hpost^T[j][tok] with the contraction index in the
  // accumulator's register order: [unit][ot][term][s][half][o][i] = B_k[32 ot + o][hb + rho(s, half, i)]
  for (int64_t i = gid; i < (int64_t)(g.J / 32) * kOtMax * 2 * 2 * 32; i += gsz) {
    const int o = (int)(i & 31), hf = (int)((i >> 5) & 1), s = (int)((i >> 6) & 1), ot = (int)((i >> 7) & 3), unit = (int)(i >> 9);
    const int k = g.unit_k[unit];
    const WideFwdMlp& d = a.m[k];
    const int hb = 32 * unit - d.joff;
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int rho = (q & 3) + 16 * s + 8 * (q >> 2) + 4 * hf;
      const int oo = 32 * ot + o;
      v[q] = (oo < d.O && hb + rho < d.h) ? d.B[(int64_t)oo * d.h + hb + rho] : 0.f;
    }
    const Frag3 f = split_pack8(v);
    unsigned char* dst = a.pack2 + ((int64_t)(unit * kOtMax + ot) * 3) * kPack2Tile + ((s * 2 + hf) * 32 + o) * 16;
    *reinterpret_cast<bf16x8*>(dst) = f.t1;
    *reinterpret_cast<bf16x8*>(dst + kPack2Tile) = f.t2;
    *reinterpret_cast<bf16x8*>(dst + 2 * kPack2Tile) = f.t3;
  }
  // (4) b2
  for (int64_t i = gid; i < (int64_t)g.K * kOtMax * 32; i += gsz) {
    const int r = (int)(i & 15), hf = (int)((i >> 4) & 1), ot = (int)((i >> 5) & 3), k = (int)(i >> 7);
    const int o = 32 * ot + cd_row(r, hf);
    a.b2[i] = o < a.m[k].O ? a.m[k].b[o] : 0.f;
  }
}

// backward: Wcat^T planes and the B^T operand tiles of dHpost^T[j][tok] = sum_o B[o][j] dY[tok][o]
__global__ void __launch_bounds__(256) wide_pack_bwd_k(const WideBwdArgs a) {
  const WideGeom& g = a.g;
  const int64_t gid = (int64_t)blockIdx.x * 256 + threadIdx.x, gsz = (int64_t)gridDim.x * 256;
  if (gid < g.J / 32) {  // (0) the work items of wide_mid_k by unit class, in unit order: thread `unit` places its own
    const int unit = (int)gid;
    const WideBwdMlp& d = a.m[g.unit_k[unit]];
    const int cls = wide_class(d.O), nsub = wide_nsub(d.ot);
    int n = 0;
    for (int u2 = 0; u2 < unit; ++u2) {
      const WideBwdMlp& d2 = a.m[g.unit_k[u2]];
      if (wide_class(d2.O) == cls) n += wide_nsub(d2.ot);
    }
    for (int sub = 0; sub < nsub; ++sub) a.slots[cls * kSlotClass + n + sub] = (uint32_t)unit | ((uint32_t)sub << 8) | ((uint32_t)nsub << 16);
  }
  // (1) Wcat^T planes: rows e (E_pad), column blocks jb (J / 16): element (e, j) = Wcat[j][e]
  const int jb_n = g.J / 16;
  for (int64_t i = gid; i < (int64_t)g.E_pad * jb_n; i += gsz) {
    const int jb = (int)(i / g.E_pad), e = (int)(i - (int64_t)jb * g.E_pad);
    const int k = g.unit_k[jb >> 1];
    const WideBwdMlp& d = a.m[k];
    float v[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int jj = 16 * jb + q - d.joff;
      v[q] = (e < g.E && jj < d.h) ? d.A[(int64_t)jj * g.E + e] : 0.f;
    }
    const float lo[8] = {v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]};
    const float hi[8] = {v[8], v[9], v[10], v[11], v[12], v[13], v[14], v[15]};
    const Frag3 f0 = split_pack8(lo), f1 = split_pack8(hi);
    const int64_t at = ((int64_t)jb * g.E_pad + e) * 32;
    *reinterpret_cast<bf16x8*>(a.w1tp[0] + at) = f0.t1;
    *reinterpret_cast<bf16x8*>(a.w1tp[0] + at + 16) = f1.t1;
    *reinterpret_cast<bf16x8*>(a.w1tp[1] + at) = f0.t2;
    *reinterpret_cast<bf16x8*>(a.w1tp[1] + at + 16) = f1.t2;
    *reinterpret_cast<bf16x8*>(a.w1tp[2] + at) = f0.t3;
    *reinterpret_cast<bf16x8*>(a.w1tp[2] + at + 16) = f1.t3;
  }
  // (2) [unit][s2][term][half][j][i] = B_k[o = 16 s2 + 8 half + i][hb + j]
  for (int64_t i = gid; i < (int64_t)(g.J / 32) * 8 * 2 * 32; i += gsz) {
    const int j = (int)(i & 31), hf = (int)((i >> 5) & 1), s2 = (int)((i >> 6) & 7), unit = (int)(i >> 9);
    const int k = g.unit_k[unit];
    const WideBwdMlp& d = a.m[k];
    const int hb = 32 * unit - d.joff;
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int o = 16 * s2 + 8 * hf + q;
      v[q] = (o < d.O && hb + j < d.h) ? d.B[(int64_t)o * d.h + hb + j] : 0.f;
    }
    const Frag3 f = split_pack8(v);
    unsigned char* dst = a.packbt + ((int64_t)(unit * 8 + s2) * 3) * kPackBtStep + (hf * 32 + j) * 16;
    *reinterpret_cast<bf16x8*>(dst) = f.t1;
    *reinterpret_cast<bf16x8*>(dst + kPackBtStep) = f.t2;
    *reinterpret_cast<bf16x8*>(dst + 2 * kPackBtStep) = f.t3;
  }
}

// ---------------------------------------------------------------------------------------------------------------
// forward, second layer: Y_k^T[o][tok] = B_k GELU(Hpre_k^T) + b_k. One wave per 32-token tile, all MLPs; the loaded
// Hpre tile is already the B operand of the product (accumulator-as-operand), the A operand tiles come from L2.
// ---------------------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) wide_out_k(const WideFwdArgs a) {
  const WideGeom& g = a.g;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int c = lane & 31, half = lane >> 5;
  const int64_t ttile = (int64_t)blockIdx.x * 4 + wv;
  if (ttile * 32 >= g.T) return;  // wave-uniform
  const int64_t tok = ttile * 32 + c;
  const bool tok_ok = tok < g.T;
  // blockIdx.y deals the MLPs round-robin: with one wave per token tile for ALL MLPs a ListOps batch is 2000 waves, eight
  // per CU, and every Hpre load is exposed
  for (int k = blockIdx.y; k < g.K; k += gridDim.y) {
    if ((a.skip_mask >> k) & 1) continue;
    const WideFwdMlp& d = a.m[k];
    const int ot_n = d.ot, unit0 = d.joff >> 5, nu = (d.h + 31) >> 5;
    f32x16 acc2[kOtMax];
#pragma unroll
    for (int ot = 0; ot < kOtMax; ++ot) {
      if (ot < ot_n) {
        const float4* bp = reinterpret_cast<const float4*>(a.b2 + ((k * kOtMax + ot) * 2 + half) * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 v = bp[q];
          acc2[ot][4 * q] = v.x, acc2[ot][4 * q + 1] = v.y, acc2[ot][4 * q + 2] = v.z, acc2[ot][4 * q + 3] = v.w;
        }
      }
    }
    for (int u = 0; u < nu; ++u) {
      const int unit = unit0 + u;
      const float4* hf = reinterpret_cast<const float4*>(a.HF + (((ttile * g.U + unit) * 16) << 6)) + lane;
      float y[16];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = hf[q * 64];
        y[4 * q] = v.x, y[4 * q + 1] = v.y, y[4 * q + 2] = v.z, y[4 * q + 3] = v.w;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) y[r] = gelu_only(y[r]);
      const Split16 ys = split16(y);
#pragma unroll
      for (int ot = 0; ot < kOtMax; ++ot) {
        if (ot < ot_n) {
          const unsigned char* wp = a.pack2 + ((int64_t)(unit * kOtMax + ot) * 3) * kPack2Tile;
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            const Frag3 wb = load_frag3(wp + ((s * 2 + half) * 32 + c) * 16, kPack2Tile);
            const Frag3 hb{acc_frag(ys, 0, s), acc_frag(ys, 1, s), acc_frag(ys, 2, s)};
            acc2[ot] = mfma6(wb, hb, acc2[ot]);
          }
        }
      }
    }
    // the lane holds Y^T[o = 32 ot + 8 q + 4 half + (0..3)][tok] in registers 4 q .. 4 q + 3 of tile ot
    const int O = d.O;
    const bool vec_ok = (O & 3) == 0 && (reinterpret_cast<uintptr_t>(d.Y) & 15) == 0;
    if (tok_ok) {
#pragma unroll
      for (int ot = 0; ot < kOtMax; ++ot) {
        if (ot < ot_n) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int o0 = 32 * ot + 8 * q + 4 * half;
            float* yp = d.Y + tok * O + o0;
            if (vec_ok) {
              if (o0 < O) *reinterpret_cast<float4*>(yp) = make_float4(acc2[ot][4 * q], acc2[ot][4 * q + 1], acc2[ot][4 * q + 2], acc2[ot][4 * q + 3]);
            } else {
#pragma unroll
              for (int i = 0; i < 4; ++i)
                if (o0 + i < O) yp[i] = acc2[ot][4 * q + i];
            }
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// backward, middle: per 32-token tile and 32-row hidden unit
//   dHpost^T[j][tok] = sum_o B[o][j] dY[tok][o]          (B operand: the lane's own dY row, straight from memory)
//   G = dHpost^T .* GELU'(Hpre^T)  -> three bf16 terms -> col16 planes GP (the operand of both big backward GEMMs)
//   dB^T[j][o] += Hpost^T[j][tok] dY[tok][o]             (contraction over tokens: Hpost^T transposed through two LDS
//                                                          planes + ds_read_b64_tr_b16; dY columns straight from memory)
//   da[j] += sum_tok G,  db[o] += sum_tok dY
// Work item = (group of kMidTiles token tiles, hidden unit), owned by ONE wave: it walks the group's token tiles with the
// unit's dB^T / da / db sums in registers and writes the item's record once — no workgroup barrier, no cross-wave
// combine (a first version combined eight waves through LDS after every unit: 96 barriers per workgroup kept the waves
// in lockstep and every Hpre load exposed, 410 us at ListOps sizes). Items are dealt round-robin over all waves of the
// launch (unit fastest, so the four units of an MLP, which read the same dY tile, run side by side); the next token
// tile's Hpre registers are requested before the current tile is processed.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kMidTiles = 8;  // token tiles per item = one 256-token GEMM tile

template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float half_sum(float v) {  // sum over the 32 lanes of the lane's half of the wave
  v = dpp_add<0xB1>(v);
  v = dpp_add<0x4E>(v);
  v = dpp_add<0x141>(v);
  v = dpp_add<0x140>(v);
  return v + __shfl_xor(v, 16, 64);
}

// OT: output tiles of 32 a unit may have (dB^T accumulators); NDY: dY tile elements per lane (8: O <= 16). Three
// instances — <1, 8>, <1, 16>, <4, 16> — each launched on its own class of units (WideBwdArgs::n_slots): with the widest
// shape's 64 accumulator registers in every instance the common narrow case spilled 100+ registers.
template <int OT, int NDY>
__global__ void __launch_bounds__(512, 1) wide_mid_k(const WideBwdArgs a, const int cls) {
  constexpr int kWaveLds = 2 * kPlaneBytes + 32 * 33 * 4 + 128;  // two transposition planes + the dY tile (<= [32 tok][33])
  __shared__ __attribute__((aligned(16))) unsigned char wave_lds[8 * kWaveLds];
  const WideGeom& g = a.g;
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 31, half = lane >> 5;
  unsigned char* HP = wave_lds + wv * kWaveLds;
  unsigned char* YP = HP + kPlaneBytes;
  float* YT = reinterpret_cast<float*>(YP + kPlaneBytes);
  const PlaneLane L = plane_lane(lane);
  const int n_slots = a.n_slots[cls];
  const uint32_t* slots = a.slots + cls * kSlotClass;
  const int64_t items = (int64_t)a.groups * n_slots;
  const int64_t waves = (int64_t)gridDim.x * 8;

  for (int64_t item = (int64_t)blockIdx.x * 8 + wv; item < items; item += waves) {
    const int64_t grp = item / n_slots;
    const uint32_t slot = slots[item - grp * n_slots];
    const int unit = (int)(slot & 0xff), sub = (int)((slot >> 8) & 0xff), nsub = (int)(slot >> 16);
    const WideBwdMlp& d = a.m[g.unit_k[unit]];
    const int O = d.O, ot_n = d.ot, ns2 = (O + 15) >> 4;
    const bool first = 32 * unit == d.joff;
    const unsigned char* btp = a.packbt + ((int64_t)unit * 8 * 3) * kPackBtStep + (half * 32 + c) * 16;
    const int n_tiles = kMidTiles / nsub;
    const int64_t tt0 = grp * kMidTiles + (int64_t)sub * n_tiles;

    f32x16 dbt[OT];
    float dasum[16], dbs[OT];
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) {
      dbs[ot] = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) dbt[ot][r] = 0.f;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) dasum[r] = 0.f;

    // Everything a token tile reads first is requested one tile ahead: its Hpre registers and the first 32 outputs of its
    // dY tile (for O <= 32 the whole tile: one contiguous burst of 32 O floats, element lane + 64 i). Loads only, no use
    // here (a use would wait for them). The tile goes through LDS and is read back in both orientations: the lane's
    // token row (B operand of dHpost^T) and the lane's output column (B operand of dB^T). A first version read those
    // straight from memory: 24 scattered dword loads per tile and unit kept the CU's address unit busy, not the ALUs.
    // Lane l takes elements l + 64 i of the tile. O < 32: the tile is 32 O contiguous floats and keeps that order in LDS
    // (row stride O). O > 32: element (row (l >> 5) + 2 i, output l & 31), LDS row stride 33. Either way memory and LDS
    // addresses are a per-lane base + i x a constant.
    const int W = O < 32 ? O : 32;
    constexpr bool two_d = OT > 1;  // the class of units with more than 32 outputs
    const int g_off = two_d ? (lane >> 5) * O + (lane & 31) : lane, g_step = two_d ? 2 * O : 64;
    const int l_off = two_d ? lane + (lane >> 5) : lane, l_step = two_d ? 66 : 64, l_row = two_d ? 33 : O;
    const int nel = two_d ? 16 : (32 * O + 63) >> 6;  // elements per lane (the last may be partial when O < 32)
    float hnext[16], dyt[NDY];
    auto prefetch = [&](int64_t ttile) {
      const float4* hfp = reinterpret_cast<const float4*>(a.HF + (((ttile * g.U + unit) * 16) << 6)) + lane;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 v = hfp[q * 64];
        hnext[4 * q] = v.x, hnext[4 * q + 1] = v.y, hnext[4 * q + 2] = v.z, hnext[4 * q + 3] = v.w;
      }
      const float* tile = d.dY + ttile * 32 * O;
      const int64_t rows = g.T - ttile * 32;  // valid rows of this tile (may be <= 0 past the end)
      if (rows >= 32) {  // wave-uniform
        const float* p0 = tile + g_off;
#pragma unroll
        for (int i = 0; i < NDY; ++i)
          if (i < nel) dyt[i] = p0[(two_d || lane + 64 * i < 32 * O) ? i * g_step : 0];
      } else {  // the partial tile: rows past T read element 0 of the array and are zeroed when consumed
#pragma unroll
        for (int i = 0; i < NDY; ++i)
          if (i < nel) {
            const int e = g_off + i * g_step;
            const bool ok = two_d ? (lane >> 5) + 2 * i < rows : e < rows * O;
            dyt[i] = ok ? tile[e] : d.dY[0];
          }
      }
    };
    prefetch(tt0);
    for (int it = 0; it < n_tiles; ++it) {
      const int64_t ttile = tt0 + it;
      const int64_t tok = ttile * 32 + c;
      const bool tok_ok = tok < g.T;
      float hpre[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) hpre[r] = hnext[r];
      {
        const int64_t rows = g.T - ttile * 32;
#pragma unroll
        for (int i = 0; i < NDY; ++i)
          if (i < nel) {
            const bool ok = rows >= 32 || (two_d ? (lane >> 5) + 2 * i < rows : g_off + i * g_step < rows * O);
            if (two_d || lane + 64 * i < 32 * O) YT[l_off + i * l_step] = ok ? dyt[i] : 0.f;
          }
      }
      // "consume the old values, THEN issue the next loads" (hipcc would hoist the loads above their predecessors' uses)
#pragma unroll
      for (int r = 0; r < 16; r += 4) asm volatile("" : "+v"(hpre[r]), "+v"(hpre[r + 1]), "+v"(hpre[r + 2]), "+v"(hpre[r + 3]) : : "memory");
      if (it + 1 < n_tiles) prefetch(ttile + 1);
      float dy0[8], yb0[16];
#pragma unroll
      for (int i = 0; i < 8; ++i) dy0[i] = 8 * half + i < W ? YT[c * l_row + 8 * half + i] : 0.f;
#pragma unroll
      for (int i = 0; i < 16; ++i) yb0[i] = c < W ? YT[(8 * half + (i & 7) + 16 * (i >> 3)) * l_row + c] : 0.f;
      // dHpost^T: B operand = the lane's own dY row, 8 outputs per k-step
      f32x16 acc3;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc3[r] = 0.f;
      acc3 = mfma6(load_frag3(btp, kPackBtStep), split_pack8(dy0), acc3);
      if (NDY > 8 && ns2 > 1) {  // outputs 16 .. 31: still inside the LDS tile
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = 16 + 8 * half + i < W ? YT[c * l_row + 16 + 8 * half + i] : 0.f;
        acc3 = mfma6(load_frag3(btp + (int64_t)3 * kPackBtStep, kPackBtStep), split_pack8(v), acc3);
      }
      const bool row4 = (O & 3) == 0 && (reinterpret_cast<uintptr_t>(d.dY) & 15) == 0;  // the lane's row in 16-byte pieces
      for (int s2 = 2; s2 < (OT > 1 ? ns2 : 2); ++s2) {
        float v[8];
        const int o0 = 16 * s2 + 8 * half;
        if (row4) {
          float4 lo = make_float4(0.f, 0.f, 0.f, 0.f), hi = lo;
          if (tok_ok && o0 < O) lo = *reinterpret_cast<const float4*>(d.dY + tok * O + o0);
          if (tok_ok && o0 + 4 < O) hi = *reinterpret_cast<const float4*>(d.dY + tok * O + o0 + 4);
          v[0] = lo.x, v[1] = lo.y, v[2] = lo.z, v[3] = lo.w, v[4] = hi.x, v[5] = hi.y, v[6] = hi.z, v[7] = hi.w;
        } else {
#pragma unroll
          for (int i = 0; i < 8; ++i) v[i] = (tok_ok && o0 + i < O) ? d.dY[tok * O + o0 + i] : 0.f;
        }
        acc3 = mfma6(load_frag3(btp + (int64_t)s2 * 3 * kPackBtStep, kPackBtStep), split_pack8(v), acc3);
      }
      float y[16], gg[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float dd;
        gelu_and_grad(hpre[r], y[r], dd);
        gg[r] = acc3[r] * dd;
        dasum[r] += gg[r];
      }
      // G planes: registers 4 q .. 4 q + 3 of lane (c, half) are rows j = 32 unit + 8 q + 4 half + (0..3) of token c, i.e. 8 of
      // the 32 bytes of the token's row in column block 2 unit + (q >> 1). v_permlane32_swap pairs the two halves of
      // the wave so that lane (c, half) ends up with rows 16 p + 8 half .. + 7 (p = 0, 1): 16 contiguous bytes, and the
      // wave stores each (term, block) as ONE fully coalesced 1 KB dwordx4 burst instead of four scattered 8-byte stores.
      {
        const Split16 gs = split16(gg);
#pragma unroll
        for (int t = 0; t < 3; ++t)
#pragma unroll
          for (int pq = 0; pq < 2; ++pq) {
            const auto s0 = __builtin_amdgcn_permlane32_swap(gs.d[t][4 * pq], gs.d[t][4 * pq + 2], false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(gs.d[t][4 * pq + 1], gs.d[t][4 * pq + 3], false, false);
            unsigned char* dst = a.gp[t] + ((int64_t)(2 * unit + pq) * g.T_pad + tok) * 32 + 16 * half;
            *reinterpret_cast<uint4*>(dst) = uint4{s0[0], s1[0], s0[1], s1[1]};
          }
      }
      // Hpost^T transposed: accumulator layout -> [tok][j] planes -> A operand with the token as k
      Frag3 ha[2];
      {
        const Split16 ys = split16(y);
        store_acc_plane(HP, L, ys, 0);
        store_acc_plane(YP, L, ys, 1);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int s = 0; s < 2; ++s) ha[s].t1 = tr_frag(HP, L, s), ha[s].t2 = tr_frag(YP, L, s);
        asm volatile("" ::: "memory");
        store_acc_plane(HP, L, ys, 2);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int s = 0; s < 2; ++s) ha[s].t3 = tr_frag(HP, L, s);
        asm volatile("" ::: "memory");
      }
      // dB^T: B operand = dY[tok = 16 s + 8 half + i][o = 32 ot + c] (k = token); rows past T, outputs past O are zero
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const float v[8] = {yb0[8 * s], yb0[8 * s + 1], yb0[8 * s + 2], yb0[8 * s + 3], yb0[8 * s + 4], yb0[8 * s + 5], yb0[8 * s + 6], yb0[8 * s + 7]};
        if (first) dbs[0] += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        dbt[0] = mfma6(ha[s], split_pack8(v), dbt[0]);
      }
#pragma unroll
      for (int ot = 1; ot < OT; ++ot) {
        if (ot < ot_n) {
          const int o = 32 * ot + c;
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
              const int64_t t2 = ttile * 32 + 16 * s + 8 * half + i;
              v[i] = (t2 < g.T && o < O) ? d.dY[t2 * O + o] : 0.f;
            }
            if (first) dbs[ot] += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
            dbt[ot] = mfma6(ha[s], split_pack8(v), dbt[ot]);
          }
        }
      }
    }
    // the item's record: [ot][32 j][32 o] dB^T | [32] da | [ot][32] db (first unit of its MLP)
    float* rec = a.part + grp * a.rec_total + a.rec_off[unit] + sub * wide_rec_size(ot_n, first);
#pragma unroll
    for (int ot = 0; ot < OT; ++ot) {
      if (ot < ot_n) {
#pragma unroll
        for (int r = 0; r < 16; ++r) rec[ot * 1024 + cd_row(r, half) * 32 + c] = dbt[ot][r];
        if (first) {
          const float tot = dbs[ot] + __shfl_xor(dbs[ot], 32, 64);
          if (half == 0) rec[ot_n * 1024 + 32 + ot * 32 + c] = tot;
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float tot = half_sum(dasum[r]);
      if (c == 0) rec[ot_n * 1024 + cd_row(r, half)] = tot;
    }
  }
}

// dB, da, db: sum the items' records in a fixed order and scatter. Two stages: kRedSlices workgroup slices each sum a
// contiguous range of the records v = group * nsub + sub with eight loads in flight per thread (one stage over all 250 -
// 1000 records of a unit was a chain of 32 - 125 dependent memory round trips: 86 us for 65 MB), then the slices are added
// in order.
constexpr int kRedSlices = 8;

__global__ void __launch_bounds__(256) wide_reduce_small1_k(const WideBwdArgs a) {
  const WideGeom& g = a.g;
  const int unit = blockIdx.y, slice = blockIdx.z;
  const int k = g.unit_k[unit];
  if (k < 0) return;
  const WideBwdMlp& d = a.m[k];
  const bool first = 32 * unit == d.joff;
  const int rec_size = wide_rec_size(d.ot, first), nsub = wide_nsub(d.ot);
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= rec_size) return;
  const float* p = a.part + a.rec_off[unit] + i;
  const int sh = nsub == 4 ? 2 : nsub == 2 ? 1 : 0, n = a.groups << sh;
  const int per = (n + kRedSlices - 1) / kRedSlices;
  const int v0 = slice * per, v1 = v0 + per < n ? v0 + per : n;
  auto at = [&](int v) { return p[(int64_t)(v >> sh) * a.rec_total + (v & (nsub - 1)) * rec_size]; };
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  int w = v0;
  for (; w + 8 <= v1; w += 8) {
    float v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = at(w + q);
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] += v[q];
  }
  for (int q = 0; w < v1; ++w, ++q) acc[q] += at(w);
  // one slot per (slice, first sub-record position of the unit): the slab of group 0 is laid out like every other
  a.red[(int64_t)slice * a.rec_total + a.rec_off[unit] + i] =
      (((acc[0] + acc[1]) + (acc[2] + acc[3])) + ((acc[4] + acc[5]) + (acc[6] + acc[7])));
}

__global__ void __launch_bounds__(256) wide_reduce_small2_k(const WideBwdArgs a) {
  const WideGeom& g = a.g;
  const int unit = blockIdx.y;
  const int k = g.unit_k[unit];
  if (k < 0) return;
  const WideBwdMlp& d = a.m[k];
  const int hb = 32 * unit - d.joff;
  const bool first = hb == 0;
  const int rec_size = wide_rec_size(d.ot, first);
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= rec_size) return;
  const float* p = a.red + a.rec_off[unit] + i;
  float total = p[0];
#pragma unroll
  for (int sl = 1; sl < kRedSlices; ++sl) total += p[(int64_t)sl * a.rec_total];
  if (i < d.ot * 1024) {
    const int ot = i >> 10, j = (i >> 5) & 31, o = 32 * ot + (i & 31);
    if (hb + j < d.h && o < d.O) d.dB[(int64_t)o * d.h + hb + j] = total;
  } else if (i < d.ot * 1024 + 32) {
    const int j = i - d.ot * 1024;
    if (hb + j < d.h) d.da[hb + j] = total;
  } else {
    const int o = i - d.ot * 1024 - 32;
    if (o < d.O) d.db[o] = total;
  }
}

// dA_k[jj][e] = sum over the K-splits of the TN GEMM's slabs, in order
__global__ void __launch_bounds__(256) wide_reduce_da_k(const WideBwdArgs a) {
  const WideGeom& g = a.g;
  const int e4n = g.E / 4;
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= (int64_t)g.J * e4n) return;
  const int j = (int)(i / e4n), e = 4 * (int)(i - (int64_t)j * e4n);
  const int k = g.unit_k[j >> 5];
  const WideBwdMlp& d = a.m[k];
  const int jj = j - d.joff;
  if (jj >= d.h) return;
  const float* p = a.dapart + (int64_t)j * g.E_pad + e;
  const int64_t slab = (int64_t)g.J_pad * g.E_pad;
  float4 acc = *reinterpret_cast<const float4*>(p);
  for (int sp = 1; sp < a.splits; ++sp) {
    const float4 v = *reinterpret_cast<const float4*>(p + sp * slab);
    acc.x += v.x, acc.y += v.y, acc.z += v.z, acc.w += v.w;
  }
  float* out = d.dA + (int64_t)jj * g.E + e;
  out[0] = acc.x, out[1] = acc.y, out[2] = acc.z, out[3] = acc.w;
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
int64_t up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

struct WidePlan {
  WideGeom g;
  int32_t joff[kMaxWideMlps], ot[kMaxWideMlps];
  // byte sizes / offsets
  int64_t xp_plane, hf_bytes;                 // saved: [3 x XP plane | HF]
  int64_t w1p_plane, w1tp_plane, gp_plane;
  int64_t pack2_bytes, packbt_bytes;
  int32_t splits, groups, rec_total, n_slots[3];
  int32_t tiles_e;   // column tiles of the two GEMMs whose columns are e (dX, dAcat)
  bool narrow_e;     // ... on the 256 x 64 configuration (E <= 128) instead of 256 x 256
  uint32_t rec_off[kMaxWideUnits];
};

bool make_wide_plan(int64_t T, int32_t E, int32_t K, const int32_t* h, const int32_t* O, WidePlan* p) {
  if (T < 1 || E < 16 || E > 1024 || (E & 15) || K < 1 || K > kMaxWideMlps || !h || !O) return false;
  WideGeom& g = p->g;
  int j = 0, rec = 0;
  p->n_slots[0] = p->n_slots[1] = p->n_slots[2] = 0;
  for (int u = 0; u < kMaxWideUnits; ++u) g.unit_k[u] = -1, p->rec_off[u] = 0;
  for (int k = 0; k < K; ++k) {
    if (h[k] < 1 || h[k] > 128 || O[k] < 1 || O[k] > 32 * kOtMax) return false;
    p->joff[k] = j;
    p->ot[k] = (O[k] + 31) / 32;
    const int nu = (h[k] + 31) / 32;
    for (int u = 0; u < nu; ++u) {
      g.unit_k[j / 32 + u] = k;
      p->rec_off[j / 32 + u] = (uint32_t)rec;
      rec += wide_nsub(p->ot[k]) * wide_rec_size(p->ot[k], u == 0);
      p->n_slots[wide_class(O[k])] += wide_nsub(p->ot[k]);
    }
    j += 32 * nu;
  }
  g.T = T, g.T_pad = up(T, 256);
  g.E = E, g.E_pad = (int32_t)up(E, 256);
  g.J = j, g.J_pad = (int32_t)up(j, 256);
  g.K = K, g.U = g.J_pad / 32;
  p->rec_total = rec;
  p->xp_plane = (int64_t)(E / 16) * g.T_pad * 32;
  p->hf_bytes = (g.T_pad / 32) * (int64_t)g.U * 4096;
  p->w1p_plane = (int64_t)(E / 16) * g.J_pad * 32;
  p->w1tp_plane = (int64_t)(g.J / 16) * g.E_pad * 32;
  p->gp_plane = (int64_t)(g.J / 16) * g.T_pad * 32;
  if (p->xp_plane >= (int64_t)1 << 31 || p->gp_plane >= (int64_t)1 << 31) return false;  // 32-bit lane offsets in the GEMM loader
  p->pack2_bytes = (int64_t)(g.J / 32) * kOtMax * 3 * kPack2Tile;
  p->packbt_bytes = (int64_t)(g.J / 32) * 8 * 3 * kPackBtStep;
  p->groups = (int32_t)(g.T_pad / 256);
  // split-K of the weight-gradient GEMM: ONE round of workgroups (one per CU, never 256 + a few: the few would run a
  // second round alone), at least 32 k-chunks (512 tokens) each
  p->narrow_e = E <= 128;
  p->tiles_e = p->narrow_e ? (E + kTileNarrow - 1) / kTileNarrow : g.E_pad / 256;
  const int tiles = (g.J_pad / 256) * p->tiles_e;
  const int64_t chunks = g.T_pad / 16;
  int s = 256 / tiles;
  if (s > chunks / 32) s = (int)(chunks / 32);
  if (s > kSplitsMax) s = kSplitsMax;
  if (s < 1) s = 1;
  p->splits = s;
  return true;
}

int64_t saved_bytes(const WidePlan& p) { return 3 * p.xp_plane + p.hf_bytes; }
int64_t fwd_ws_bytes(const WidePlan& p) {
  return 3 * p.w1p_plane + up((int64_t)p.g.J_pad * 4, 256) + p.pack2_bytes + up((int64_t)p.g.K * kOtMax * 32 * 4, 256);
}
int64_t bwd_ws_bytes(const WidePlan& p) {
  return 3 * p.w1tp_plane + p.packbt_bytes + kSlotBytes + 3 * p.gp_plane + up((int64_t)(p.groups + 8) * p.rec_total * 4, 256) +
         (int64_t)p.splits * p.g.J_pad * p.g.E_pad * 4;
}

hipError_t launch_gemm(bool tn, bool narrow, const GemmArgs& ga, hipStream_t s) {
  if (ga.splits < 1 || ga.chunks < ga.splits) return hipErrorInvalidValue;  // every (tile, split) item needs a K range
  const int64_t items = (int64_t)ga.tiles_m * ga.tiles_n * ga.splits;
  const dim3 grid((unsigned)(items < 256 ? items : 256));  // persistent: one workgroup per CU of the MI355X
  if (tn && narrow) hipLaunchKernelGGL((x3_gemm_k<true, GemmNarrow>), grid, dim3(kGemmThreads), 0, s, ga);
  else if (tn) hipLaunchKernelGGL((x3_gemm_k<true, GemmSquare>), grid, dim3(kGemmThreads), 0, s, ga);
  else if (narrow) hipLaunchKernelGGL((x3_gemm_k<false, GemmNarrow>), grid, dim3(kGemmThreads), 0, s, ga);
  else hipLaunchKernelGGL((x3_gemm_k<false, GemmSquare>), grid, dim3(kGemmThreads), 0, s, ga);
  return hipGetLastError();
}

}  // namespace

extern "C" {

int64_t psf_mlp_wide_saved_bytes(int64_t T, int32_t E, int32_t K, const int32_t* h, const int32_t* O) {
  WidePlan p;
  return make_wide_plan(T, E, K, h, O, &p) ? saved_bytes(p) : -1;
}
int64_t psf_mlp_wide_fwd_workspace(int64_t T, int32_t E, int32_t K, const int32_t* h, const int32_t* O) {
  WidePlan p;
  return make_wide_plan(T, E, K, h, O, &p) ? fwd_ws_bytes(p) : -1;
}
int64_t psf_mlp_wide_bwd_workspace(int64_t T, int32_t E, int32_t K, const int32_t* h, const int32_t* O) {
  WidePlan p;
  return make_wide_plan(T, E, K, h, O, &p) ? bwd_ws_bytes(p) : -1;
}

int psf_mlp_wide_fwd_f32(const float* X, int64_t T, int32_t E, int32_t K, const float* const* A, const float* const* a,
                         const float* const* B, const float* const* b, const int32_t* h, const int32_t* O, float* const* Y,
                         void* saved, int64_t saved_bytes_given, void* workspace, int64_t workspace_bytes, void* stream) {
  if (!X || !A || !a || !B || !b || !h || !O || !Y || !workspace) return psf_internal_fail(PSF_E_NULL, "psf_mlp_wide_fwd: NULL argument");
  WidePlan p;
  if (!make_wide_plan(T, E, K, h, O, &p))
    return psf_internal_fail(PSF_E_SHAPE, "psf_mlp_wide_fwd: need T >= 1, E a multiple of 16 in [16, 1024], 1 <= K <= 24, 1 <= h <= 128, 1 <= O <= 128");
  if ((reinterpret_cast<uintptr_t>(X) & 15) || (reinterpret_cast<uintptr_t>(saved) & 255) || (reinterpret_cast<uintptr_t>(workspace) & 255))
    return psf_internal_fail(PSF_E_ALIGN, "psf_mlp_wide_fwd: X must be 16-byte aligned, saved and workspace 256-byte aligned");
  // saved == NULL: inference. The record lives in the workspace (which must then hold both) and need not be complete.
  const bool keep = saved != nullptr;
  const int64_t ws_need = fwd_ws_bytes(p) + (keep ? 0 : saved_bytes(p));
  if ((keep && saved_bytes_given < saved_bytes(p)) || workspace_bytes < ws_need)
    return psf_internal_fail(PSF_E_SHAPE, "psf_mlp_wide_fwd: saved / workspace too small (psf_mlp_wide_saved_bytes, psf_mlp_wide_fwd_workspace; "
                                          "with saved == NULL the workspace must hold both)");
  WideFwdArgs fa;
  for (int k = 0; k < kMaxWideMlps; ++k) fa.m[k] = WideFwdMlp{nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0};
  for (int k = 0; k < K; ++k) {
    if (!A[k] || !a[k] || !B[k] || !b[k] || !Y[k]) return psf_internal_fail(PSF_E_NULL, "psf_mlp_wide_fwd: NULL layer pointer");
    if (reinterpret_cast<uintptr_t>(A[k]) & 15) return psf_internal_fail(PSF_E_ALIGN, "psf_mlp_wide_fwd: first-layer weights must be 16-byte aligned");
    fa.m[k] = WideFwdMlp{A[k], a[k], B[k], b[k], Y[k], h[k], O[k], p.joff[k], p.ot[k]};
  }
  fa.g = p.g;
  unsigned char* ws = reinterpret_cast<unsigned char*>(workspace);
  unsigned char* sv = keep ? reinterpret_cast<unsigned char*>(saved) : ws + fwd_ws_bytes(p);
  for (int t = 0; t < 3; ++t) fa.w1p[t] = ws + t * p.w1p_plane;
  fa.acat = reinterpret_cast<float*>(ws + 3 * p.w1p_plane);
  fa.pack2 = ws + 3 * p.w1p_plane + up((int64_t)p.g.J_pad * 4, 256);
  fa.b2 = reinterpret_cast<float*>(fa.pack2 + p.pack2_bytes);
  float* HF = reinterpret_cast<float*>(sv + 3 * p.xp_plane);
  fa.HF = HF;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);

  hipLaunchKernelGGL(wide_pack_fwd_k, dim3(512), dim3(256), 0, s, fa);
  SplitArgs sa;
  sa.src = X;
  for (int t = 0; t < 3; ++t) sa.p[t] = sv + t * p.xp_plane;
  sa.rows = T, sa.rows_pad = p.g.T_pad, sa.ld = E, sa.blocks = E / 16;
  {
    const int by = (sa.blocks + 3) / 4;
    hipLaunchKernelGGL(x3_split_planes_k, dim3((unsigned)(p.g.T_pad / 64), (unsigned)(by < 8 ? by : 8)), dim3(256), 0, s, sa);
  }
  GemmArgs ga{};
  for (int t = 0; t < 3; ++t) ga.A.p[t] = fa.w1p[t], ga.B.p[t] = sa.p[t];
  ga.A.rows_pad = p.g.J_pad, ga.A.blocks = E / 16;
  ga.B.rows_pad = p.g.T_pad, ga.B.blocks = E / 16;
  ga.tiles_m = p.g.J_pad / 256, ga.tiles_n = (int32_t)(p.g.T_pad / 256), ga.splits = 1, ga.chunks = E / 16;
  ga.n_fast = 0, ga.epilogue = kEpiFragBias, ga.out = HF, ga.bias = fa.acat;
  ga.rows_valid = p.g.J_pad, ga.cols_valid = p.g.T_pad;
  // Second layers of the MLPs with <= 32 outputs run in the GEMM's epilogue when every MLP is exactly one wave tile of
  // 128 (padded) hidden rows: the GEMM wave that holds an MLP's Hpre applies GELU and the second layer to it in registers.
  bool whole = true;
  for (int k = 0; k < K; ++k) whole = whole && (h[k] + 31) / 32 == 4;
  fa.skip_mask = 0;
  ga.fo.enabled = 0;
  if (whole && psf_g_wide_fuse.load()) {
    ga.fo.enabled = 1;
    ga.fo.store_hpre = keep ? 1 : 0;
    ga.fo.pack2 = fa.pack2;
    ga.fo.b2 = fa.b2;
    ga.fo.T = T;
    for (int k = 0; k < kFusedMlpsMax; ++k) ga.fo.Y[k] = nullptr, ga.fo.O[k] = 0;
    for (int k = 0; k < K; ++k)
      if (p.ot[k] == 1) ga.fo.Y[k] = Y[k], ga.fo.O[k] = O[k], fa.skip_mask |= 1u << k;
  }
  hipError_t e = launch_gemm(false, false, ga, s);
  if (e != hipSuccess) return psf_internal_fail((int)e, hipGetErrorString(e));
  int left = 0;
  for (int k = 0; k < K; ++k) left += !((fa.skip_mask >> k) & 1);
  if (left) hipLaunchKernelGGL(wide_out_k, dim3((unsigned)((T + 127) / 128), (unsigned)(left < 12 ? left : 12)), dim3(256), 0, s, fa);
  e = hipGetLastError();
  return e == hipSuccess ? PSF_OK : psf_internal_fail((int)e, hipGetErrorString(e));
}

int psf_mlp_wide_bwd_f32(const void* saved, int64_t saved_bytes_given, int64_t T, int32_t E, int32_t K, const float* const* A,
                         const float* const* B, const int32_t* h, const int32_t* O, const float* const* dY, float* dX,
                         float* const* dA, float* const* da, float* const* dB, float* const* db, void* workspace,
                         int64_t workspace_bytes, void* stream) {
  if (!saved || !A || !B || !h || !O || !dY || !dA || !da || !dB || !db || !workspace) return psf_internal_fail(PSF_E_NULL, "psf_mlp_wide_bwd: NULL argument");
  WidePlan p;
  if (!make_wide_plan(T, E, K, h, O, &p))
    return psf_internal_fail(PSF_E_SHAPE, "psf_mlp_wide_bwd: need T >= 1, E a multiple of 16 in [16, 1024], 1 <= K <= 24, 1 <= h <= 128, 1 <= O <= 128");
  if ((reinterpret_cast<uintptr_t>(saved) & 255) || (reinterpret_cast<uintptr_t>(workspace) & 255))
    return psf_internal_fail(PSF_E_ALIGN, "psf_mlp_wide_bwd: saved and workspace must be 256-byte aligned");
  if (saved_bytes_given < saved_bytes(p) || workspace_bytes < bwd_ws_bytes(p))
    return psf_internal_fail(PSF_E_SHAPE, "psf_mlp_wide_bwd: saved / workspace too small (psf_mlp_wide_saved_bytes, psf_mlp_wide_bwd_workspace)");
  WideBwdArgs ba;
  for (int k = 0; k < kMaxWideMlps; ++k) ba.m[k] = WideBwdMlp{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0};
  for (int k = 0; k < K; ++k) {
    if (!A[k] || !B[k] || !dY[k] || !dA[k] || !da[k] || !dB[k] || !db[k]) return psf_internal_fail(PSF_E_NULL, "psf_mlp_wide_bwd: NULL layer pointer");
    ba.m[k] = WideBwdMlp{A[k], B[k], dY[k], dA[k], da[k], dB[k], db[k], h[k], O[k], p.joff[k], p.ot[k]};
  }
  ba.g = p.g;
  const unsigned char* sv = reinterpret_cast<const unsigned char*>(saved);
  unsigned char* ws = reinterpret_cast<unsigned char*>(workspace);
  for (int t = 0; t < 3; ++t) ba.w1tp[t] = ws + t * p.w1tp_plane;
  ba.packbt = ws + 3 * p.w1tp_plane;
  ba.slots = reinterpret_cast<uint32_t*>(ba.packbt + p.packbt_bytes);
  for (int t = 0; t < 3; ++t) ba.gp[t] = ba.packbt + p.packbt_bytes + kSlotBytes + t * p.gp_plane;
  ba.part = reinterpret_cast<float*>(ba.gp[0] + 3 * p.gp_plane);
  ba.red = ba.part + (int64_t)p.groups * p.rec_total;  // kRedSlices = 8 slabs
  float* dapart = ba.part + up((int64_t)(p.groups + 8) * p.rec_total * 4, 256) / 4;
  ba.dapart = dapart;
  ba.HF = reinterpret_cast<const float*>(sv + 3 * p.xp_plane);
  for (int u = 0; u < kMaxWideUnits; ++u) ba.rec_off[u] = p.rec_off[u];
  ba.rec_total = p.rec_total, ba.groups = p.groups, ba.splits = p.splits;
  for (int c3 = 0; c3 < 3; ++c3) ba.n_slots[c3] = p.n_slots[c3];
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);

  hipLaunchKernelGGL(wide_pack_bwd_k, dim3(512), dim3(256), 0, s, ba);
  for (int cls = 0; cls < 3; ++cls) {  // (256-token group, hidden unit [, part of the group]): one wave per item
    const int64_t items = (int64_t)p.groups * p.n_slots[cls];
    if (!items) continue;
    const int64_t wgs = (items + 7) / 8;
    const dim3 grid((unsigned)(wgs < 256 ? wgs : 256));
    if (cls == 0) hipLaunchKernelGGL((wide_mid_k<1, 8>), grid, dim3(512), 0, s, ba, cls);
    else if (cls == 1) hipLaunchKernelGGL((wide_mid_k<1, 16>), grid, dim3(512), 0, s, ba, cls);
    else hipLaunchKernelGGL((wide_mid_k<kOtMax, 16>), grid, dim3(512), 0, s, ba, cls);
  }
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return psf_internal_fail((int)e, hipGetErrorString(e));
  if (dX) {  // dX[tok][e] = sum_j G[tok][j] Wcat[j][e]
    GemmArgs ga{};
    for (int t = 0; t < 3; ++t) ga.A.p[t] = ba.gp[t], ga.B.p[t] = ba.w1tp[t];
    ga.A.rows_pad = p.g.T_pad, ga.A.blocks = p.g.J / 16;
    ga.B.rows_pad = p.g.E_pad, ga.B.blocks = p.g.J / 16;
    ga.tiles_m = (int32_t)(p.g.T_pad / 256), ga.tiles_n = p.tiles_e, ga.splits = 1, ga.chunks = p.g.J / 16;
    ga.n_fast = 1, ga.epilogue = kEpiRowMajor, ga.out = dX, ga.ld = E, ga.rows_alloc = T;
    ga.rows_valid = T, ga.cols_valid = E;
    e = launch_gemm(false, p.narrow_e, ga, s);
    if (e != hipSuccess) return psf_internal_fail((int)e, hipGetErrorString(e));
  }
  {  // dAcat[j][e] = sum_tok G[tok][j] X[tok][e]
    GemmArgs ga{};
    for (int t = 0; t < 3; ++t) ga.A.p[t] = ba.gp[t], ga.B.p[t] = sv + t * p.xp_plane;
    ga.A.rows_pad = p.g.T_pad, ga.A.blocks = p.g.J / 16;
    ga.B.rows_pad = p.g.T_pad, ga.B.blocks = E / 16;
    ga.tiles_m = p.g.J_pad / 256, ga.tiles_n = p.tiles_e, ga.splits = p.splits, ga.chunks = (int32_t)(p.g.T_pad / 16);
    ga.n_fast = 0, ga.epilogue = kEpiRowMajor, ga.out = dapart, ga.ld = p.g.E_pad, ga.rows_alloc = p.g.J_pad;
    ga.rows_valid = p.g.J_pad, ga.cols_valid = p.g.E_pad;
    e = launch_gemm(true, p.narrow_e, ga, s);
    if (e != hipSuccess) return psf_internal_fail((int)e, hipGetErrorString(e));
  }
  static_assert(kRedSlices == 8, "workspace layout");
  {
    const dim3 grid((kOtMax * 1024 + 32 + kOtMax * 32 + 255) / 256, (unsigned)(p.g.J / 32), kRedSlices);
    hipLaunchKernelGGL(wide_reduce_small1_k, grid, dim3(256), 0, s, ba);
    hipLaunchKernelGGL(wide_reduce_small2_k, dim3(grid.x, grid.y), dim3(256), 0, s, ba);
  }
  hipLaunchKernelGGL(wide_reduce_da_k, dim3((unsigned)(((int64_t)p.g.J * (E / 4) + 255) / 256)), dim3(256), 0, s, ba);
  e = hipGetLastError();
  return e == hipSuccess ? PSF_OK : psf_internal_fail((int)e, hipGetErrorString(e));
}

}  // extern "C"
