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
// accumulator-fragment order every consumer wants: register r of lane l of the 32 x 32 tile (hidden unit, token tile)
// at ((tile_t * U + unit) * 16 + r) * 64 + l — 256-byte rows, coalesced for producer and consumers alike).
// Nothing is recomputed: at these widths recomputing Hpre would be a fourth 100-GFLOP GEMM, keeping it is 4 J bytes
// per token of HBM traffic. All reductions run in a fixed order (no float atomics): bit-reproducible.
//
// Limits: E a multiple of 16, 16 <= E <= 1024; 1 <= h <= 128; 1 <= O <= 128; K <= 24; every plane below 2 GiB.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/psf_chord.h"
#include "mlp_planes.h"
#include "mlp_x3_common.h"
#include "x3_gemm.h"

extern "C" int psf_internal_fail(int code, const char* message);

namespace {

using namespace psf_wide;

constexpr int kMaxWideMlps = 24;
constexpr int kMaxWideUnits = 96;      // 24 MLPs x 4 units of 32 hidden rows
constexpr int kOtMax = 4;              // output tiles of 32 per MLP (O <= 128)
constexpr int kPack2Tile = 2048;       // one term of one (unit, output tile): [2 s][2 half][32 o][8] bf16
constexpr int kPackBtStep = 1024;      // one term of one (unit, k-step of 16 outputs): [2 half][32 j][8] bf16
constexpr int kSplitsMax = 64;

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
__device__ __forceinline__ float gelu_only(float x) {
  const float t = __builtin_amdgcn_rcpf(fmaf(fabsf(x), 0.2316419f, 1.0f));
  float p = fmaf(0.53070271f, t, -0.72657602f);
  p = fmaf(p, t, 0.71070687f);
  p = fmaf(p, t, -0.14224837f);
  p = fmaf(p, t, 0.12741479f);
  p = p * t;
  const float E = __builtin_amdgcn_exp2f((x * x) * -0.72134752044448170368f);
  return x * (0.5f + copysignf(0.5f - p * E, x));
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
  float* part;             // [workgroups][rec_total] partial sums of dB^T, da, db
  const float* dapart;     // [splits][J_pad][E_pad] partial dAcat
  uint32_t rec_off[kMaxWideUnits];  // float offset of a unit's record: [ot][1024] dB^T | [32] da | [ot][32] db (first unit of its MLP)
  int32_t rec_total, groups, splits;
};

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
  for (int k = 0; k < g.K; ++k) {
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
      const float* hf = a.HF + (((ttile * g.U + unit) * 16) << 6) + lane;
      float y[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) y[r] = hf[r * 64];
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
// 512 threads, one 32-token tile per wave (8 tiles = one 256-token GEMM tile per workgroup); per unit the eight waves'
// sums are combined through LDS in a fixed order and flushed to the workgroup's record (one barrier per unit and
// output tile: two combine buffers alternate).
// ---------------------------------------------------------------------------------------------------------------
constexpr int kMidScr = 1024 + 32 + 32;   // floats: dB^T tile [32 j][32 o] | da [32] | db [32]
constexpr int kMidWaveBytes = 2 * kPlaneBytes + 2 * kMidScr * 4;

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

__global__ void __launch_bounds__(512, 1) wide_mid_k(const WideBwdArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char wave_lds[8 * kMidWaveBytes];
  const WideGeom& g = a.g;
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 31, half = lane >> 5;
  unsigned char* HP = wave_lds + wv * kMidWaveBytes;
  unsigned char* YP = HP + kPlaneBytes;
  const PlaneLane L = plane_lane(lane);
  const int64_t ttile = (int64_t)blockIdx.x * 8 + wv;  // < T_pad / 32 by construction of the grid
  const int64_t tok = ttile * 32 + c;
  const bool tok_ok = tok < g.T;
  float* part = a.part + (int64_t)blockIdx.x * a.rec_total;
  int par = 0;  // combine buffer parity (block-uniform: every wave runs the same sequence)

  auto scr_of = [&](int w, int p) -> float* {
    return reinterpret_cast<float*>(wave_lds + w * kMidWaveBytes + 2 * kPlaneBytes) + p * kMidScr;
  };
  auto sum8 = [&](int p, int off) {  // float4 at float offset `off` of the combine buffers, waves in order
    float4 acc = *reinterpret_cast<const float4*>(scr_of(0, p) + off);
#pragma unroll
    for (int w = 1; w < 8; ++w) {
      const float4 v = *reinterpret_cast<const float4*>(scr_of(w, p) + off);
      acc.x += v.x, acc.y += v.y, acc.z += v.z, acc.w += v.w;
    }
    return acc;
  };

  for (int k = 0; k < g.K; ++k) {
    const WideBwdMlp& d = a.m[k];
    const int O = d.O, ot_n = d.ot, ns2 = (O + 15) >> 4, unit0 = d.joff >> 5, nu = (d.h + 31) >> 5;
    // B operand of dB^T: dY[tok = 16 s + 8 half + i][o = 32 ot + c] (k = token). Rows past T and outputs past O are zero.
    auto load_yb = [&](int ot, int s) {
      float v[8];
      const int o = 32 * ot + c;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int64_t t2 = ttile * 32 + 16 * s + 8 * half + i;
        v[i] = (t2 < g.T && o < O) ? d.dY[t2 * O + o] : 0.f;
      }
      return split_pack8(v);
    };
    Frag3 yb0[2];  // output tile 0, kept for all units of the MLP
    float dbs[kOtMax];
#pragma unroll
    for (int ot = 0; ot < kOtMax; ++ot) dbs[ot] = 0.f;
#pragma unroll
    for (int s = 0; s < 2; ++s) yb0[s] = load_yb(0, s);
    // db[o = 32 ot + c]: the lane's half sums 16 tokens, the two halves are added below
#pragma unroll
    for (int ot = 0; ot < kOtMax; ++ot) {
      if (ot < ot_n) {
        const int o = 32 * ot + c;
        float sacc = 0.f;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            const int64_t t2 = ttile * 32 + 16 * s + 8 * half + i;
            sacc += (t2 < g.T && o < O) ? d.dY[t2 * O + o] : 0.f;
          }
        dbs[ot] = sacc + __shfl_xor(sacc, 32, 64);
      }
    }

    for (int u = 0; u < nu; ++u) {
      const int unit = unit0 + u;
      const float* hfp = a.HF + (((ttile * g.U + unit) * 16) << 6) + lane;
      float hpre[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) hpre[r] = hfp[r * 64];
      // dHpost^T
      f32x16 acc3;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc3[r] = 0.f;
      for (int s2 = 0; s2 < ns2; ++s2) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int o = 16 * s2 + 8 * half + i;
          v[i] = (tok_ok && o < O) ? d.dY[tok * O + o] : 0.f;
        }
        const Frag3 dy3 = split_pack8(v);
        const Frag3 bt = load_frag3(a.packbt + ((int64_t)(unit * 8 + s2) * 3) * kPackBtStep + (half * 32 + c) * 16, kPackBtStep);
        acc3 = mfma6(bt, dy3, acc3);
      }
      float y[16], gg[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float dd;
        gelu_and_grad(hpre[r], y[r], dd);
        gg[r] = acc3[r] * dd;
      }
      // G planes: registers 4 q .. 4 q + 3 are rows j = 32 unit + 8 q + 4 half + (0..3) of token c: 8 bytes of the token's
      // 32-byte row in column block 2 unit + (q >> 1)
      const Split16 gs = split16(gg);
#pragma unroll
      for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          unsigned char* dst = a.gp[t] + ((int64_t)(2 * unit + (q >> 1)) * g.T_pad + tok) * 32 + (8 * (q & 1) + 4 * half) * 2;
          *reinterpret_cast<uint2*>(dst) = uint2{gs.d[t][2 * q], gs.d[t][2 * q + 1]};
        }
      // Hpost^T transposed: accumulator layout -> [tok][j] planes -> A operand with the token as k
      const Split16 ys = split16(y);
      Frag3 ha[2];
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
      // da
      float das[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) das[r] = half_sum(gg[r]);

      const uint32_t rec = a.rec_off[unit];
      for (int ot = 0; ot < ot_n; ++ot) {
        f32x16 dbt;
#pragma unroll
        for (int r = 0; r < 16; ++r) dbt[r] = 0.f;
        if (ot == 0) {
#pragma unroll
          for (int s = 0; s < 2; ++s) dbt = mfma6(ha[s], yb0[s], dbt);
        } else {
#pragma unroll
          for (int s = 0; s < 2; ++s) dbt = mfma6(ha[s], load_yb(ot, s), dbt);
        }
        float* scr = scr_of(wv, par);
#pragma unroll
        for (int r = 0; r < 16; ++r) scr[cd_row(r, half) * 32 + c] = dbt[r];
        if (ot == 0) {
          if (c == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) scr[1024 + cd_row(r, half)] = das[r];
          }
        }
        // db of output tile ot travels with the dB^T tile of the MLP's first unit
        if (u == 0 && half == 0) scr[1056 + c] = ot == 0 ? dbs[0] : ot == 1 ? dbs[1] : ot == 2 ? dbs[2] : dbs[3];
        __syncthreads();
        if (wv < 4) *reinterpret_cast<float4*>(part + rec + ot * 1024 + wv * 256 + 4 * lane) = sum8(par, wv * 256 + 4 * lane);
        if (wv == 4 && ot == 0 && lane < 8) *reinterpret_cast<float4*>(part + rec + ot_n * 1024 + 4 * lane) = sum8(par, 1024 + 4 * lane);
        if (wv == 5 && u == 0 && lane < 8)
          *reinterpret_cast<float4*>(part + rec + ot_n * 1024 + 32 + ot * 32 + 4 * lane) = sum8(par, 1056 + 4 * lane);
        par ^= 1;
      }
    }
  }
}

// dB, da, db: sum the workgroups' records in a fixed order (four interleaved running sums) and scatter.
__global__ void __launch_bounds__(256) wide_reduce_small_k(const WideBwdArgs a) {
  const WideGeom& g = a.g;
  const int unit = blockIdx.y;
  const int k = g.unit_k[unit];
  if (k < 0) return;
  const WideBwdMlp& d = a.m[k];
  const int hb = 32 * unit - d.joff;
  const bool first = hb == 0;
  const int rec_size = d.ot * 1024 + 32 + (first ? d.ot * 32 : 0);
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= rec_size) return;
  const float* p = a.part + a.rec_off[unit] + i;
  float acc[4] = {0.f, 0.f, 0.f, 0.f};
  int w = 0;
  for (; w + 4 <= a.groups; w += 4) {
    float v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = p[(int64_t)(w + q) * a.rec_total];
#pragma unroll
    for (int q = 0; q < 4; ++q) acc[q] += v[q];
  }
  for (int q = 0; w < a.groups; ++w, ++q) acc[q] += p[(int64_t)w * a.rec_total];
  const float total = ((acc[0] + acc[1]) + acc[2]) + acc[3];
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
  int32_t splits, groups, rec_total;
  uint32_t rec_off[kMaxWideUnits];
};

bool make_wide_plan(int64_t T, int32_t E, int32_t K, const int32_t* h, const int32_t* O, WidePlan* p) {
  if (T < 1 || E < 16 || E > 1024 || (E & 15) || K < 1 || K > kMaxWideMlps || !h || !O) return false;
  WideGeom& g = p->g;
  int j = 0, rec = 0;
  for (int u = 0; u < kMaxWideUnits; ++u) g.unit_k[u] = -1, p->rec_off[u] = 0;
  for (int k = 0; k < K; ++k) {
    if (h[k] < 1 || h[k] > 128 || O[k] < 1 || O[k] > 32 * kOtMax) return false;
    p->joff[k] = j;
    p->ot[k] = (O[k] + 31) / 32;
    const int nu = (h[k] + 31) / 32;
    for (int u = 0; u < nu; ++u) {
      g.unit_k[j / 32 + u] = k;
      p->rec_off[j / 32 + u] = (uint32_t)rec;
      rec += p->ot[k] * 1024 + 32 + (u == 0 ? p->ot[k] * 32 : 0);
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
  // split-K of the weight-gradient GEMM: enough workgroups for every CU, at least 32 k-chunks (512 tokens) each
  const int tiles = (g.J_pad / 256) * (g.E_pad / 256);
  const int64_t chunks = g.T_pad / 16;
  int s = (256 + tiles - 1) / tiles;
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
  return 3 * p.w1tp_plane + p.packbt_bytes + 3 * p.gp_plane + up((int64_t)p.groups * p.rec_total * 4, 256) +
         (int64_t)p.splits * p.g.J_pad * p.g.E_pad * 4;
}

hipError_t launch_gemm(bool tn, const GemmArgs& ga, hipStream_t s) {
  const unsigned grid = (unsigned)(ga.tiles_m * ga.tiles_n * ga.splits);
  if (tn) hipLaunchKernelGGL(x3_gemm_k<true>, dim3(grid), dim3(kGemmThreads), 0, s, ga);
  else hipLaunchKernelGGL(x3_gemm_k<false>, dim3(grid), dim3(kGemmThreads), 0, s, ga);
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
  if (!X || !A || !a || !B || !b || !h || !O || !Y || !saved || !workspace) return psf_internal_fail(PSF_E_NULL, "psf_mlp_wide_fwd: NULL argument");
  WidePlan p;
  if (!make_wide_plan(T, E, K, h, O, &p))
    return psf_internal_fail(PSF_E_SHAPE, "psf_mlp_wide_fwd: need T >= 1, E a multiple of 16 in [16, 1024], 1 <= K <= 24, 1 <= h <= 128, 1 <= O <= 128");
  if ((reinterpret_cast<uintptr_t>(X) & 15) || (reinterpret_cast<uintptr_t>(saved) & 255) || (reinterpret_cast<uintptr_t>(workspace) & 255))
    return psf_internal_fail(PSF_E_ALIGN, "psf_mlp_wide_fwd: X must be 16-byte aligned, saved and workspace 256-byte aligned");
  if (saved_bytes_given < saved_bytes(p) || workspace_bytes < fwd_ws_bytes(p))
    return psf_internal_fail(PSF_E_SHAPE, "psf_mlp_wide_fwd: saved / workspace too small (psf_mlp_wide_saved_bytes, psf_mlp_wide_fwd_workspace)");
  WideFwdArgs fa;
  for (int k = 0; k < kMaxWideMlps; ++k) fa.m[k] = WideFwdMlp{nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, 0, 0};
  for (int k = 0; k < K; ++k) {
    if (!A[k] || !a[k] || !B[k] || !b[k] || !Y[k]) return psf_internal_fail(PSF_E_NULL, "psf_mlp_wide_fwd: NULL layer pointer");
    if (reinterpret_cast<uintptr_t>(A[k]) & 15) return psf_internal_fail(PSF_E_ALIGN, "psf_mlp_wide_fwd: first-layer weights must be 16-byte aligned");
    fa.m[k] = WideFwdMlp{A[k], a[k], B[k], b[k], Y[k], h[k], O[k], p.joff[k], p.ot[k]};
  }
  fa.g = p.g;
  unsigned char* sv = reinterpret_cast<unsigned char*>(saved);
  unsigned char* ws = reinterpret_cast<unsigned char*>(workspace);
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
  hipError_t e = launch_gemm(false, ga, s);
  if (e != hipSuccess) return psf_internal_fail((int)e, hipGetErrorString(e));
  hipLaunchKernelGGL(wide_out_k, dim3((unsigned)((T + 127) / 128)), dim3(256), 0, s, fa);
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
  for (int t = 0; t < 3; ++t) ba.gp[t] = ba.packbt + p.packbt_bytes + t * p.gp_plane;
  ba.part = reinterpret_cast<float*>(ba.gp[0] + 3 * p.gp_plane);
  float* dapart = ba.part + up((int64_t)p.groups * p.rec_total * 4, 256) / 4;
  ba.dapart = dapart;
  ba.HF = reinterpret_cast<const float*>(sv + 3 * p.xp_plane);
  for (int u = 0; u < kMaxWideUnits; ++u) ba.rec_off[u] = p.rec_off[u];
  ba.rec_total = p.rec_total, ba.groups = p.groups, ba.splits = p.splits;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);

  hipLaunchKernelGGL(wide_pack_bwd_k, dim3(512), dim3(256), 0, s, ba);
  hipLaunchKernelGGL(wide_mid_k, dim3((unsigned)p.groups), dim3(512), 0, s, ba);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return psf_internal_fail((int)e, hipGetErrorString(e));
  if (dX) {  // dX[tok][e] = sum_j G[tok][j] Wcat[j][e]
    GemmArgs ga{};
    for (int t = 0; t < 3; ++t) ga.A.p[t] = ba.gp[t], ga.B.p[t] = ba.w1tp[t];
    ga.A.rows_pad = p.g.T_pad, ga.A.blocks = p.g.J / 16;
    ga.B.rows_pad = p.g.E_pad, ga.B.blocks = p.g.J / 16;
    ga.tiles_m = (int32_t)(p.g.T_pad / 256), ga.tiles_n = p.g.E_pad / 256, ga.splits = 1, ga.chunks = p.g.J / 16;
    ga.n_fast = 1, ga.epilogue = kEpiRowMajor, ga.out = dX, ga.ld = E, ga.rows_alloc = T;
    ga.rows_valid = T, ga.cols_valid = E;
    e = launch_gemm(false, ga, s);
    if (e != hipSuccess) return psf_internal_fail((int)e, hipGetErrorString(e));
  }
  {  // dAcat[j][e] = sum_tok G[tok][j] X[tok][e]
    GemmArgs ga{};
    for (int t = 0; t < 3; ++t) ga.A.p[t] = ba.gp[t], ga.B.p[t] = sv + t * p.xp_plane;
    ga.A.rows_pad = p.g.T_pad, ga.A.blocks = p.g.J / 16;
    ga.B.rows_pad = p.g.T_pad, ga.B.blocks = E / 16;
    ga.tiles_m = p.g.J_pad / 256, ga.tiles_n = p.g.E_pad / 256, ga.splits = p.splits, ga.chunks = (int32_t)(p.g.T_pad / 16);
    ga.n_fast = 0, ga.epilogue = kEpiRowMajor, ga.out = dapart, ga.ld = p.g.E_pad, ga.rows_alloc = p.g.J_pad;
    ga.rows_valid = p.g.J_pad, ga.cols_valid = p.g.E_pad;
    e = launch_gemm(true, ga, s);
    if (e != hipSuccess) return psf_internal_fail((int)e, hipGetErrorString(e));
  }
  hipLaunchKernelGGL(wide_reduce_small_k, dim3((kOtMax * 1024 + 32 + kOtMax * 32 + 255) / 256, (unsigned)(p.g.J / 32)), dim3(256), 0, s, ba);
  hipLaunchKernelGGL(wide_reduce_da_k, dim3((unsigned)(((int64_t)p.g.J * (E / 4) + 255) / 256)), dim3(256), 0, s, ba);
  e = hipGetLastError();
  return e == hipSuccess ? PSF_OK : psf_internal_fail((int)e, hipGetErrorString(e));
}

}  // extern "C"
