// fwd_mlp_step.h — forward chord step whose W tile is COMPUTED on chip from the tile's rows of `data`: W never exists in
// memory (the reference produces W_m = fs[m](data) and hands it to spmm through memory,
// SyntheticExperiments/psf.py:175-188, LRA/psf.py:227-240).
//
//   W[p,:] = B_m GELU(A_m x_p + a_m) + b_m            MLPBlock, psf.py:35-60 — on the bf16 matrix pipe at f32 accuracy
//   out[b,p,:] = sum_k W[p,k] * V[b,(p+off_k) mod N,:] (+ res[b,p,:])          spmul/spmul_cuda.cu:20-27
//
// One workgroup = one tile of TR consecutive rows of one sequence, exactly the tile of fwd_window.h (LDS window for the near
// links, far links from L2, residual fused into the store, links ascending with uncontracted multiply and add). What differs
// is where the tile's W rows come from: instead of a DMA of TR*L floats, each wave takes TPW token tiles of 32 rows, loads
// their `data` rows straight into MFMA operand order (lane = token, 8 consecutive features per lane and k-step), splits them
// exactly into three bf16 terms and runs the two GEMMs of the step's MLP as mlp_fwd_x3.hip does (six-term products, GELU on
// the accumulator registers, accumulator-as-operand for the second layer); the finished Y^T tile is written to LDS as W rows
// of WS floats (16-byte aligned, conflict-free for the stores and for the row reads of the accumulate phase).
//
// Bytes per row-step: 4 (E + 2C + [res] C) instead of 4 (L + 2C + [res] C) — the step reads the 128-byte data row instead of
// the 60-byte W row — but the producer kernel's 4 (E + M L) bytes per token (read data, write every W_m) disappear, and
// its arithmetic now runs beside a kernel that is waiting for memory.
//
// Narrow rows only (C <= 32: TGS <= 3), E <= 32, h <= 128, L <= 20; tiles with rows >= N or missing channel groups take the
// EDGE instance (clamped loads, predicated store).
#pragma once

#include "fwd_mlp_step_launch.h"
#include "fwd_window.h"
#include "mlp_x3_image.h"
#include "psf_common.h"

namespace psf {

template <int L, int TGS>
struct MlpStepCfg {
  static constexpr int NT = 256, NW = 4;
  static constexpr int R = mlp_step_rows(TGS);
  static constexpr int TG = 1 << TGS;
  static constexpr int RS = NT >> TGS;
  static constexpr int TR = RS * R;
  static constexpr int WR = 2 * TR;
  static constexpr int KN = imin(L, ilog2_floor(TR) + 2);
  static constexpr int NF = L - KN;
  static constexpr int win_vecs = WR * TG;
  static constexpr int win_bytes = win_vecs * 16;
  static constexpr int WS = L <= 12 ? 12 : 20;  // floats per W row in LDS: 12 r and 20 r (mod 32 and mod 64) are conflict-free
  static constexpr int w_bytes = TR * WS * 4;
  static constexpr int TT = TR / 32;                     // token tiles per workgroup tile
  static constexpr int G = TT >= NW ? 1 : NW / TT;       // waves per token tile: they share its hidden units
  static constexpr int TPW = TT >= NW ? TT / NW : 1;     // token tiles per wave
  static constexpr int part_bytes = G > 1 ? (G - 1) * TT * 32 * WS * 4 : 0;
  static constexpr int img_off = win_bytes + w_bytes + part_bytes;
  static_assert(TR % 32 == 0 && (TT >= NW ? TT % NW == 0 : NW % TT == 0), "token tiles divide among the waves");
  static_assert(win_vecs % NT == 0, "window slots are a whole number of passes");
  static_assert(L <= 20, "WS covers 20 links");
};



// A lane's eight consecutive features e0 .. e0+7 of row `row` of `data` for sequence b, by the input recipe (MixerIn,
// fwd_mlp_step_launch.h): read from X, or computed from what X itself was computed from — the affine layer of the Adding /
// Temporal-Order networks (SyntheticExperiments/psf.py:153-154: init_linear) or the token embedding plus positional row
// (psf.py:151-152,157-162; LRA/psf.py:204-209) — so that `data` does not have to exist in memory either.
template <int KIND>  // compiled per recipe: with the three of them behind a run-time branch in one kernel the step took 7 % longer
__device__ __forceinline__ void data_row8(const MixerIn& in, const float* sAff, int b, int row, int N, int E, int e0, bool skip,
                                          float (&v)[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = 0.f;
  if (skip) return;
  constexpr int kind = KIND;  // (profiles/r04k_mixer_kind_ab.log: 884 us with in.kind read at run time, 826 compiled in)
  if constexpr (kind == 0) {
    const float* __restrict__ xr = reinterpret_cast<const float*>(in.src) + ((int64_t)b * N + row) * E;
    if (e0 < E) {
      const float4 lo = *reinterpret_cast<const float4*>(xr + e0);
      v[0] = lo.x, v[1] = lo.y, v[2] = lo.z, v[3] = lo.w;
    }
    if (e0 + 4 < E) {
      const float4 hi = *reinterpret_cast<const float4*>(xr + e0 + 4);
      v[4] = hi.x, v[5] = hi.y, v[6] = hi.z, v[7] = hi.w;
    }
    return;
  }
  if constexpr (kind == 1) {  // x W_i^T + b_i with K <= 3 inputs per position; sAff[e] = {w_e0, w_e1, w_e2, b_e}
    const float* __restrict__ xin = reinterpret_cast<const float*>(in.src) + ((int64_t)b * N + row) * in.K;
    const float x0 = xin[0], x1 = in.K > 1 ? xin[1] : 0.f, x2 = in.K > 2 ? xin[2] : 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
      if (e0 + i < E) {
        const float4 w = *reinterpret_cast<const float4*>(sAff + 4 * (e0 + i));
        v[i] = __fadd_rn(fmaf(x2, w.z, fmaf(x1, w.y, __fmul_rn(x0, w.x))), w.w);  // the products summed, then the bias
      }
  } else {  // table[token]
    int64_t tok = reinterpret_cast<const int64_t*>(in.src)[(int64_t)b * N + row];
    tok = tok < 0 ? 0 : (tok >= in.K ? in.K - 1 : tok);  // never read outside the table (as psf_embed_tokens_f32 clamps)
    const float* __restrict__ tr = in.weight + tok * E;
    if (e0 < E) {
      const float4 lo = *reinterpret_cast<const float4*>(tr + e0);
      v[0] = lo.x, v[1] = lo.y, v[2] = lo.z, v[3] = lo.w;
    }
    if (e0 + 4 < E) {
      const float4 hi = *reinterpret_cast<const float4*>(tr + e0 + 4);
      v[4] = hi.x, v[5] = hi.y, v[6] = hi.z, v[7] = hi.w;
    }
  }
  if (in.pos != nullptr) {  // + pos[p] (one rounded add, as psf_embed_tokens_f32)
    const float* __restrict__ pr = in.pos + (int64_t)row * E;
    if (e0 < E) {
      const float4 lo = *reinterpret_cast<const float4*>(pr + e0);
      v[0] = __fadd_rn(v[0], lo.x), v[1] = __fadd_rn(v[1], lo.y), v[2] = __fadd_rn(v[2], lo.z), v[3] = __fadd_rn(v[3], lo.w);
    }
    if (e0 + 4 < E) {
      const float4 hi = *reinterpret_cast<const float4*>(pr + e0 + 4);
      v[4] = __fadd_rn(v[4], hi.x), v[5] = __fadd_rn(v[5], hi.y), v[6] = __fadd_rn(v[6], hi.z), v[7] = __fadd_rn(v[7], hi.w);
    }
  }
}

// the affine recipe's table {w_e0, w_e1, w_e2, b_e} per feature e, 512 bytes of LDS (written before the first barrier)
__device__ __forceinline__ void stage_affine(const MixerIn& in, float* sAff, int E, int tid) {
  if (in.kind == 1 && tid < 32) {
    float4 w = make_float4(0.f, 0.f, 0.f, 0.f);
    if (tid < E) {
      w.x = in.weight[tid * in.K];
      if (in.K > 1) w.y = in.weight[tid * in.K + 1];
      if (in.K > 2) w.z = in.weight[tid * in.K + 2];
      w.w = in.bias ? in.bias[tid] : 0.f;
    }
    *reinterpret_cast<float4*>(sAff + 4 * tid) = w;
  }
}

// One token tile through the hidden units [u0, u1) of the MLP whose images sit at sImg: returns Y^T[o][tok] (register r of
// lane (tok = c, half) is output o = (r & 3) + 8 (r >> 2) + 4 half), starting from the output bias when `with_bias`.
// Weight fragments are read from the LDS image where they are used (registers: see the kernel's note on occupancy).
__device__ __forceinline__ psf_x3::f32x16 mlp_tile(const unsigned char* sImg, int u0, int u1, bool with_bias,
                                                   const float (&xv)[2][8], int c, int half) {
  using namespace psf_x3;
  auto bias4 = [&](const float* base, int q) {  // registers 4q..4q+3 are rows 8q + 4 half + (0..3)
    return *reinterpret_cast<const float4*>(base + 8 * q + 4 * half);
  };
  Frag3 xf[2];
#pragma unroll
  for (int s = 0; s < 2; ++s) xf[s] = split_pack8_pk(xv[s]);
  f32x16 acc2;
  {
    const float* sb = reinterpret_cast<const float*>(sImg + kOffSb);  // the output bias is in every unit's image
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v = with_bias ? bias4(sb, q) : make_float4(0.f, 0.f, 0.f, 0.f);
      acc2[4 * q] = v.x, acc2[4 * q + 1] = v.y, acc2[4 * q + 2] = v.z, acc2[4 * q + 3] = v.w;
    }
  }
  for (int u = u0; u < u1; ++u) {
    const unsigned char* img = sImg + u * kImgBytes;
    const float* sa = reinterpret_cast<const float*>(img + kOffSa);
    f32x16 acc1;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 v = bias4(sa, q);
      acc1[4 * q] = v.x, acc1[4 * q + 1] = v.y, acc1[4 * q + 2] = v.z, acc1[4 * q + 3] = v.w;
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const unsigned char* pa = img + c * kARow + 32 * s + 16 * half;
      Frag3 wa;
      wa.t1 = *reinterpret_cast<const bf16x8*>(pa);
      wa.t2 = *reinterpret_cast<const bf16x8*>(pa + kATerm);
      wa.t3 = *reinterpret_cast<const bf16x8*>(pa + 2 * kATerm);
      acc1 = mfma6(wa, xf[s], acc1);
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {  // GELU + split of registers 8s..8s+7: the B fragment of k-step s
      float gl[8];
#pragma unroll
      for (int i = 0; i < 8; i += 2) {
        const f32x2 y = gelu2(f32x2{acc1[8 * s + i], acc1[8 * s + i + 1]});
        gl[i] = y.x;
        gl[i + 1] = y.y;
      }
      const unsigned char* pb = img + kOffB + ((s * 2 + half) * 32 + c) * 16;
      Frag3 wb;
      wb.t1 = *reinterpret_cast<const bf16x8*>(pb);
      wb.t2 = *reinterpret_cast<const bf16x8*>(pb + kBTerm);
      wb.t3 = *reinterpret_cast<const bf16x8*>(pb + 2 * kBTerm);
      acc2 = mfma6(wb, split_pack8_pk(gl), acc2);
    }
  }
  return acc2;
}

template <int L, int TGS, bool RES, bool EDGE>
__global__ void __launch_bounds__(256, 3)
chord_fwd_mlp_k(const MixerIn in, const float* __restrict__ V, const float* __restrict__ res,
                float* __restrict__ out, const unsigned char* __restrict__ images, const int nu, const int E, const Geom gm,
                const Offsets offs, const int ablate_arg) {
  using namespace psf_x3;
  using Cfg = MlpStepCfg<L, TGS>;
#ifdef PSF_MIXER_ABLATE_LAB  // timing experiments (tuning key "mixer_ablate"): compiled out of the product, where the key is ignored
  const int ablate = ablate_arg;
#else
  constexpr int ablate = 0;
  (void)ablate_arg;
#endif
  constexpr int NT = Cfg::NT, R = Cfg::R, TG = Cfg::TG, RS = Cfg::RS, TR = Cfg::TR, KN = Cfg::KN, NF = Cfg::NF;
  constexpr int WS = Cfg::WS, TT = Cfg::TT, G = Cfg::G, TPW = Cfg::TPW;
  using V4 = Vec<float, 4>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  V4* __restrict__ sWin = reinterpret_cast<V4*>(smem);
  float* __restrict__ sW = reinterpret_cast<float*>(smem + Cfg::win_bytes);
  float* __restrict__ sPart = reinterpret_cast<float*>(smem + Cfg::win_bytes + Cfg::w_bytes);
  unsigned char* __restrict__ sImg = reinterpret_cast<unsigned char*>(smem + Cfg::img_off);

  int b, tile, chunk;
  decode_block(gm, b, tile, chunk);  // chunk == 0: rows of <= 32 channels are never split
  const int p0 = tile * TR;
  const int tid = threadIdx.x, lane = tid & 63, wave64 = tid & ~63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);  // provably wave-uniform: the branches on it are scalar
  const int c = lane & 31, half = lane >> 5;  // MLP phase: token column, k half
  const int g = tid & (TG - 1), rs = tid >> TGS;  // accumulate phase: channel group, row slot
  const int N = gm.N, C = gm.C;
  const bool cg_ok = !EDGE || g < gm.CG;
  const int cgc = cg_ok ? g : gm.CG - 1;
  const float* __restrict__ Vb = V + (int64_t)b * gm.v_bstride;

  // ---- (0a) this wave's rows of `data`, straight into B-operand order: k-step s covers e = 16 s + 8 half + (0..7) ----
  const int grp = G == 1 ? 0 : wv / TT;  // which share of the hidden units this wave takes (wave-uniform)
  // (`data` is given as its rows here. The recipes of psf_mixer_input — affine input layer, embedding lookup — are evaluated
  // by the single-launch kernel of mixer_lds.h only: inside the per-step kernels they measured slower than rows written once,
  // r04h_mixer_bench.log, and tripled this unit's instance count; round 5 took them out.)
  constexpr int KIND = 0;
  float* const sAff = nullptr;
  float xv[TPW][2][8];
#pragma unroll
  for (int tp = 0; tp < TPW; ++tp) {
    const int t = G == 1 ? wv * TPW + tp : wv % TT;
    int row = p0 + 32 * t + c;
    if (EDGE && row >= N) row = N - 1;
    if constexpr (KIND == 0 && !EDGE) {
      // rows of 32 features in a full tile: one scalar block address per token tile plus one lane offset (the row c, the
      // lane's 8-feature group), four unconditional 16-byte loads (other widths keep the predicated form below)
      if (E == 32 && !(ablate & 8)) {
        const PSF_GLOBAL char* blk = sbase(reinterpret_cast<const char*>(in.src) + ((int64_t)b * N + p0 + 32 * t) * (32 * 4));
        const uint32_t xo = (uint32_t)c * 128u + (uint32_t)half * 32u;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const Vec<float, 4> lo = ldg<float, 4>(blk + (xo + 64u * s)), hi = ldg<float, 4>(blk + (xo + 64u * s + 16u));
#pragma unroll
          for (int i = 0; i < 4; ++i) xv[tp][s][i] = lo.e[i], xv[tp][s][4 + i] = hi.e[i];
        }
        continue;
      }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) data_row8<KIND>(in, sAff, b, row, N, E, 16 * s + 8 * half, (ablate & 8) != 0, xv[tp][s]);
  }

  // ---- (0b) unit images of this step's MLP and the V window [p0, p0 + 2 TR) mod N, by LDS-DMA ----
  const int img_vecs = nu * kImgVecs;
  for (int v0 = 0; v0 < img_vecs; v0 += NT) {
    const int v = v0 + tid;
    if (v < img_vecs)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(images + 16 * (size_t)v),
                                       (__attribute__((address_space(3))) void*)(sImg + 16 * (v0 + wave64)), 16, 0, 0);
  }
  V4 far[R][NF > 0 ? NF : 1];
  V4 rres[R];
  // Full-tile launches (!EDGE; host-checked: N and every far offset multiples of TR, C = 4 TG, N C 4 < 2^31): every row block
  // the workgroup touches is TR-aligned and never wraps inside — scalar block addresses plus one lane offset per row slot
  // (fwd_window.h says what that saves; here the vector unit is the bottleneck, so it counts double).
  constexpr uint32_t rowB = TG * 16u;
  uint32_t voff[R];
#pragma unroll
  for (int j = 0; j < R; ++j) voff[j] = lane_off((uint32_t)(j * RS + rs) * rowB + (uint32_t)g * 16u);  // (not the image loop's tid * 16)
  const char* __restrict__ Vbb = reinterpret_cast<const char*>(Vb);
  if constexpr (!EDGE) {
    int p1 = p0 + TR;
    if (p1 >= N) p1 -= N;
#pragma unroll
    for (int n = 0; n < Cfg::win_vecs / NT; ++n)
      stage16g<0>(sbase(Vbb + (uint32_t)(n / R == 0 ? p0 : p1) * rowB) + voff[n % R], sWin + n * NT + wave64);
    // (requesting the far rows behind the barrier instead — nothing reads them before the matrix phase is over — measured
    // no gain: 752 vs 747 us per mixer forward at cfg2, 516 vs 499 in the Order shape, one process, five rounds)
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      int s0 = p0 + offs.v[KN + f];
      if (s0 >= N) s0 -= N;
      if (ablate & 4) s0 = p0;  // timing only: the tile's own rows (L2 / window hits) instead of the far ones
      const PSF_GLOBAL char* blk = sbase(Vbb + (uint32_t)s0 * rowB);
#pragma unroll
      for (int j = 0; j < R; ++j) far[j][f] = ldg<float, 4>(blk + voff[j]);
    }
  } else {
#pragma unroll
    for (int n = 0; n < Cfg::win_vecs / NT; ++n) {
      const int i = n * NT + tid;
      const int wr = i >> TGS, gg = i & (TG - 1);
      int src = p0 + wr;
      if (src >= N) src -= N;
      if (!EDGE || gg < gm.CG)
        stage16<float, 4, true>(Vb + (int64_t)src * C + (int64_t)gg * 4, sWin + n * NT + wave64, lane);
    }

    // (0c) far rows -> registers
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const int pr = p0 + j * RS + rs;
      const int p = EDGE ? imin(pr, N - 1) : pr;
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        int src = p + offs.v[KN + f];
        if (src >= N) src -= N;
        if (ablate & 4) src = p;  // timing only: the row itself (an L2 / window hit) instead of the far row
        far[j][f] = ld<float, 4>(Vb + (int64_t)src * C + (int64_t)cgc * 4);
      }
    }
  }

  __syncthreads();  // images and window have landed (hipcc drains vmcnt before the barrier)

  // ---- (1) the tile's W rows: W^T[o][tok] = b + sum over this wave's hidden units of B_u GELU(A_u X^T + a_u) ----
  // One token tile at a time, weight fragments read from the LDS image where they are used: the phase then needs ~100
  // registers beside the far rows it keeps alive, which is what lets three workgroups share a CU (<= 168 registers) —
  // two tiles in flight with the fragments hoisted took 228 (two workgroups per CU: profiles/r04c_mixer_ablate.log shows
  // the memory, data-row and matrix phases of the step adding up instead of overlapping).
  const int per = (nu + G - 1) / G;
  const int u0 = imin(nu, grp * per), u1 = imin(nu, u0 + per);
  f32x16 acc2[TPW];
#pragma unroll
  for (int tp = 0; tp < TPW; ++tp) {
    acc2[tp] = mlp_tile(sImg, (ablate & 1) ? u1 : u0, u1, grp == 0, xv[tp], c, half);  // ablate & 1 (timing only): W = bias
    if constexpr (G == 1) {  // the finished tile goes to LDS at once (its registers are free for the next tile)
      constexpr int NQ1 = (L + 7) / 8;
      float* dst = sW + (32 * (wv * TPW + tp) + c) * WS;
#pragma unroll
      for (int q = 0; q < NQ1; ++q)
        if (8 * q + 4 * half < L)
          *reinterpret_cast<float4*>(dst + 8 * q + 4 * half) =
              make_float4(acc2[tp][4 * q], acc2[tp][4 * q + 1], acc2[tp][4 * q + 2], acc2[tp][4 * q + 3]);
      asm volatile("" ::: "memory");  // the next tile's LDS reads stay behind this tile's stores (no hoisting across tiles)
    }
  }

  // The lane holds W^T[o = 8 q + 4 half + (0..3)][tok = c] in registers 4 q .. 4 q + 3: one 16-byte LDS store per q.
  constexpr int NQ = (L + 7) / 8;  // q groups that hold links < L (for either half)
  if constexpr (G > 1) {           // several waves share a token tile: the others hand their partial sums over through LDS
    if (grp > 0 && u0 < u1) {
      float* dst = sPart + (((grp - 1) * TT + wv % TT) * 32 + c) * WS;
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        if (8 * q + 4 * half < L)
          *reinterpret_cast<float4*>(dst + 8 * q + 4 * half) =
              make_float4(acc2[0][4 * q], acc2[0][4 * q + 1], acc2[0][4 * q + 2], acc2[0][4 * q + 3]);
    }
    __syncthreads();
    if (grp == 0) {
      for (int og = 1; og < G; ++og) {
        if (og * per >= nu) break;  // that group had no units (wave-uniform)
        const float* src = sPart + (((og - 1) * TT + wv % TT) * 32 + c) * WS;
        using F4 = float __attribute__((ext_vector_type(4)));
        F4 v[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q)
          v[q] = 8 * q + 4 * half < L ? *reinterpret_cast<const F4*>(src + 8 * q + 4 * half) : F4{0.f, 0.f, 0.f, 0.f};
        lds_wait_all();  // (the same rule: the partial sums are added behind a full wait)
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
          behind_wait(v[q]);
          acc2[0][4 * q] += v[q].x, acc2[0][4 * q + 1] += v[q].y, acc2[0][4 * q + 2] += v[q].z, acc2[0][4 * q + 3] += v[q].w;
        }
      }
    }
    if (grp == 0) {  // (G > 1 means one token tile per wave)
      float* dst = sW + (32 * (wv % TT) + c) * WS;
#pragma unroll
      for (int q = 0; q < NQ; ++q)
        if (8 * q + 4 * half < L)
          *reinterpret_cast<float4*>(dst + 8 * q + 4 * half) =
              make_float4(acc2[0][4 * q], acc2[0][4 * q + 1], acc2[0][4 * q + 2], acc2[0][4 * q + 3]);
    }
  }
  if constexpr (RES) {  // the residual rows: requested here, behind the matrix phase (8 registers it does not have to carry)
    if constexpr (!EDGE) {
      const PSF_GLOBAL char* rb = sbase(reinterpret_cast<const char*>(res + ((int64_t)b * N + p0) * C));
#pragma unroll
      for (int j = 0; j < R; ++j) rres[j] = ldg<float, 4>(rb + lane_off(voff[j]));
    } else {
#pragma unroll
      for (int j = 0; j < R; ++j) {
        const int pr = p0 + j * RS + rs;
        const int p = imin(pr, N - 1);
        rres[j] = ld<float, 4>(res + ((int64_t)b * N + p) * C + (int64_t)cgc * 4);
      }
    }
  }
  __syncthreads();

  // ---- (2) accumulate, links ascending (the arithmetic and order of fwd_window.h) ----
  // (All R rows are finished before the stores are issued; the stores are the last instructions of the kernel.)
  V4 acc[R];
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const int pl = j * RS + rs;
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[j].e[i] = 0.f;
    const float* __restrict__ wrow = sW + pl * WS;
    // Every LDS operand of the row is in registers, and waited for in full, before the first multiply. With the reads
    // consumed behind counted waits (s_waitcnt lgkmcnt(6), (4), ... as hipcc schedules them) this phase sporadically came out
    // wrong in dword 0 or dword 2 of lanes 48..63 of one wave — 21 to 59 of 300 launches at N = 16384, C = 32, B = 16, with
    // the LDS window, the W tile and the far rows all verified intact in the same launch (profiles/r04b_mixer_lds_wait.md).
    V4 xs[KN];
    float wk[L];
#pragma unroll
    for (int k = 0; k < KN; ++k) xs[k] = sWin[((pl + chord_off(k)) << TGS) + g];
#pragma unroll
    for (int k = 0; k < L; ++k) wk[k] = wrow[k];
    lds_wait_all();  // psf_common.h, "LDS results in kernels that also issue MFMAs": the consumers below depend on this wait
#pragma unroll
    for (int k = 0; k < KN; ++k) behind_wait(xs[k]);
#pragma unroll
    for (int k = 0; k < L; ++k) behind_wait(wk[k]);
    __builtin_amdgcn_sched_barrier(0);
    if (ablate & 2) {  // timing only: operands consumed, no multiply-add chain
#pragma unroll
      for (int k = 0; k < KN; ++k) acc[j].e[k & 3] += xs[k].e[k & 3] + wk[k];
    } else
#pragma unroll
    for (int k = 0; k < KN; ++k) axpy_rn<float, 4>(acc[j], wk[k], xs[k]);
#pragma unroll
    for (int f = 0; f < NF; ++f) axpy_rn<float, 4>(acc[j], wk[KN + f], far[j][f]);
    if constexpr (RES) {
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[j].e[i] = add_rn(acc[j].e[i], rres[j].e[i]);
    }
  }
  __builtin_amdgcn_sched_barrier(0);  // nothing but the stores below this line: no later instruction can reuse their registers
  if constexpr (!EDGE) {
    PSF_GLOBAL char* ob = sbase(reinterpret_cast<char*>(out + ((int64_t)b * N + p0) * C));
#pragma unroll
    for (int j = 0; j < R; ++j) stg<float, 4>(ob + lane_off(voff[j]), acc[j]);
  } else {
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const int p = p0 + j * RS + rs;
      if (p < N && cg_ok) st<float, 4>(out + ((int64_t)b * N + p) * C + (int64_t)g * 4, acc[j]);
    }
  }
}

// V0 = g(data) on the same tiles: the matrix phase only (MLP 0: E -> h -> C), its Y^T tiles stored as rows of V0. A wave takes
// whole token tiles with all hidden units (waves beyond the tile count idle: one launch per forward, not the hot kernel).
template <int TGS, bool EDGE>
__global__ void __launch_bounds__(256, 3)
chord_mixer_g_k(const MixerIn in, float* __restrict__ out, const unsigned char* __restrict__ images, const int nu, const int E,
                const Geom gm) {
  using namespace psf_x3;
  constexpr int TR = mlp_step_tile_rows(TGS), TT = TR / 32, NW = 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned char* __restrict__ sImg = reinterpret_cast<unsigned char*>(smem);
  constexpr int KIND = 0;  // rows of `data` only (see chord_fwd_mlp_k)
  float* const sAff = nullptr;
  int b, tile, chunk;
  decode_block(gm, b, tile, chunk);
  const int p0 = tile * TR, N = gm.N, C = gm.C;
  const int tid = threadIdx.x, lane = tid & 63, wave64 = tid & ~63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 31, half = lane >> 5;
  const int img_vecs = nu * kImgVecs;
  for (int v0 = 0; v0 < img_vecs; v0 += 256) {
    const int v = v0 + tid;
    if (v < img_vecs)
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(images + 16 * (size_t)v),
                                       (__attribute__((address_space(3))) void*)(sImg + 16 * (v0 + wave64)), 16, 0, 0);
  }
  // the wave's token tiles (t = wv, wv + 4, ...): every tile's data rows are requested before the first barrier, so that their
  // memory latency passes beside the image DMA instead of once per tile (the first form loaded inside the loop: 57 us for a
  // launch that moves 42 MB)
  constexpr int TPWG = (TT + NW - 1) / NW;
  float xv[TPWG][2][8];
#pragma unroll
  for (int i = 0; i < TPWG; ++i) {
    const int t = wv + i * NW;
    const int row = p0 + 32 * (t < TT ? t : 0) + c;
    const int rowc = (EDGE && row >= N) ? N - 1 : row;
    if constexpr (KIND == 0 && !EDGE) {  // rows of 32 features in a full tile: scalar block address + one lane offset (as the step kernel)
      if (E == 32) {
        const PSF_GLOBAL char* blk =
            sbase(reinterpret_cast<const char*>(in.src) + ((int64_t)b * N + p0 + 32 * (t < TT ? t : 0)) * (32 * 4));
        const uint32_t xo = (uint32_t)c * 128u + (uint32_t)half * 32u;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const Vec<float, 4> lo = ldg<float, 4>(blk + (xo + 64u * s)), hi = ldg<float, 4>(blk + (xo + 64u * s + 16u));
#pragma unroll
          for (int q = 0; q < 4; ++q) xv[i][s][q] = lo.e[q], xv[i][s][4 + q] = hi.e[q];
        }
        continue;
      }
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) data_row8<KIND>(in, sAff, b, rowc, N, E, 16 * s + 8 * half, false, xv[i][s]);
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < TPWG; ++i) {
    const int t = wv + i * NW;
    if (t >= TT) break;  // wave-uniform
    const int row = p0 + 32 * t + c;
    const f32x16 y = mlp_tile(sImg, 0, nu, true, xv[i], c, half);
    if (!EDGE || row < N) {
      float* __restrict__ orow = out + ((int64_t)b * N + row) * C;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (8 * q + 4 * half < C)  // C is a multiple of 4: whole 16-byte groups
          *reinterpret_cast<float4*>(orow + 8 * q + 4 * half) = make_float4(y[4 * q], y[4 * q + 1], y[4 * q + 2], y[4 * q + 3]);
    }
  }
}

}  // namespace psf
