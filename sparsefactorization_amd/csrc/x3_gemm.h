// x3_gemm.h — f32-accurate GEMM on the bf16 matrix pipe for the WIDE producer MLPs (ListOps: E = 512, 12 x h = 128;
// MLPBlock, LRA/psf.py:35-60, applied M + 1 times to the same input, LRA/psf.py:214,227).
//
// Every f32 operand is split exactly into three bf16 terms (mlp_x3_common.h) ONCE, by whoever produces it, and lives in
// memory as three bf16 "col16 planes":
//     plane[t][(col >> 4) * rows_pad + row][col & 15]          t = 0, 1, 2;  rows_pad a multiple of 256
// i.e. 16-column blocks, each a dense [rows_pad][16] array of 32-byte rows. One layout serves both roles an activation
// plays in training:
//   * the contraction runs over the COLUMNS (Hpre = X A^T over e; dX = G A over j): a k-step is one column block, a
//     tile's 256 rows of it are one contiguous 8 KB burst and the MFMA operand fragment of a lane (8 consecutive k of its
//     row) is one ds_read_b128 of a linear LDS image                                               -> "NT" kernel
//   * the contraction runs over the ROWS (dA = G^T X over tokens): a k-step is 16 rows, a tile is 16 column blocks of
//     512 contiguous bytes, and the fragment (8 consecutive k of the lane's COLUMN) comes out of the same bytes through
//     two ds_read_b64_tr_b16 (the hardware transposes 4 x 16 blocks on the way to the registers)   -> "TN" kernel
// so X and G are split once and never transposed in memory.
//
// Kernel: 256 x 256 output tile per 512-thread workgroup (8 waves as 2 x 4; a wave owns 128 x 64 = 4 x 2 accumulator
// tiles of v_mfma_f32_32x32x16_bf16: 128 registers; a 256 x 64 configuration serves narrow column dimensions), one
// workgroup per CU. A k16 chunk of both operands is 6 planes x
// 8 KB = 48 KB; three LDS stages, filled by LDS-DMA (global_load_lds_dwordx4, no registers) two chunks ahead of the
// chunk being multiplied; per chunk ONE counted s_waitcnt vmcnt + ONE s_barrier:
//     wait (own DMA of chunk i) -> barrier (everyone's DMA of chunk i landed, everyone done reading chunk i-1)
//     -> issue DMA of chunk i+2 into the stage chunk i-1 used -> 18 fragment reads + 48 MFMAs on chunk i.
// Workgroups are persistent (one per CU, items dealt round-robin) and the ring runs across item boundaries.
// The six product terms of a fragment pair reuse the same six fragment registers, so LDS traffic per MFMA is half of a
// plain bf16 GEMM's at the same tile.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mlp_planes.h"
#include "mlp_x3_common.h"

namespace psf_wide {

using namespace psf_x3;

constexpr int kTile = 256;                     // output tile rows (and columns of the square configuration)
constexpr int kTileNarrow = 64;                // output tile columns of the narrow configuration
constexpr int kStages = 3;
constexpr int kGemmThreads = 512;

// Wave decomposition of a 512-thread workgroup: WM x WN waves, each MT x NTL accumulator tiles of 32 x 32.
//   square  2 x 4 waves of 4 x 2 tiles: 256 x 256 (128 accumulator registers per lane)
//   narrow  8 x 1 waves of 1 x 2 tiles: 256 x  64 — for a column dimension of <= 128 (E = 64: the square tile would
//           spend three quarters of its MFMAs and B-operand loads on padding)
template <int WM_, int WN_, int MT_, int NTL_>
struct GemmShape {
  static constexpr int WM = WM_, WN = WN_, MT = MT_, NTL = NTL_;
  static constexpr int BM = WM * MT * 32, BN = WN * NTL * 32;
  static constexpr int plane_a = BM * 32, plane_b = BN * 32;  // bytes of one plane of one k16 chunk
  static constexpr int stage_bytes = 3 * (plane_a + plane_b);
  static_assert(WM * WN * 64 == kGemmThreads && BM == kTile && plane_b % 1024 == 0, "workgroup shape");
};
using GemmSquare = GemmShape<2, 4, 4, 2>;
using GemmNarrow = GemmShape<8, 1, 1, 2>;

struct Operand {
  const unsigned char* p[3];  // the three term planes
  int64_t rows_pad;           // rows per 16-column block
  int32_t blocks;             // 16-column blocks held
  int32_t pad_;
};

enum Epilogue : int32_t {
  kEpiRowMajor = 0,  // out[(split * rows_alloc + m) * ld + n] = D[m][n]   for m < rows_valid, n < cols_valid
  kEpiFragBias = 1   // out in accumulator-fragment order, + bias[m]: block (n >> 5, m >> 5) = [4][64 lanes][4 registers]
};

constexpr int kOutTilesMax = 4;    // output tiles of 32 of a second layer (O <= 128)
constexpr int kPack2Tile = 2048;   // one term of one (hidden unit, output tile) of the second-layer operand: [2 s][2 half][32 o][8] bf16
constexpr int kFusedMlpsMax = 24;

// The second layer inside the forward GEMM's epilogue (kEpiFragBias, square shape, every MLP exactly 128 padded hidden
// rows): a wave's 128 x 64 tile is ONE MLP x two token tiles, so Y^T = B GELU(Hpre^T) + b needs nothing but the wave's own
// accumulators (accumulator-as-operand). Y[k] == nullptr: MLP k is not fused (wide outputs) and only its Hpre is stored.
struct FusedOut {
  float* Y[kFusedMlpsMax];   // [T, O[k]] row-major
  int32_t O[kFusedMlpsMax];
  const unsigned char* pack2;  // [unit][kOutTilesMax][3 terms][kPack2Tile], operand order of wide_pack_fwd_k
  const float* b2;             // [k][kOutTilesMax][2 half][16]
  int64_t T;
  int32_t enabled;             // 0: plain kEpiFragBias
  int32_t store_hpre;          // 0 (inference): fused MLPs do not store their Hpre
};

// y = GELU(x) = x Phi(x); Phi by Abramowitz & Stegun 26.2.17 (|error| <= 7.5e-8), scalar f32 instructions (mlp_bwd.hip)
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

struct GemmArgs {
  Operand A, B;          // NT: rows = m (n), column blocks = k chunks.  TN: rows = k, column blocks = m (n) / 16
  int32_t tiles_m, tiles_n, splits;
  int32_t chunks;        // k16 chunks of the whole contraction
  int32_t n_fast;        // 1: consecutive workgroups walk n first (share the A tile), 0: m first
  int32_t epilogue;
  float* out;
  const float* bias;     // kEpiFragBias: [tiles_m * 256]
  int64_t ld;            // kEpiRowMajor: leading dimension (floats)
  int64_t rows_alloc;    // kEpiRowMajor with splits: rows per split slab
  int64_t rows_valid, cols_valid;
  FusedOut fo;           // kEpiFragBias only
};

// 16 bytes per lane global -> LDS; the wave's 64 lanes fill 1 KB at the LDS address in M0. Issued as inline assembly:
// hipcc's alias rule for __builtin_amdgcn_global_load_lds makes every later LDS read wait for vmcnt(0), which would
// serialise the pipeline (see mlp_bwd.hip). Address = SGPR base + 32-bit lane offset.
__device__ __forceinline__ void glds16(uint32_t lds_at, uint32_t voff, const unsigned char* base) {
  uint32_t m0_saved;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, %3\n\ts_mov_b32 m0, %0"
      : "=&s"(m0_saved)
      : "s"(lds_at), "v"(voff), "s"(base)
      : "memory");
}

template <bool TN, class S = GemmSquare>
__global__ void __launch_bounds__(kGemmThreads, 1) x3_gemm_k(const GemmArgs g) {
  constexpr int MT = S::MT, NTL = S::NTL, BM = S::BM, BN = S::BN;
  constexpr int kStageBytes = S::stage_bytes, kPlaneA = S::plane_a, kPlaneB = S::plane_b;
  __shared__ __attribute__((aligned(1024))) unsigned char lds[kStages * kStageBytes];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wv / S::WN, wn = wv % S::WN;
  const int c = lane & 31, half = lane >> 5;
  const bool loads_b = wv * 1024 < kPlaneB;  // wave-uniform: the B plane of a chunk is kPlaneB / 1024 wave-loads

  // ---- work items = (tile, K split). Persistent workgroups: item = round * gridDim.x + XCD-aware bijective remap of the
  // block id (consecutive logical ids run on one XCD at the same time and share operand tiles in its L2). The LDS ring is
  // fed across item boundaries: the first two chunks of the NEXT item are in flight while this item's last chunks are
  // multiplied and its tile is stored, so only a workgroup's first item pays the load latency of an empty pipeline.
  uint32_t lb;
  {
    const uint32_t nb = gridDim.x, bid = blockIdx.x;
    const uint32_t xq = nb >> 3, xr = nb & 7, xcd = bid & 7, idx = bid >> 3;
    lb = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + idx;
  }
  const uint32_t per_split = (uint32_t)(g.tiles_m * g.tiles_n);
  const uint32_t items = per_split * (uint32_t)g.splits;
  const int per = (g.chunks + g.splits - 1) / g.splits;  // host guarantees every split a non-empty K range
  struct Item {
    int64_t m0, n0;
    int sp, k_begin, nk;
  };
  auto decode = [&](uint32_t item) {
    Item it;
    it.sp = (int)(item / per_split);
    const uint32_t rem = item - (uint32_t)it.sp * per_split;
    int tm, tn;
    if (g.n_fast) {
      tm = (int)(rem / (uint32_t)g.tiles_n);
      tn = (int)(rem - (uint32_t)tm * (uint32_t)g.tiles_n);
    } else {
      tn = (int)(rem / (uint32_t)g.tiles_m);
      tm = (int)(rem - (uint32_t)tn * (uint32_t)g.tiles_m);
    }
    it.m0 = (int64_t)tm * BM, it.n0 = (int64_t)tn * BN;
    it.k_begin = it.sp * per;
    const int k_end = it.k_begin + per < g.chunks ? it.k_begin + per : g.chunks;
    it.nk = k_end - it.k_begin;
    return it;
  };

  // ---- loader: per-lane offsets (constant over an item's chunks) and per-chunk scalar bases of the ISSUE cursor
  uint32_t voff_a, voff_b;
  int64_t step_a, step_b;  // bytes from one chunk's base to the next
  const unsigned char* base_a[3];
  const unsigned char* base_b[3];
  uint32_t item_i = lb;  // the item whose chunks are being requested, and how many of them are left
  int left_i = 0;
  auto aim = [&](const Item& it) {  // point the issue cursor at the first chunk of an item
    if constexpr (!TN) {
      voff_a = voff_b = (uint32_t)tid * 16u;
      step_a = g.A.rows_pad * 32;
      step_b = g.B.rows_pad * 32;
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        base_a[t] = g.A.p[t] + ((int64_t)it.k_begin * g.A.rows_pad + it.m0) * 32;
        base_b[t] = g.B.p[t] + ((int64_t)it.k_begin * g.B.rows_pad + it.n0) * 32;
      }
    } else {
      // LDS image of an operand plane: [16 column blocks][16 k rows][32 B]; the 128-byte groups of four k rows are swapped
      // pairwise in odd blocks (source-side swizzle) so that the two blocks a transposed read touches in one LDS cycle
      // fall on different banks.
      const int b = tid >> 5, q = tid & 31, r_lin = q >> 1, h16 = q & 1;
      const int r = ((((r_lin >> 2) ^ (b & 1)) << 2) | (r_lin & 3));
      const int64_t blk_a = it.m0 / 16 + b < g.A.blocks ? it.m0 / 16 + b : g.A.blocks - 1;  // clamped: tiles past the array
      const int64_t blk_b = it.n0 / 16 + b < g.B.blocks ? it.n0 / 16 + b : g.B.blocks - 1;  // re-read its last block
      voff_a = (uint32_t)((blk_a * g.A.rows_pad + r) * 32 + h16 * 16);
      voff_b = (uint32_t)((blk_b * g.B.rows_pad + r) * 32 + h16 * 16);
      step_a = step_b = 16 * 32;
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        base_a[t] = g.A.p[t] + (int64_t)it.k_begin * 512;
        base_b[t] = g.B.p[t] + (int64_t)it.k_begin * 512;
      }
    }
    left_i = it.nk;
  };
  if (item_i < items) aim(decode(item_i));
  const uint32_t lds0 = (uint32_t)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)lds);
  const uint32_t lds_wave = lds0 + (uint32_t)wv * 1024u;
  int in_flight = 0;  // chunks requested and not yet multiplied (0..2)
  auto issue = [&](int stage) {  // the issue cursor's chunk into LDS stage `stage`; the cursor moves on (block-uniform)
    if (left_i == 0) return;
    const uint32_t at = lds_wave + (uint32_t)stage * kStageBytes;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      glds16(at + t * kPlaneA, voff_a, base_a[t]);
      if (loads_b) glds16(at + 3 * kPlaneA + t * kPlaneB, voff_b, base_b[t]);
      base_a[t] += step_a;
      base_b[t] += step_b;
    }
    ++in_flight;
    if (--left_i == 0) {
      item_i += gridDim.x;
      if (item_i < items) aim(decode(item_i));
    }
  };

  // ---- fragment read offsets of this lane
  int fa[2], fb[2];
  if constexpr (!TN) {
    fa[0] = (wm * MT * 32 + c) * 32 + half * 16;   // + i * 1024
    fb[0] = (wn * NTL * 32 + c) * 32 + half * 16;  // + j * 1024
    fa[1] = fb[1] = 0;
  } else {
    const int cg = (lane >> 4) & 1, qq = (lane >> 2) & 3, p = lane & 3;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int o = cg * 512 + (((2 * half + t) ^ cg) * 128) + qq * 32 + 8 * p;
      fa[t] = wm * MT * 1024 + o;   // + i * 1024
      fb[t] = wn * NTL * 1024 + o;  // + j * 1024
    }
  }
  auto frag = [&](const unsigned char* plane, const int (&f)[2], int tile) -> bf16x8 {
    if constexpr (!TN) {
      return *reinterpret_cast<const bf16x8*>(plane + f[0] + tile * 1024);
    } else {
      return join8(tr_read(plane + f[0] + tile * 1024), tr_read(plane + f[1] + tile * 1024));
    }
  };

  issue(0);
  issue(1);
  int st = 0;
  for (uint32_t item_c = lb; item_c < items; item_c += gridDim.x) {
  const Item cur = decode(item_c);
  const int64_t m0 = cur.m0, n0 = cur.n0;
  const int sp = cur.sp;
  f32x16 acc[MT][NTL];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NTL; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  for (int i = 0; i < cur.nk; ++i) {
    // this wave's DMA of the chunk to multiply has landed; the chunk requested after it (six loads) may stay in flight.
    // (After an item boundary the previous tile's stores are younger than both: the counted wait then covers them too.)
    if (in_flight > 1) {
      if (loads_b) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");  // a wave that loads no B plane has three loads per chunk
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    --in_flight;
    issue(st == 0 ? 2 : st - 1);  // into the stage the previous chunk used
    const unsigned char* sA = lds + st * kStageBytes;
    const unsigned char* sB = sA + 3 * kPlaneA;
    Frag3 bf[NTL];
#pragma unroll
    for (int j = 0; j < NTL; ++j)
      bf[j] = Frag3{frag(sB, fb, j), frag(sB + kPlaneB, fb, j), frag(sB + 2 * kPlaneB, fb, j)};
#pragma unroll
    for (int ii = 0; ii < MT; ++ii) {
      const Frag3 af{frag(sA, fa, ii), frag(sA + kPlaneA, fa, ii), frag(sA + 2 * kPlaneA, fa, ii)};
#pragma unroll
      for (int j = 0; j < NTL; ++j) acc[ii][j] = mfma6(af, bf[j], acc[ii][j]);
    }
    st = st == 2 ? 0 : st + 1;
  }

  // ---- epilogue. Register r of lane (c, half) of tile (i, j) is
  //      D[m0 + (wm * MT + i) * 32 + cd_row(r, half)][n0 + (wn * NTL + j) * 32 + c].
  if (g.epilogue == kEpiFragBias) {
    const int64_t units = (int64_t)g.tiles_m * (BM / 32);
    // block (token tile, hidden unit) of Hpre = 4 KB: [q = r >> 2][lane][r & 3] — four 1 KB dwordx4 bursts per tile
    float* Yk = nullptr;
    int k_mlp = 0, O = 0;
    if constexpr (!TN && MT == 4) {
      if (g.fo.enabled) {
        k_mlp = (int)((m0 + wm * 128) >> 7);  // wave-uniform: the wave's 128 rows are MLP k_mlp's four units
        Yk = g.fo.Y[k_mlp];
        O = g.fo.O[k_mlp];
      }
    }
    const bool keep_hpre = Yk == nullptr || g.fo.store_hpre;
#pragma unroll
    for (int j = 0; j < NTL; ++j) {
      const int64_t ncol = n0 + (wn * NTL + j) * 32;
      if (ncol >= g.cols_valid) continue;  // wave-uniform
      f32x16 acc2;
      if (Yk) {
        const float4* bp = reinterpret_cast<const float4*>(g.fo.b2 + ((k_mlp * kOutTilesMax) * 2 + half) * 16);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 v = bp[q];
          acc2[4 * q] = v.x, acc2[4 * q + 1] = v.y, acc2[4 * q + 2] = v.z, acc2[4 * q + 3] = v.w;
        }
      }
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int64_t mrow = m0 + (wm * MT + i) * 32;
        float h[16];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 v = *reinterpret_cast<const float4*>(g.bias + mrow + 8 * q + 4 * half);
          h[4 * q] = acc[i][j][4 * q] + v.x, h[4 * q + 1] = acc[i][j][4 * q + 1] + v.y;
          h[4 * q + 2] = acc[i][j][4 * q + 2] + v.z, h[4 * q + 3] = acc[i][j][4 * q + 3] + v.w;
        }
        if (keep_hpre) {
          float4* dst = reinterpret_cast<float4*>(g.out + (((ncol >> 5) * units + (mrow >> 5)) * 16) * 64) + lane;
#pragma unroll
          for (int q = 0; q < 4; ++q) dst[q * 64] = make_float4(h[4 * q], h[4 * q + 1], h[4 * q + 2], h[4 * q + 3]);
        }
        if (Yk) {
#pragma unroll
          for (int r = 0; r < 16; ++r) h[r] = gelu_only(h[r]);
          const Split16 ys = split16(h);
          const unsigned char* wp = g.fo.pack2 + ((int64_t)((mrow >> 5) * kOutTilesMax) * 3) * kPack2Tile;
#pragma unroll
          for (int s2 = 0; s2 < 2; ++s2) {
            const unsigned char* wq = wp + ((s2 * 2 + half) * 32 + c) * 16;
            const Frag3 wb{*reinterpret_cast<const bf16x8*>(wq), *reinterpret_cast<const bf16x8*>(wq + kPack2Tile),
                           *reinterpret_cast<const bf16x8*>(wq + 2 * kPack2Tile)};
            const Frag3 hb{acc_frag(ys, 0, s2), acc_frag(ys, 1, s2), acc_frag(ys, 2, s2)};
            acc2 = mfma6(wb, hb, acc2);
          }
        }
      }
      if (Yk) {  // the lane holds Y^T[o = 8 q + 4 half + (0..3)][tok = ncol + c] in registers 4 q .. 4 q + 3
        const int64_t tok = ncol + c;
        const bool vec_ok = (O & 3) == 0 && (reinterpret_cast<uintptr_t>(Yk) & 15) == 0;
        if (tok < g.fo.T) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int o0 = 8 * q + 4 * half;
            float* yp = Yk + tok * O + o0;
            if (vec_ok) {
              if (o0 < O) *reinterpret_cast<float4*>(yp) = make_float4(acc2[4 * q], acc2[4 * q + 1], acc2[4 * q + 2], acc2[4 * q + 3]);
            } else {
#pragma unroll
              for (int i2 = 0; i2 < 4; ++i2)
                if (o0 + i2 < O) yp[i2] = acc2[4 * q + i2];
            }
          }
        }
      }
    }
  } else {
    float* out = g.out + (int64_t)sp * g.rows_alloc * g.ld;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NTL; ++j) {
        const int64_t ncol = n0 + (wn * NTL + j) * 32 + c;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int64_t mrow = m0 + (wm * MT + i) * 32 + cd_row(r, half);
          if (mrow < g.rows_valid && ncol < g.cols_valid) out[mrow * g.ld + ncol] = acc[i][j][r];
        }
      }
  }
  }  // items
}

// f32 [rows][cols] (row stride ld floats) -> three col16 planes; rows in [rows, rows_pad) are written as zeros.
// A wave takes 64 consecutive rows of one column block: each lane reads its row's 64 bytes and the wave writes 2 KB
// contiguous per plane.
struct SplitArgs {
  const float* src;
  unsigned char* p[3];
  int64_t rows, rows_pad, ld;
  int32_t blocks;  // cols / 16
};

__global__ void __launch_bounds__(256) x3_split_planes_k(const SplitArgs a) {
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int64_t row = ((int64_t)blockIdx.x) * 64 + lane;
  if (row >= a.rows_pad) return;
  for (int b = blockIdx.y * 4 + wv; b < a.blocks; b += gridDim.y * 4) {
    float v[16];
    if (row < a.rows) {
      const float4* s = reinterpret_cast<const float4*>(a.src + row * a.ld + 16 * b);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float4 x = s[q];
        v[4 * q] = x.x, v[4 * q + 1] = x.y, v[4 * q + 2] = x.z, v[4 * q + 3] = x.w;
      }
    } else {
#pragma unroll
      for (int q = 0; q < 16; ++q) v[q] = 0.f;
    }
    const float lo[8] = {v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]};
    const float hi[8] = {v[8], v[9], v[10], v[11], v[12], v[13], v[14], v[15]};
    const Frag3 f0 = split_pack8(lo), f1 = split_pack8(hi);
    const int64_t at = ((int64_t)b * a.rows_pad + row) * 32;
    *reinterpret_cast<bf16x8*>(a.p[0] + at) = f0.t1;
    *reinterpret_cast<bf16x8*>(a.p[0] + at + 16) = f1.t1;
    *reinterpret_cast<bf16x8*>(a.p[1] + at) = f0.t2;
    *reinterpret_cast<bf16x8*>(a.p[1] + at + 16) = f1.t2;
    *reinterpret_cast<bf16x8*>(a.p[2] + at) = f0.t3;
    *reinterpret_cast<bf16x8*>(a.p[2] + at + 16) = f1.t3;
  }
}

}  // namespace psf_wide
