// fwd_kernels.h — forward step  out[b,p,:] = sum_k W[b,p,k] * V[b,(p+off_k) mod N,:]  (+ res[b,p,:])
//
// Semantics follow spmul/spmul_cuda.cu:20-27 (the reference's own statement of the operator) and the
// gather -> multiply -> scatter_add that torch_sparse.spmm performs on the index list of
// SyntheticExperiments/psf.py:7-32; the sum runs k ascending with uncontracted mul/add.
//
// Why this is not a generic SpMM: the chord pattern makes every gather affine. For a tile of consecutive
// rows [p0, p0+TR) and a fixed link k the sources are the consecutive rows [p0+off_k, p0+off_k+TR) mod N:
// one coalesced burst, no index array is ever read. The kernels are HBM-bound (about 1.5 flop/byte), so the
// work here is about bytes: stream W once at full width, fetch each V row from L2/HBM as few times as
// possible, fuse the residual add into the store.
#pragma once

#include "psf_common.h"

namespace psf {

// ------------------------------------------------------------------------------------------------------
// Variant 1: generic direct-gather kernel. Any N, any L <= 64, any C (VEC = 1 when C is not a multiple of
// the 16-byte vector width), any offsets. Thread (r, g) owns VEC channels of one row and walks the links
// in order; the TG lanes of a row read the same W element (a wave-level broadcast).
// ------------------------------------------------------------------------------------------------------
template <typename T, int VEC>
__global__ void __launch_bounds__(kBlock)
chord_fwd_generic_k(const T* __restrict__ W, const T* __restrict__ V, const T* __restrict__ res,
                    T* __restrict__ out, const Geom gm, const Offsets offs) {
  int b, tile, chunk;
  decode_block(gm, b, tile, chunk);
  const int TG = 1 << gm.tg_shift;
  const int g = threadIdx.x & (TG - 1);
  const int r = threadIdx.x >> gm.tg_shift;
  const int p = tile * gm.TR + r;
  const int cg = chunk * TG + g;
  if (p >= gm.N || cg >= gm.CG) return;

  const T* __restrict__ Wrow = W + ((int64_t)b * gm.N + p) * gm.L;
  const T* __restrict__ Vb = V + (int64_t)b * gm.v_bstride + (int64_t)cg * VEC;

  Vec<T, VEC> acc;
#pragma unroll
  for (int i = 0; i < VEC; ++i) acc.e[i] = T(0);

#pragma unroll 4
  for (int k = 0; k < gm.L; ++k) {
    int src = p + offs.v[k];
    if (src >= gm.N) src -= gm.N;
    const T w = Wrow[k];
    const Vec<T, VEC> x = ld<T, VEC>(Vb + (int64_t)src * gm.C);
    axpy_rn<T, VEC>(acc, w, x);
  }

  const int64_t o = ((int64_t)b * gm.N + p) * gm.C + (int64_t)cg * VEC;
  if (res != nullptr) {
    const Vec<T, VEC> rv = ld<T, VEC>(res + o);
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc.e[i] = add_rn(acc.e[i], rv.e[i]);
  }
  st<T, VEC>(out + o, acc);
}

// ------------------------------------------------------------------------------------------------------
// Variant 2: LDS-window kernel for the chord pattern (C a multiple of the vector width).
//
//   tile        TR = RS*R rows (RS = 256 / TG row slots, R rows per thread), TG channel groups
//   near links  the first KN links have off_k <= H = TR, so their sources lie in the window
//               [p0, p0+TR+H): staged ONCE in LDS and read back as conflict-free 16-byte LDS reads
//   far links   the remaining L-KN links stream from L2/HBM straight to registers, one coalesced burst per link
//   W tile      TR*L contiguous elements (rows are L*4 bytes: 60 B at L=15, not 16-B aligned). The 16-byte
//               chunks that cover the tile are copied flat into LDS, so the LDS image starts `mis` elements
//               before the tile (global and LDS addresses agree mod 16); each thread then reads its row's L
//               weights as LDS broadcasts. Chunks that are not wholly inside the W buffer (possible only at
//               its first and last 16 bytes) are copied element-wise.
//   schedule    every global access of the tile is issued before the single barrier: W tile and window by
//               LDS-DMA (global_load_lds_dwordx4: no VGPRs, no ds_write) or, with DMA = false, through
//               registers and aligned ds_write_b128; far rows and the residual row to registers, branch-free
//               (rows past N are clamped, only the final store is predicated).
//
// Per output row this moves (2 + L-KN) V rows through the L2->CU path instead of L (7 instead of 15 at L=15,
// C=8, R=2) and keeps the accumulation order of variant 1, so both variants agree bit for bit.
// Requires N >= 2*TR (the window wraps at most once) — the dispatcher checks.
// ------------------------------------------------------------------------------------------------------
template <typename T, int L, int TGS, int R>
struct FwdWinCfg {
  static constexpr int VEC = 16 / (int)sizeof(T);
  static constexpr int TG = 1 << TGS;
  static constexpr int RS = kBlock >> TGS;
  static constexpr int TR = RS * R;
  static constexpr int H = TR;
  static constexpr int WR = TR + H;
  static constexpr int KN = imin(L, ilog2_floor(H) + 2);  // offsets 0,1,2,...,2^(KN-2) <= H
  static constexpr int NF = L - KN;
  static constexpr int win_vecs = WR * TG;
  static constexpr int win_bytes = win_vecs * 16;
  static constexpr int w_vecs = (TR * L + VEC - 1) / VEC + 1;  // chunks covering a tile at any misalignment
  static constexpr int w_passes = (w_vecs + kBlock - 1) / kBlock;
  static constexpr int lds_bytes = win_bytes + w_passes * kBlock * 16;
};

template <typename T, int VEC, bool DMA>
__device__ __forceinline__ void stage16(const T* __restrict__ gsrc, Vec<T, VEC>* sdst_wave_base, int lane) {
  // one 16-byte element per lane: global (per-lane address) -> LDS (wave-uniform base + lane*16)
  if constexpr (DMA) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)sdst_wave_base, 16, 0, 0);
  } else {
    sdst_wave_base[lane] = ld<T, VEC>(gsrc);
  }
}

template <typename T, int L, int TGS, int R, bool DMA>
__global__ void __launch_bounds__(kBlock)
chord_fwd_win_k(const T* __restrict__ W, const T* __restrict__ V, const T* __restrict__ res,
                T* __restrict__ out, const Geom gm, const Offsets offs, const int64_t w_total) {
  using Cfg = FwdWinCfg<T, L, TGS, R>;
  constexpr int VEC = Cfg::VEC, TG = Cfg::TG, RS = Cfg::RS, TR = Cfg::TR;
  constexpr int KN = Cfg::KN, NF = Cfg::NF;
  using V4 = Vec<T, VEC>;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  V4* __restrict__ sWin = reinterpret_cast<V4*>(smem);
  V4* __restrict__ sWv = reinterpret_cast<V4*>(smem + Cfg::win_bytes);
  const T* __restrict__ sWf = reinterpret_cast<const T*>(smem + Cfg::win_bytes);

  int b, tile, chunk;
  decode_block(gm, b, tile, chunk);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave64 = tid & ~63;  // first thread of this wave: wave-uniform
  const int g = tid & (TG - 1);
  const int rs = tid >> TGS;
  const int p0 = tile * TR;
  const int N = gm.N, C = gm.C;
  const int cg = chunk * TG + g;
  const bool cg_ok = cg < gm.CG;
  const int cgc = cg_ok ? cg : gm.CG - 1;  // clamped: loads are unconditional, the store is not

  const T* __restrict__ Vb = V + (int64_t)b * gm.v_bstride;

  // ---- (1) W tile: the 16-byte chunks covering elements [e_lo, e_lo + wcount) of the flat W buffer ----
  const int rows_here = imin(TR, N - p0);
  const int wcount = rows_here * L;
  const int64_t e_lo = ((int64_t)b * N + p0) * L;
  const int mis = (int)(((reinterpret_cast<uintptr_t>(W) / sizeof(T)) + (uint64_t)e_lo) & (VEC - 1));
  const int64_t e_al = e_lo - mis;            // element index of chunk 0 (may be -mis at the buffer start)
  const int nvec = (mis + wcount + VEC - 1) / VEC;
  const T* __restrict__ Wal = W + e_al;
#pragma unroll
  for (int n = 0; n < Cfg::w_passes; ++n) {
    const int i = n * kBlock + tid;
    const int64_t e0 = e_al + (int64_t)i * VEC;
    const bool whole = e0 >= 0 && e0 + VEC <= w_total;
    if (i < nvec) {
      if (whole) {
        stage16<T, VEC, DMA>(Wal + (int64_t)i * VEC, sWv + n * kBlock + wave64, lane);
      } else {  // first / last 16 bytes of the whole buffer only
        T* se = reinterpret_cast<T*>(sWv + i);
#pragma unroll
        for (int u = 0; u < VEC; ++u)
          if (e0 + u >= 0 && e0 + u < w_total) se[u] = Wal[(int64_t)i * VEC + u];
      }
    }
  }

  // ---- (2) V window [p0, p0+WR) mod N ----
  static_assert(Cfg::win_vecs % kBlock == 0, "window slots are a whole number of passes");
#pragma unroll
  for (int n = 0; n < Cfg::win_vecs / kBlock; ++n) {
    const int i = n * kBlock + tid;
    const int wr = i >> TGS, gg = i & (TG - 1);
    int src = p0 + wr;
    if (src >= N) src -= N;
    const int cgi = chunk * TG + gg;
    if (cgi < gm.CG)
      stage16<T, VEC, DMA>(Vb + (int64_t)src * C + (int64_t)cgi * VEC, sWin + n * kBlock + wave64, lane);
  }

  // ---- (3) far rows and residual -> registers ----
  V4 far[R][NF > 0 ? NF : 1];
  V4 rres[R];
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const int p = imin(p0 + j * RS + rs, N - 1);
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      int src = p + offs.v[KN + f];
      if (src >= N) src -= N;
      far[j][f] = ld<T, VEC>(Vb + (int64_t)src * C + (int64_t)cgc * VEC);
    }
    if (res != nullptr) rres[j] = ld<T, VEC>(res + ((int64_t)b * N + p) * C + (int64_t)cgc * VEC);
  }

  __syncthreads();  // (hipcc drains vmcnt here: the DMA'd tiles and the register loads have all landed)

  // ---- (4) accumulate, links ascending ----
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const int pl = j * RS + rs;
    const int p = p0 + pl;
    V4 acc;
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc.e[i] = T(0);
    const T* __restrict__ wrow = sWf + mis + pl * L;
#pragma unroll
    for (int k = 0; k < KN; ++k) {
      const V4 x = sWin[((pl + offs.v[k]) << TGS) + g];
      axpy_rn<T, VEC>(acc, wrow[k], x);
    }
#pragma unroll
    for (int f = 0; f < NF; ++f) axpy_rn<T, VEC>(acc, wrow[KN + f], far[j][f]);
    if (res != nullptr) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc.e[i] = add_rn(acc.e[i], rres[j].e[i]);
    }
    if (p < N && cg_ok) st<T, VEC>(out + ((int64_t)b * N + p) * C + (int64_t)cg * VEC, acc);
  }
}

}  // namespace psf
