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
//               [p0, p0+TR+H): staged ONCE in LDS (coalesced, 2 vectors per thread per row handled)
//               and read back as conflict-free 16-byte LDS reads
//   far links   the remaining L-KN links stream from L2/HBM straight to registers, one coalesced
//               burst per link, issued before anything else so they are in flight during the staging
//   W tile      TR*L contiguous elements (rows are L*4 bytes: 60 B at L=15, not 16-B aligned) copied flat
//               into LDS with 16-byte accesses; the LDS image is shifted so global and LDS addresses agree
//               mod 16; each thread then reads its row's L weights as LDS broadcasts
//
// Per output row this moves (2 + L-KN) V rows through the L2->CU path instead of L (e.g. 8 instead of 15
// at L=15, C=8, R=1) and keeps the accumulation order of variant 1, so both variants agree bit for bit.
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
  static constexpr int win_bytes = WR * TG * 16;
  static constexpr int w_elems = TR * L + VEC;  // + VEC: room for the alignment shift
  static constexpr int lds_bytes = win_bytes + ((w_elems * (int)sizeof(T) + 15) & ~15);
};

template <typename T, int L, int TGS, int R>
__global__ void __launch_bounds__(kBlock)
chord_fwd_win_k(const T* __restrict__ W, const T* __restrict__ V, const T* __restrict__ res,
                T* __restrict__ out, const Geom gm, const Offsets offs) {
  using Cfg = FwdWinCfg<T, L, TGS, R>;
  constexpr int VEC = Cfg::VEC, TG = Cfg::TG, RS = Cfg::RS, TR = Cfg::TR, WR = Cfg::WR;
  constexpr int KN = Cfg::KN, NF = Cfg::NF;
  using V4 = Vec<T, VEC>;

  extern __shared__ __attribute__((aligned(16))) char smem[];
  V4* __restrict__ sWin = reinterpret_cast<V4*>(smem);
  T* __restrict__ sWbase = reinterpret_cast<T*>(smem + Cfg::win_bytes);

  int b, tile, chunk;
  decode_block(gm, b, tile, chunk);
  const int tid = threadIdx.x;
  const int g = tid & (TG - 1);
  const int rs = tid >> TGS;
  const int p0 = tile * TR;
  const int cg = chunk * TG + g;
  const bool cg_ok = cg < gm.CG;
  const int N = gm.N, C = gm.C;

  const T* __restrict__ Vb = V + (int64_t)b * gm.v_bstride;

  // (1) far links -> registers (oldest loads in flight)
  V4 far[R][NF > 0 ? NF : 1];
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const int p = p0 + j * RS + rs;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      int src = p + offs.v[KN + f];
      if (src >= N) src -= N;
      if (p < N && cg_ok) far[j][f] = ld<T, VEC>(Vb + (int64_t)src * C + (int64_t)cg * VEC);
    }
  }

  // (2) W tile -> LDS, flat 16-byte copy with matching alignment on both sides
  const int rows_here = imin(TR, N - p0);
  const int wcount = rows_here * L;
  const T* __restrict__ Wt = W + ((int64_t)b * N + p0) * L;
  const int mis = (int)((reinterpret_cast<uintptr_t>(Wt) & 15) / sizeof(T));
  const int nhead = imin(mis ? VEC - mis : 0, wcount);
  T* __restrict__ sW = sWbase + ((VEC - nhead) & (VEC - 1));  // sW + nhead is 16-byte aligned
  const int nbody = (wcount - nhead) / VEC;
  const int ntail = wcount - nhead - nbody * VEC;
  for (int i = tid; i < nbody; i += kBlock)
    st<T, VEC>(sW + nhead + i * VEC, ld<T, VEC>(Wt + nhead + i * VEC));
  if (tid < nhead) sW[tid] = Wt[tid];
  if (tid < ntail) sW[nhead + nbody * VEC + tid] = Wt[nhead + nbody * VEC + tid];

  // (3) V window [p0, p0+WR) mod N -> LDS
  static_assert((WR * TG) % kBlock == 0, "window slots are a whole number of passes");
#pragma unroll
  for (int n = 0; n < (WR * TG) / kBlock; ++n) {
    const int i = tid + n * kBlock;
    const int wr = i >> TGS, gg = i & (TG - 1);
    int src = p0 + wr;
    if (src >= N) src -= N;
    const int cgi = chunk * TG + gg;
    if (cgi < gm.CG) sWin[i] = ld<T, VEC>(Vb + (int64_t)src * C + (int64_t)cgi * VEC);
  }
  __syncthreads();

  // (4) accumulate, links ascending
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const int pl = j * RS + rs;
    const int p = p0 + pl;
    if (p < N && cg_ok) {
      V4 acc;
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc.e[i] = T(0);
      const T* __restrict__ wrow = sW + pl * L;
#pragma unroll
      for (int k = 0; k < KN; ++k) {
        const V4 x = sWin[((pl + offs.v[k]) << TGS) + g];
        axpy_rn<T, VEC>(acc, wrow[k], x);
      }
#pragma unroll
      for (int f = 0; f < NF; ++f) axpy_rn<T, VEC>(acc, wrow[KN + f], far[j][f]);

      const int64_t o = ((int64_t)b * N + p) * C + (int64_t)cg * VEC;
      if (res != nullptr) {
        const V4 rv = ld<T, VEC>(res + o);
#pragma unroll
        for (int i = 0; i < VEC; ++i) acc.e[i] = add_rn(acc.e[i], rv.e[i]);
      }
      st<T, VEC>(out + o, acc);
    }
  }
}

}  // namespace psf
