// bwd_dw_chunk.h — dW for the chord pattern, any row width: channel chunks looped INSIDE the workgroup.
//
//   dW[b,p,k] = sum_c dZ[b,p,c] * V[b,(p+off_k) mod N,c]                       spmul/spmul_cuda.cu:102-111
//
// Why a second dW kernel (profiles/r02_bwd_summary.md): the whole-row kernel of bwd_window.h gives a row of C = 128
// channels 32 lanes, so a 256-thread workgroup covers 8 rows — 8000 tiny tiles at ListOps' shape, 7 of 12 links
// "far" (window of 16 rows), and 60 ds_bpermute per thread for the 32-lane butterflies: 35.6 us = 0.24 of the HBM
// roofline, 4.8x the algorithmic bytes through the L2->CU path.
//
//   lanes     TG <= 8 lanes share a row and cover one CHUNK of 4*TG channels (TG = 8: 32 channels = one 128-byte
//             line per row); the workgroup walks the row's C / (4*TG) chunks one after the other and every thread
//             keeps its R x L partial dots in registers across chunks. The tile is TR = (NT/TG)*R rows whatever C
//             is: 128 rows at TG = 8 with 1024 threads (9 of 12 links near at ListOps' shape instead of 5). Long
//             tiles come from MORE THREADS, not more rows per thread: a row costs ~5L+4 registers (L sums, 4 per
//             far row, 4 per window read in flight — hipcc issues them all up front), 264 VGPRs at R = 4.
//   window    per chunk, rows [p0, p0+2TR) x chunk staged by LDS-DMA (16 B per lane); far links and the tile's own
//             dZ rows to registers; one barrier to read, one before the next chunk overwrites the window.
//   reduce    once per tile, not per chunk: the TG lanes of a row are combined with DPP-fused adds
//             (row_half_mirror, quad_perm) — one VALU instruction per stage and value, no LDS crossbar.
//   store     the [TR, L] tile is assembled in LDS and written flat in 16-byte chunks, as in bwd_window.h.
//
// Summation order differs from the oracle's (channels are split over lanes and chunks): dW is held to 1e-5, not to
// bit equality, like the whole-row kernel.
#pragma once

#include "bwd_window.h"

namespace psf {

template <int L, int TGS, int R, int NT>
struct DwChunkCfg {
  using F = FwdWinCfg<float, L, TGS, R, NT>;
  static constexpr int TG = F::TG, RS = F::RS, TR = F::TR, KN = F::KN, NF = F::NF;
  static constexpr int win_vecs = F::win_vecs, win_bytes = F::win_bytes;
  static constexpr int w_passes = F::w_passes;
  static constexpr int lds_bytes = win_bytes + w_passes * NT * 16;
};

template <int L, int TGS, int R, int NT, bool EDGE>
__global__ void __launch_bounds__(NT)
chord_dw_chunk_k(const float* __restrict__ dZ, const float* __restrict__ V, float* __restrict__ dW, const Geom gm,
                 const Offsets offs, const int64_t w_total) {
  using T = float;
  using Cfg = DwChunkCfg<L, TGS, R, NT>;
  constexpr int VEC = 4, TG = Cfg::TG, RS = Cfg::RS, TR = Cfg::TR, KN = Cfg::KN, NF = Cfg::NF;
  using V4 = Vec<T, VEC>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  V4* __restrict__ sWin = reinterpret_cast<V4*>(smem);
  V4* __restrict__ sOutV = reinterpret_cast<V4*>(smem + Cfg::win_bytes);
  T* __restrict__ sOutF = reinterpret_cast<T*>(smem + Cfg::win_bytes);

  int b, tile, chunk;
  decode_block(gm, b, tile, chunk);  // launched with chunks_c == 1: the chunk loop is in here
  (void)chunk;
  const int tid = threadIdx.x, lane = tid & 63, wave64 = tid & ~63;
  const int g = tid & (TG - 1), rs = tid >> TGS;
  const int p0 = tile * TR, N = gm.N, C = gm.C;
  const int n_chunks = gm.CG >> TGS;  // host guarantees CG % TG == 0
  const T* __restrict__ Vb = V + (int64_t)b * gm.v_bstride;
  const T* __restrict__ Zb = dZ + (int64_t)b * N * C;

  float acc[R][L];
#pragma unroll
  for (int j = 0; j < R; ++j)
#pragma unroll
    for (int k = 0; k < L; ++k) acc[j][k] = 0.f;

  for (int cc = 0; cc < n_chunks; ++cc) {
    // Everything below that depends only on the thread (row indices, source rows, 64-bit addresses) is invariant
    // across chunks; hoisted out of this loop it costs ~100 registers at R = 4 (measured: 349 VGPRs + scratch).
    // Launder the thread id so that it is recomputed per chunk instead (a few dozen VALU per chunk).
    int tid_c = tid;
    asm volatile("" : "+v"(tid_c));
    const int g_c = tid_c & (TG - 1), rs_c = tid_c >> TGS;
    // chunk base pointers are wave-uniform; per-lane element offsets fit 32 bits (N*C < 2^31 checked by the host)
    const T* __restrict__ Vc = Vb + (int64_t)(cc << TGS) * VEC;
    const T* __restrict__ Zc = Zb + (int64_t)(cc << TGS) * VEC;
    // (1) V window [p0, p0+2TR) mod N, this chunk's channels
#pragma unroll
    for (int n = 0; n < Cfg::win_vecs / NT; ++n) {
      const int i = n * NT + tid_c;
      const int wr = i >> TGS, gg = i & (TG - 1);
      int src = p0 + wr;
      if (src >= N) src -= N;
      stage16<T, VEC, true>(Vc + (uint32_t)(src * C + gg * VEC), sWin + n * NT + wave64, lane);
    }
    // (2) far V rows and the tile's dZ rows -> registers
    V4 far[R][NF > 0 ? NF : 1];
    V4 dz[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const int pr = p0 + j * RS + rs_c;
      const int p = EDGE ? imin(pr, N - 1) : pr;
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        int src = p + offs.v[KN + f];
        if (src >= N) src -= N;
        far[j][f] = ld<T, VEC>(Vc + (uint32_t)(src * C + g_c * VEC));
      }
      dz[j] = ld<T, VEC>(Zc + (uint32_t)(p * C + g_c * VEC));
    }
    __syncthreads();
    // (3) partial dots of this chunk. Near links are the chord pattern itself (0, 1, 2, 4, ...: checked by the host),
    //     so every window read is one base address per row + a compile-time offset.
    const V4* __restrict__ sRow0 = sWin + ((rs_c << TGS) + g_c);
#pragma unroll
    for (int j = 0; j < R; ++j) {
#pragma unroll
      for (int k = 0; k < L; ++k) {
        V4 x;
        if (k < KN) x = sRow0[(j * RS + chord_off(k)) << TGS];
        else x = far[j][k - KN < NF ? k - KN : 0];
        T part = mul_rn(dz[j].e[0], x.e[0]);
#pragma unroll
        for (int i = 1; i < VEC; ++i) part = add_rn(part, mul_rn(dz[j].e[i], x.e[i]));
        acc[j][k] = add_rn(acc[j][k], part);
      }
    }
    if (cc + 1 < n_chunks) __syncthreads();  // the next chunk's DMA overwrites the window
  }

  // (4) combine the TG lanes of each row, assemble the [TR, L] tile in LDS:
  //     LDS float index (mis + pl*L + k) <-> global element e_lo + pl*L + k
  const int rows_here = EDGE ? imin(TR, N - p0) : TR;
  const int count = rows_here * L;
  const int64_t e_lo = ((int64_t)b * N + p0) * L;
  const int mis = (int)(((reinterpret_cast<uintptr_t>(dW) / sizeof(T)) + (uint64_t)e_lo) & (VEC - 1));
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const int pl = j * RS + rs;
#pragma unroll
    for (int k = 0; k < L; ++k) {
      const T s = row_group_sum<TG>(acc[j][k]);
      if (g == 0) sOutF[mis + pl * L + k] = s;
    }
  }
  __syncthreads();

  // (5) flat store: 16-byte chunks that lie wholly inside the tile, element-wise at its two ends
  const int64_t e_al = e_lo - mis;
  const int nvec = (mis + count + VEC - 1) / VEC;
  T* __restrict__ Oal = dW + e_al;
#pragma unroll
  for (int n = 0; n < Cfg::w_passes; ++n) {
    const int i = n * NT + tid;
    if (i < nvec) {
      const int f0 = i * VEC;
      if (!EDGE || (f0 >= mis && f0 + VEC <= mis + count)) {
        st<T, VEC>(Oal + (int64_t)i * VEC, sOutV[i]);
      } else {
#pragma unroll
        for (int u = 0; u < VEC; ++u)
          if (f0 + u >= mis && f0 + u < mis + count) Oal[(int64_t)i * VEC + u] = sOutF[f0 + u];
      }
    }
  }
  (void)w_total;
}

}  // namespace psf
