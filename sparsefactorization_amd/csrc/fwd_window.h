// fwd_window.h — LDS-window forward kernel for the chord pattern (C a multiple of the 16-byte vector width).
//
//   out[b,p,:] = sum_k W[b,p,k] * V[b,(p+off_k) mod N,:] (+ res[b,p,:])          spmul/spmul_cuda.cu:20-27
//
//   tile        TR = RS*R rows (RS = NT / TG row slots, R rows per thread, NT threads), TG channel groups
//   near links  the first KN links have off_k <= H = TR, so their sources lie in the window
//               [p0, p0+TR+H): staged ONCE in LDS and read back as conflict-free 16-byte LDS reads
//   far links   the remaining L-KN links stream from L2 straight to registers, one coalesced burst per link.
//               They are the expensive part (measured r01: 6 far links = 5.6 of 29 us at cfg2 although they
//               hit L2), which is why tiles are made long: every doubling of TR turns one far link into a
//               near one. NT = 512 / 1024 lengthens the tile without lowering the wave count per CU.
//   W tile      TR*L contiguous elements (rows are L*4 bytes: 60 B at L=15, not 16-B aligned). The 16-byte
//               chunks that cover the tile are copied flat into LDS, so the LDS image starts `mis` elements
//               before the tile (global and LDS addresses agree mod 16); each thread then reads its row's L
//               weights as LDS broadcasts.
//   schedule    every global access of the tile is issued before the single barrier: W tile and window by
//               LDS-DMA (global_load_lds_dwordx4: no VGPRs, no ds_write) or, with DMA = false, through
//               registers and aligned ds_write_b128; far rows and the residual row to registers.
//   edges       a tile is "full" when all its rows are < N, all its channel groups exist and every 16-byte W
//               chunk lies inside the W buffer. Full tiles (all of them at the benchmark shapes) run the
//               EDGE = false kernel, which has no per-lane predicate; the others run the EDGE = true kernel
//               (clamped rows / groups, element-wise copy of the at most two partial W chunks, predicated
//               store). The host picks per launch.
//
// Per output row this moves (2 + L-KN) V rows through the L2->CU path instead of L and keeps the accumulation
// order of the generic kernel, so both agree bit for bit. Requires N >= 2*TR (the window wraps at most once).
#pragma once

#include "psf_common.h"

// cache policy of the W-tile DMA (W is read exactly once per step): 0 = default, 2 = non-temporal.
// r01 lab (profiles/fwdlab.hip, A/B/A in one call): nt 27.2 us vs default 27.6 us per launch at cfg2.
#ifndef PSF_W_DMA_AUX
#define PSF_W_DMA_AUX 2
#endif

namespace psf {

template <typename T, int L, int TGS, int R, int NT>
struct FwdWinCfg {
  static constexpr int VEC = 16 / (int)sizeof(T);
  static constexpr int TG = 1 << TGS;
  static constexpr int RS = NT >> TGS;
  static constexpr int TR = RS * R;
  static constexpr int H = TR;
  static constexpr int WR = TR + H;
  static constexpr int KN = imin(L, ilog2_floor(H) + 2);  // offsets 0,1,2,...,2^(KN-2) <= H
  static constexpr int NF = L - KN;
  static constexpr int win_vecs = WR * TG;
  static constexpr int win_bytes = win_vecs * 16;
  static constexpr int w_vecs = (TR * L + VEC - 1) / VEC + 1;  // chunks covering a tile at any misalignment
  static constexpr int w_passes = (w_vecs + NT - 1) / NT;
  static constexpr int lds_bytes = win_bytes + w_passes * NT * 16;
  static_assert(RS >= 1 && win_vecs % NT == 0, "window slots are a whole number of passes");
};

// one 16-byte element per lane: global (per-lane address) -> LDS (wave-uniform base + lane*16)
// AUX = cache-policy bits of the DMA (0 = default, 2 = nt: streamed-once data).
template <typename T, int VEC, bool DMA, int AUX = 0>
__device__ __forceinline__ void stage16(const T* __restrict__ gsrc, Vec<T, VEC>* sdst_wave_base, int lane) {
  if constexpr (DMA) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)sdst_wave_base, 16, 0, AUX);
  } else {
    sdst_wave_base[lane] = ld<T, VEC>(gsrc);
  }
}

// the same from a global-address-space byte pointer (sbase(block) + lane offset, psf_common.h); always LDS-DMA
template <int AUX = 0>
__device__ __forceinline__ void stage16g(const PSF_GLOBAL char* gsrc, void* sdst_wave_base) {
  __builtin_amdgcn_global_load_lds((const PSF_GLOBAL void*)gsrc, (__attribute__((address_space(3))) void*)sdst_wave_base, 16, 0,
                                   AUX);
}

template <typename T, int L, int TGS, int R, int NT, bool DMA, bool RES, bool EDGE, bool ALIGNED>
__device__ __forceinline__ void fwd_win_body(const T* __restrict__ W, const T* __restrict__ V,
                                             const T* __restrict__ res, T* __restrict__ out, const Geom& gm,
                                             const Offsets& offs, const int64_t w_total, char* smem, int b, int p0,
                                             int chunk, int mis, int64_t e_al) {
  using Cfg = FwdWinCfg<T, L, TGS, R, NT>;
  constexpr int VEC = Cfg::VEC, TG = Cfg::TG, RS = Cfg::RS, TR = Cfg::TR;
  constexpr int KN = Cfg::KN, NF = Cfg::NF;
  using V4 = Vec<T, VEC>;

  V4* __restrict__ sWin = reinterpret_cast<V4*>(smem);
  V4* __restrict__ sWv = reinterpret_cast<V4*>(smem + Cfg::win_bytes);
  const T* __restrict__ sWf = reinterpret_cast<const T*>(smem + Cfg::win_bytes);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave64 = tid & ~63;  // first thread of this wave: wave-uniform
  const int g = tid & (TG - 1);
  const int rs = tid >> TGS;
  const int N = gm.N, C = gm.C;
  const int cg = chunk * TG + g;
  const bool cg_ok = !EDGE || cg < gm.CG;
  const int cgc = cg_ok ? cg : gm.CG - 1;  // clamped: loads are unconditional, the store is not

  const T* __restrict__ Vb = V + (int64_t)b * gm.v_bstride;

  V4 far[R][NF > 0 ? NF : 1];
  V4 rres[R];
  const int rows_here = EDGE ? imin(TR, N - p0) : TR;
  static_assert(!(EDGE && ALIGNED), "aligned launches have full tiles only");
  if constexpr (ALIGNED) {
    // Full tiles of a launch the host found aligned (Geom::aligned: N a multiple of TR, so is every far offset, every channel group
    // valid, chunk-clean W, N * C * sizeof(T) < 2^31): every row block this workgroup touches is TR-aligned and never wraps
    // inside, so every address is a wave-uniform base (scalar arithmetic) plus ONE per-lane byte offset — the request phase
    // costs a handful of vector instructions instead of a 64-bit multiply-add chain per load (r04: 151 -> 46 vector
    // instructions before the barrier at cfg2).
    const uint32_t rowB = (uint32_t)C * (uint32_t)sizeof(T);
    const char* __restrict__ Vbb = reinterpret_cast<const char*>(Vb);
    uint32_t voff[R];  // lane's row rs + j RS of a TR-aligned block, channel group cg
#pragma unroll
    for (int j = 0; j < R; ++j) voff[j] = (uint32_t)(j * RS + rs) * rowB + (uint32_t)cg * 16u;

    // (1) W tile: TR L / VEC flat 16-byte chunks exactly (an aligned launch's tiles start on 16-byte boundaries: TR is a
    //     multiple of 4 and W is chunk-clean, so mis = 0). No lane predicate anywhere in the request phase — a predicate
    //     splits the basic block and the loads behind it lose the scalar-base form: the last, partial pass clamps its chunk
    //     index instead and its surplus lanes re-read the last chunk into the pad behind the image.
    static_assert(TR % VEC == 0, "tiles start on 16-byte boundaries");
    const char* __restrict__ Wb16 = reinterpret_cast<const char*>(W + e_al);
    constexpr int kVecs = TR * L / VEC, kFull = kVecs / NT, kRem = kVecs - kFull * NT;
#pragma unroll
    for (int n = 0; n < kFull; ++n)
      stage16g<PSF_W_DMA_AUX>(sbase(Wb16 + (size_t)n * NT * 16) + (uint32_t)tid * 16u, sWv + n * NT + wave64);
    if constexpr (kRem > 0)
      stage16g<PSF_W_DMA_AUX>(sbase(Wb16 + (size_t)kFull * NT * 16) + (uint32_t)imin_rt(tid, kRem - 1) * 16u,
                              sWv + kFull * NT + wave64);
    // (2) V window: pass n holds rows n RS + rs of [p0, p0 + 2 TR) = block n / R (0: the tile, 1: the next one, mod N)
    int p1 = p0 + TR;
    if (p1 >= N) p1 -= N;
#pragma unroll
    for (int n = 0; n < Cfg::win_vecs / NT; ++n) {
      stage16g<0>(sbase(Vbb + (uint32_t)(n / R == 0 ? p0 : p1) * rowB) + voff[n % R], sWin + n * NT + wave64);
    }
    // (3) far rows and residual -> registers, issued AFTER every DMA request (the barrier below waits for the DMAs only)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      int s0 = p0 + offs.v[KN + f];
      if (s0 >= N) s0 -= N;
      const PSF_GLOBAL char* blk = sbase(Vbb + (uint32_t)s0 * rowB);
#pragma unroll
      for (int j = 0; j < R; ++j) far[j][f] = ldg<T, VEC>(blk + voff[j]);
    }
    if constexpr (RES) {
      const PSF_GLOBAL char* rb = sbase(reinterpret_cast<const char*>(res + ((int64_t)b * N + p0) * C));
#pragma unroll
      for (int j = 0; j < R; ++j) rres[j] = ldg<T, VEC>(rb + voff[j]);
    }
  } else {
    // ---- (1) W tile: the 16-byte chunks covering the tile's elements of the flat W buffer ----
    const int nvec = (mis + rows_here * L + VEC - 1) / VEC;
    const T* __restrict__ Wal = W + e_al;
#pragma unroll
    for (int n = 0; n < Cfg::w_passes; ++n) {
      const int i = n * NT + tid;
      if (i < nvec) {
        bool whole = true;
        if constexpr (EDGE) {
          const int64_t e0 = e_al + (int64_t)i * VEC;
          whole = e0 >= 0 && e0 + VEC <= w_total;
          if (!whole) {  // first / last 16 bytes of the whole buffer only
            T* se = reinterpret_cast<T*>(sWv + i);
#pragma unroll
            for (int u = 0; u < VEC; ++u)
              if (e0 + u >= 0 && e0 + u < w_total) se[u] = Wal[(int64_t)i * VEC + u];
          }
        }
        if (whole) stage16<T, VEC, DMA, PSF_W_DMA_AUX>(Wal + (int64_t)i * VEC, sWv + n * NT + wave64, lane);
      }
    }

    // ---- (2) V window [p0, p0+WR) mod N ----
#pragma unroll
    for (int n = 0; n < Cfg::win_vecs / NT; ++n) {
      const int i = n * NT + tid;
      const int wr = i >> TGS, gg = i & (TG - 1);
      int src = p0 + wr;
      if (src >= N) src -= N;
      const int cgi = chunk * TG + gg;
      if (!EDGE || cgi < gm.CG)
        stage16<T, VEC, DMA>(Vb + (int64_t)src * C + (int64_t)cgi * VEC, sWin + n * NT + wave64, lane);
    }

    // ---- (3) far rows and residual -> registers ----
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const int pr = p0 + j * RS + rs;
      const int p = EDGE ? imin(pr, N - 1) : pr;
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        int src = p + offs.v[KN + f];
        if (src >= N) src -= N;
        far[j][f] = ld<T, VEC>(Vb + (int64_t)src * C + (int64_t)cgc * VEC);
      }
      if constexpr (RES) rres[j] = ld<T, VEC>(res + ((int64_t)b * N + p) * C + (int64_t)cgc * VEC);
    }
  }

  if constexpr (ALIGNED) {
    // The tile and the window have to be in LDS before anyone reads them; the far rows and the residual only before their
    // own first use, which comes after the near links' arithmetic. Vector-memory operations retire in issue order, so
    // "at most the register loads still outstanding" means every DMA has landed; the compiler's own counted waits cover the
    // registers where they are read. (__syncthreads() would put a full vmcnt(0) here.)
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NF * R + (RES ? R : 0)) : "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  } else {
    __syncthreads();  // (hipcc drains vmcnt here: the DMA'd tiles and the register loads have all landed)
  }

  // ---- (4) accumulate, links ascending ----
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const int pl = j * RS + rs;
    const int p = p0 + pl;
    V4 acc;
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc.e[i] = T(0);
    const T* __restrict__ wrow = sWf + (ALIGNED ? 0 : mis) + pl * L;
#pragma unroll
    for (int k = 0; k < KN; ++k) {
      const V4 x = sWin[((pl + chord_off(k)) << TGS) + g];
      axpy_rn<T, VEC>(acc, wrow[k], x);
    }
#pragma unroll
    for (int f = 0; f < NF; ++f) axpy_rn<T, VEC>(acc, wrow[KN + f], far[j][f]);
    if constexpr (RES) {
#pragma unroll
      for (int i = 0; i < VEC; ++i) acc.e[i] = add_rn(acc.e[i], rres[j].e[i]);
    }
    if constexpr (ALIGNED) {
      PSF_GLOBAL char* ob = sbase(reinterpret_cast<char*>(out + ((int64_t)b * N + p0) * C));
      stg<T, VEC>(ob + lane_off((uint32_t)pl * ((uint32_t)C * (uint32_t)sizeof(T)) + (uint32_t)cg * 16u), acc);
    } else {
      if (!EDGE || (p < N && cg_ok)) st<T, VEC>(out + ((int64_t)b * N + p) * C + (int64_t)cg * VEC, acc);  // full tiles: no predicate
    }
  }
}

// MODE is chosen by the host per launch: 0 = full tiles, 1 = EDGE, 2 = full tiles of an aligned launch (Geom::aligned).
// EDGE is chosen by the host per launch (one code path per kernel: letting the two bodies share a kernel made
// hipcc tail-merge them and split the stores). EDGE = false requires of EVERY tile in the launch: all TR rows
// < N, all TG channel groups < CG, W base 16-byte aligned and B*N*L a multiple of the vector width (no partial
// 16-byte chunk anywhere). The dispatcher sends the full tiles to that kernel and the ragged last tile of each
// sequence, if any, to the EDGE = true kernel in a second small launch (gm.tile0 = first tile of the launch).
template <typename T, int L, int TGS, int R, int NT, bool DMA, bool RES, int MODE>
__global__ void __launch_bounds__(NT)
chord_fwd_win_k(const T* __restrict__ W, const T* __restrict__ V, const T* __restrict__ res,
                T* __restrict__ out, const Geom gm, const Offsets offs, const int64_t w_total) {
  using Cfg = FwdWinCfg<T, L, TGS, R, NT>;
  constexpr int VEC = Cfg::VEC;
  extern __shared__ __attribute__((aligned(16))) char smem[];

  int b, tile, chunk;
  decode_block(gm, b, tile, chunk);
  const int p0 = tile * Cfg::TR;
  const int64_t e_lo = ((int64_t)b * gm.N + p0) * L;  // first W element of the tile in the flat buffer
  const int mis = (int)(((reinterpret_cast<uintptr_t>(W) / sizeof(T)) + (uint64_t)e_lo) & (VEC - 1));
  const int64_t e_al = e_lo - mis;                    // element index of chunk 0 (-mis at the buffer start)
  fwd_win_body<T, L, TGS, R, NT, DMA, RES, MODE == 1, MODE == 2>(W, V, res, out, gm, offs, w_total, smem, b, p0, chunk, mis, e_al);
}

}  // namespace psf
