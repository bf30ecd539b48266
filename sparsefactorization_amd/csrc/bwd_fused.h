// bwd_fused.h — ONE kernel for a whole backward step of the chord operator on narrow rows (f32, C = 4, 8, 16 or 32):
//
//   dV[b,q,:] = sum_k W[b,(q-off_k) mod N,k] * dZ[b,(q-off_k) mod N,:]          spmul/spmul_cuda.cu:75-84
//   dW[b,p,k] = sum_c dZ[b,p,c] * V[b,(p+off_k) mod N,c]                         spmul/spmul_cuda.cu:102-111
//
// the two kernels the reference's backward_host launches over one dZ (spmul/spmul_cuda.cu:114-159). As separate launches
// (bwd_window.h) each stages the tile's dZ rows and pays a kernel boundary (~2.7 us at these sizes); here a 512-thread
// workgroup (one row per thread, TR = 512 / TG rows) stages, all by LDS-DMA and all before ONE barrier,
//   the dZ window [q0 - TR, q0 + TR)  — dV's backward window; its upper half IS dW's tile of dZ rows,
//   the V window  [q0, q0 + 2 TR)     — dW's forward window,
//   the two W tiles under the dZ window (flat 16-byte-chunk images),
// loads the far-link operands into registers (dZ rows and W column elements for dV, V rows for dW), then computes dV
// (links ascending, uncontracted:
// bit-identical to chord_dv_win_k and the oracle), the row dots of dW (same order as chord_dw_win_k) and writes the dW
// tile flat through LDS (the image reuses the first W tile's bytes).
// Full tiles only: N a multiple of TR, C / 4 = TG exactly, chunk-clean W / dW buffers; the host sends anything else to
// the two-kernel path.
#pragma once

#include "bwd_window.h"

namespace psf {

constexpr int kFusedThreads = 512;

// W image in LDS: the tile under the window's lower half, then the tile under its upper half, as ONE flat array of 2 TR rows
// (row wr of the window, link k at float wr L + k: every window read of W is one base register plus an immediate), then a
// pad for the surplus lanes of the last pass. Full tiles of chunk-clean buffers start on 16-byte boundaries (TR is a multiple
// of 4), so a tile is exactly TR L / 4 chunks.
template <int L, int TGS, int NT = kFusedThreads>
struct BwdFusedCfg {
  using B = BwdWinCfg<float, L, TGS, 1, NT>;
  static constexpr int tile_vecs = B::TR * L / 4;
  static constexpr int passes = (tile_vecs + NT - 1) / NT;
  static constexpr int full = tile_vecs / NT;            // passes with every lane inside the tile
  static constexpr int rem = tile_vecs - full * NT;      // lanes of the last pass inside it (0: no partial pass)
  static constexpr int w_img_bytes = (tile_vecs + passes * NT) * 16;
  static constexpr int lds_bytes = 2 * B::win_bytes + w_img_bytes;
  static_assert(B::TR % 4 == 0, "tiles start on 16-byte boundaries");
};

// 256 threads at C <= 8: five workgroups per CU (32 KB of LDS each at C = 8) need <= 96 registers; unbounded hipcc takes 100.
// Wider rows have more far links to hold (8 at C = 32): under that bound the C = 32 instance spilled 22 registers and took
// 99 us per step where the 512-thread one takes 54 (genome shape, profiles/r03ap_bwd_fused_c32.log), so the bound is for the
// narrow instances only.
//
// ABL (diagnostic builds only, -DPSF_BWD_ABLATE_LAB, tuning knob "bwd_ablate"; 0 in the product): leave parts out to see what
// the step's time is made of — 1 far dZ / V rows, 2 far W elements, 4 the dZ and V windows, 8 the W tiles, 16 dV's
// arithmetic, 32 dW's arithmetic, 64 the stores (results are then wrong by construction); variants that stay correct: 128 dW
// stored non-temporally, 512 dW's dots contracted to FMAs, 256 (round 6) the far dZ / V rows by LDS-DMA into LDS (2 NF NT 16
// more bytes of LDS: 48 KB at C = 8) instead of 2 NF 16-byte register loads per thread.
template <int L, int TGS, int NT, int ABL = 0>
__global__ void __launch_bounds__(NT, (NT == 256 && TGS <= 1) ? 5 : 2)
chord_bwd_fused_k(const float* __restrict__ dZ, const float* __restrict__ W, const float* __restrict__ V,
                  float* __restrict__ dW, float* __restrict__ dV, const Geom gm, const Offsets offs, const int64_t w_total) {
  using T = float;
  using Cfg = BwdWinCfg<T, L, TGS, 1, NT>;
  constexpr int VEC = Cfg::VEC, TG = Cfg::TG, TR = Cfg::TR, KN = Cfg::KN, NF = Cfg::NF;
  using V4 = Vec<T, VEC>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  V4* __restrict__ sZ = reinterpret_cast<V4*>(smem);
  V4* __restrict__ sV = reinterpret_cast<V4*>(smem + Cfg::win_bytes);
  using FC = BwdFusedCfg<L, TGS, NT>;
  V4* __restrict__ sWV = reinterpret_cast<V4*>(smem + 2 * Cfg::win_bytes);
  const T* __restrict__ sWF = reinterpret_cast<const T*>(sWV);
  T* __restrict__ sOutF = reinterpret_cast<T*>(sWV);  // the dW tile image: written after the last read of the W tiles
  V4* __restrict__ sFarZ = reinterpret_cast<V4*>(smem + FC::lds_bytes);  // ABL & 256 only: [NF][NT] far dZ rows, then far V rows
  V4* __restrict__ sFarV = sFarZ + (NF > 0 ? NF : 1) * NT;

  int b, tile, chunk;
  decode_block(gm, b, tile, chunk);  // chunks_c == 1
  const int tid = threadIdx.x, wave64 = tid & ~63;
  const int g = tid & (TG - 1), pl = tid >> TGS;  // one row per thread: row slot = local row
  const int q0 = tile * TR, N = gm.N, C = gm.C;
  const T* __restrict__ Zb = dZ + (int64_t)b * N * C;
  const T* __restrict__ Wb = W + (int64_t)b * N * L;
  const T* __restrict__ Vb = V + (int64_t)b * gm.v_bstride;

  // Every row block this workgroup touches is TR-aligned and never wraps inside (host-checked: N and every far offset are
  // multiples of TR, a batch element's rows span < 2^31 bytes), so every address is a wave-uniform base — scalar arithmetic —
  // plus one per-lane byte offset: the request phase issues its 26 memory instructions behind ~40 vector instructions instead
  // of 180 (a 64-bit multiply-add chain per load before, r04).
  constexpr uint32_t rowB = TG * 16u;          // C = 4 TG exactly (host-checked)
  const uint32_t voff = (uint32_t)tid * 16u;  // lane's row pl of a block, channel group g: pl rowB + 16 g
  const char* __restrict__ Zbb = reinterpret_cast<const char*>(Zb);
  const char* __restrict__ Vbb = reinterpret_cast<const char*>(Vb);
  int prev0 = q0 - TR;
  if (prev0 < 0) prev0 += N;
  int next0 = q0 + TR;
  if (next0 >= N) next0 -= N;
  // All requests sit in ONE basic block (the scalar-base form is selected per block: a lane predicate around any of them
  // splits the block and the loads after it fall back to 64-bit vector addresses): the W tiles' last, partial pass clamps its
  // chunk index instead of masking lanes — the surplus lanes re-read the tile's last chunk into LDS slots nobody reads.
  // (1) dZ window: slot wr <-> row (q0 - TR + wr) mod N;  V window: slot wr <-> row (q0 + wr) mod N  (one pass per block)
  static_assert(Cfg::win_vecs / NT == 2, "one row per thread: a window is two passes of TR rows");
  if constexpr (!(ABL & 4)) {
    stage16g<0>(sbase(Zbb + (uint32_t)prev0 * rowB) + voff, sZ + wave64);
    stage16g<0>(sbase(Vbb + (uint32_t)q0 * rowB) + voff, sV + wave64);
    stage16g<0>(sbase(Zbb + (uint32_t)q0 * rowB) + voff, sZ + NT + wave64);
    stage16g<0>(sbase(Vbb + (uint32_t)next0 * rowB) + voff, sV + NT + wave64);
  }
  // (2) far links -> registers
  V4 farZ[NF > 0 ? NF : 1], farV[NF > 0 ? NF : 1];
  T farW[NF > 0 ? NF : 1];
  int src0[NF > 0 ? NF : 1];
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    src0[f] = q0 - offs.v[KN + f];
    if (src0[f] < 0) src0[f] += N;
    int dst0 = q0 + offs.v[KN + f];
    if (dst0 >= N) dst0 -= N;
    if constexpr ((ABL & 256) != 0) {
      stage16g<0>(sbase(Zbb + (uint32_t)src0[f] * rowB) + voff, sFarZ + f * NT + wave64);
      stage16g<0>(sbase(Vbb + (uint32_t)dst0 * rowB) + voff, sFarV + f * NT + wave64);
    } else if constexpr (!(ABL & 1)) {
      farZ[f] = ldg<T, VEC>(sbase(Zbb + (uint32_t)src0[f] * rowB) + voff);
      farV[f] = ldg<T, VEC>(sbase(Vbb + (uint32_t)dst0 * rowB) + voff);
    } else {
#pragma unroll
      for (int i = 0; i < VEC; ++i) farZ[f].e[i] = T(src0[f]), farV[f].e[i] = T(dst0);
    }
  }
  // (3) the two W tiles under the backward window, flat 16-byte chunks. The upper tile's last pass clamps (its surplus lanes
  //     land in the pad); the lower tile's would land on the upper tile's first chunks, so it is lane-masked and comes LAST
  //     of all requests (3c).
  auto w_tile = [&](int row0) { return reinterpret_cast<const char*>(W + ((int64_t)b * N + row0) * L); };
  if constexpr (!(ABL & 8)) {
    const char* __restrict__ wp = w_tile(prev0);
    const char* __restrict__ wc = w_tile(q0);
#pragma unroll
    for (int n = 0; n < FC::full; ++n) {
      stage16g<0>(sbase(wp + (size_t)n * NT * 16) + voff, sWV + n * NT + wave64);
      stage16g<0>(sbase(wc + (size_t)n * NT * 16) + voff, sWV + FC::tile_vecs + n * NT + wave64);
    }
    if constexpr (FC::rem > 0) {
      const uint32_t i = (uint32_t)imin_rt(tid, FC::rem - 1);
      stage16g<0>(sbase(wc + (size_t)FC::full * NT * 16) + i * 16u, sWV + FC::tile_vecs + FC::full * NT + wave64);
    }
  }
  // (3b) far-link W elements: element (src0 + pl) of the link's column of row-major W (a contiguous side copy, lane-packed
  //      loads and the far rows through LDS were all measured and dropped: DESIGN.md 4.10)
  if constexpr (!(ABL & 2)) {
#pragma unroll
    for (int f = 0; f < NF; ++f)
      farW[f] = *reinterpret_cast<const PSF_GLOBAL T*>(sbase(reinterpret_cast<const char*>(Wb + (int64_t)src0[f] * L + (KN + f))) +
                                                       (uint32_t)pl * (uint32_t)(L * sizeof(T)));
  } else {
#pragma unroll
    for (int f = 0; f < NF; ++f) farW[f] = T(f);
  }
  // (3c) the lower W tile's partial pass
  if constexpr (!(ABL & 8) && FC::rem > 0) {
    if (tid < FC::rem)
      stage16g<0>(sbase(w_tile(prev0) + (size_t)FC::full * NT * 16) + voff, sWV + FC::full * NT + wave64);
  }
  __syncthreads();

  // (4) dV, links ascending
  {
    V4 acc;
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc.e[i] = T(0);
    if constexpr (!(ABL & 16)) {
#pragma unroll
      for (int k = 0; k < KN; ++k) {
        const int wr = TR + pl - chord_off(k);  // in [0, 2 TR)
        axpy_rn<T, VEC>(acc, sWF[wr * L + k], sZ[(wr << TGS) + g]);
      }
    }
#pragma unroll
    for (int f = 0; f < NF; ++f) axpy_rn<T, VEC>(acc, farW[f], (ABL & 256) ? sFarZ[f * NT + tid] : farZ[f]);
    if (!(ABL & 64) || acc.e[0] == T(12345.678)) {
      // dV is a plain store: the next (earlier) step reads it at once (non-temporal measured and dropped: DESIGN.md 4.10)
      stg<T, VEC>(sbase(reinterpret_cast<char*>(dV + ((int64_t)b * N + q0) * C)) + lane_off(voff), acc);
    }
  }
  // (5) dW row dots (the tile's dZ rows are the upper half of the dZ window)
  T dots[L];
  {
    const V4 dz = sZ[((TR + pl) << TGS) + g];
#pragma unroll
    for (int k = 0; k < L; ++k) {
      V4 x;
      if (k < KN) x = (ABL & 32) ? dz : sV[((pl + chord_off(k)) << TGS) + g];
      else x = (ABL & 256) ? sFarV[(k - KN < NF ? k - KN : 0) * NT + tid] : farV[k - KN < NF ? k - KN : 0];
      // the lane's four products as two packed multiplies, summed pairwise: (p0 + p2) + (p1 + p3) — four instructions. (Written
      // as a running sum, hipcc paired the sums of two LINKS into packed adds and paid four register moves per pair: eight
      // instructions per link. dW is held to 1e-5, not to the oracle's bits: its sum over a row's lanes is a tree already.)
      using F2 = float __attribute__((ext_vector_type(2)));
      T part;
      if constexpr ((ABL & 512) != 0) {
        part = T(0);
#pragma unroll
        for (int i = 0; i < VEC; ++i) part = __builtin_fmaf(dz.e[i], x.e[i], part);
      } else {
        const F2 pa = F2{dz.e[0], dz.e[1]} * F2{x.e[0], x.e[1]}, pb = F2{dz.e[2], dz.e[3]} * F2{x.e[2], x.e[3]};
        const F2 ps = pa + pb;
        part = add_rn(ps.x, ps.y);
      }
      dots[k] = row_group_sum<TG>(part);
    }
  }
  if constexpr ((ABL & 64) != 0) {
    if (dots[0] != T(12345.678)) return;
  }
  __syncthreads();  // every thread is done with the W tiles: their first image becomes the dW tile
  if (g == 0) {
#pragma unroll
    for (int k = 0; k < L; ++k) sOutF[pl * L + k] = dots[k];
  }
  __syncthreads();
  // (6) flat store of the dW tile: TR L / 4 whole chunks (full tiles of chunk-clean buffers: host-checked)
  PSF_GLOBAL char* ob = sbase(reinterpret_cast<char*>(dW + ((int64_t)b * N + q0) * L));
  const uint32_t vo = lane_off(voff);
  const V4* __restrict__ sOutV = reinterpret_cast<const V4*>(sOutF);
#pragma unroll
  for (int n = 0; n < FC::passes; ++n) {
    const int i = n * NT + tid;
    if (n < FC::full || i < FC::tile_vecs) {
      // The dW tile leaves non-temporally: nothing reads it before the MLP backward at the end of the chain, and parked in
      // L2 it pushes out the dZ / V rows the tiles re-read through their far links (Order step 2.104 -> 2.079 ms: DESIGN.md 4.3)
      using F4 = float __attribute__((ext_vector_type(4)));
      __builtin_nontemporal_store(*reinterpret_cast<const F4*>(&sOutV[i]), reinterpret_cast<PSF_GLOBAL F4*>(ob + ((uint32_t)(n * NT) * 16u + vo)));
    }
  }
}

// The same step for ANY sequence length of at least two tiles and any far offsets — LRA's CLS-token column makes N = 2^k + 1
// (LRA/listops_training.py:65-72: IMDb N = 4097), where no row block is tile-aligned: per-lane wrapped addresses, the last
// tile of a sequence partial (its rows >= N request clamped addresses and store nothing), W tiles and the dW tile at any
// misalignment (the ends of a dW tile that share a 16-byte chunk with a neighbour's rows go out element by element, chunks
// sticking out of the buffer are staged element by element). Arithmetic and order as above: dV bit-identical to the oracle.
// Before it, these shapes took the two-kernel path: 26.1 -> 23.5 us per step at IMDb's shape (N = 4097, C = 32, B = 32; 22.3 at
// N = 4096), 13.0 -> 10.2 at N = 1025, C = 32, B = 64; no change for 8-channel rows (profiles/r04ae_*, r04af_*).
template <int L, int TGS, int NT>
__global__ void __launch_bounds__(NT, (NT == 256 && TGS <= 1) ? 5 : 2)
chord_bwd_fused_edge_k(const float* __restrict__ dZ, const float* __restrict__ W, const float* __restrict__ V,
                       float* __restrict__ dW, float* __restrict__ dV, const Geom gm, const Offsets offs, const int64_t w_total) {
  using T = float;
  using Cfg = BwdWinCfg<T, L, TGS, 1, NT>;
  constexpr int VEC = Cfg::VEC, TG = Cfg::TG, TR = Cfg::TR, KN = Cfg::KN, NF = Cfg::NF;
  using V4 = Vec<T, VEC>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  V4* __restrict__ sZ = reinterpret_cast<V4*>(smem);
  V4* __restrict__ sV = reinterpret_cast<V4*>(smem + Cfg::win_bytes);
  V4* __restrict__ sWpV = reinterpret_cast<V4*>(smem + 2 * Cfg::win_bytes);
  V4* __restrict__ sWcV = reinterpret_cast<V4*>(smem + 2 * Cfg::win_bytes + Cfg::w_tile_bytes);
  const T* __restrict__ sWpF = reinterpret_cast<const T*>(sWpV);
  const T* __restrict__ sWcF = reinterpret_cast<const T*>(sWcV);
  T* __restrict__ sOutF = reinterpret_cast<T*>(sWpV);  // the dW tile image: written after the last read of the W tiles

  int b, tile, chunk;
  decode_block(gm, b, tile, chunk);  // chunks_c == 1
  const int tid = threadIdx.x, lane = tid & 63, wave64 = tid & ~63;
  const int g = tid & (TG - 1), pl = tid >> TGS;  // one row per thread: row slot = local row
  const int q0 = tile * TR, N = gm.N, C = gm.C, q = q0 + pl;
  const int rows_here = imin_rt(TR, N - q0);
  const bool row_ok = pl < rows_here;
  const int qc = row_ok ? q : N - 1;  // a row that exists, for the requests of the lanes past the end
  const T* __restrict__ Zb = dZ + (int64_t)b * N * C;
  const T* __restrict__ Wb = W + (int64_t)b * N * L;
  const T* __restrict__ Vb = V + (int64_t)b * gm.v_bstride;

  // (1) the two W tiles under the backward window: rows [q0 - TR, q0) mod N (contiguous: N >= 2 TR) and [q0, q0 + rows_here)
  int prev0 = q0 - TR;
  if (prev0 < 0) prev0 += N;
  int misP, misC;
  stage_flat_tile<T, VEC, NT, Cfg::w_passes, true>(W, w_total, ((int64_t)b * N + prev0) * L, TR * L, sWpV, misP);
  stage_flat_tile<T, VEC, NT, Cfg::w_passes, true>(W, w_total, ((int64_t)b * N + q0) * L, rows_here * L, sWcV, misC);
  // (2) dZ window: slot wr <-> row (q0 - TR + wr) mod N;  V window: slot wr <-> row (q0 + wr) mod N
#pragma unroll
  for (int n = 0; n < Cfg::win_vecs / NT; ++n) {
    const int i = n * NT + tid;
    const int wr = i >> TGS, gg = i & (TG - 1);
    int sz = q0 - TR + wr;
    if (sz < 0) sz += N;
    if (sz >= N) sz -= N;
    int sv = q0 + wr;
    if (sv >= N) sv -= N;
    stage16<T, VEC, true>(Zb + (int64_t)sz * C + (int64_t)gg * VEC, sZ + n * NT + wave64, lane);
    stage16<T, VEC, true>(Vb + (int64_t)sv * C + (int64_t)gg * VEC, sV + n * NT + wave64, lane);
  }
  // (3) far links -> registers
  V4 farZ[NF > 0 ? NF : 1], farV[NF > 0 ? NF : 1];
  T farW[NF > 0 ? NF : 1];
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    int src = qc - offs.v[KN + f];
    if (src < 0) src += N;
    farZ[f] = ld<T, VEC>(Zb + (int64_t)src * C + (int64_t)g * VEC);
    farW[f] = Wb[(int64_t)src * L + (KN + f)];
    int dst = qc + offs.v[KN + f];
    if (dst >= N) dst -= N;
    farV[f] = ld<T, VEC>(Vb + (int64_t)dst * C + (int64_t)g * VEC);
  }
  __syncthreads();

  // (4) dV, links ascending
  {
    V4 acc;
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc.e[i] = T(0);
#pragma unroll
    for (int k = 0; k < KN; ++k) {
      const int wr = TR + pl - chord_off(k);  // in [0, 2 TR)
      const T w = wr < TR ? sWpF[misP + wr * L + k] : sWcF[misC + (wr - TR) * L + k];
      axpy_rn<T, VEC>(acc, w, sZ[(wr << TGS) + g]);
    }
#pragma unroll
    for (int f = 0; f < NF; ++f) axpy_rn<T, VEC>(acc, farW[f], farZ[f]);
    if (row_ok) st<T, VEC>(dV + ((int64_t)b * N + q) * C + (int64_t)g * VEC, acc);
  }
  // (5) dW row dots (the tile's dZ rows are the upper half of the dZ window)
  T dots[L];
  {
    using F2 = float __attribute__((ext_vector_type(2)));
    const V4 dz = sZ[((TR + pl) << TGS) + g];
#pragma unroll
    for (int k = 0; k < L; ++k) {
      V4 x;
      if (k < KN) x = sV[((pl + chord_off(k)) << TGS) + g];
      else x = farV[k - KN < NF ? k - KN : 0];
      const F2 pa = F2{dz.e[0], dz.e[1]} * F2{x.e[0], x.e[1]}, pb = F2{dz.e[2], dz.e[3]} * F2{x.e[2], x.e[3]};
      const F2 ps = pa + pb;
      dots[k] = row_group_sum<TG>(add_rn(ps.x, ps.y));
    }
  }
  __syncthreads();  // every thread is done with the W tiles: their first image becomes the dW tile
  const int64_t e_lo = ((int64_t)b * N + q0) * L;
  const int misO = (int)(((reinterpret_cast<uintptr_t>(dW) / sizeof(T)) + (uint64_t)e_lo) & (VEC - 1));
  if (g == 0 && row_ok) {
#pragma unroll
    for (int k = 0; k < L; ++k) sOutF[misO + pl * L + k] = dots[k];
  }
  __syncthreads();
  // (6) flat store of the dW tile's rows_here * L elements: whole 16-byte chunks inside, element by element at the two ends
  const int n_el = rows_here * L;
  const int nvec = (misO + n_el + VEC - 1) / VEC;
  T* __restrict__ Oal = dW + (e_lo - misO);
  const V4* __restrict__ sOutV = reinterpret_cast<const V4*>(sOutF);
#pragma unroll
  for (int n = 0; n < Cfg::w_passes; ++n) {
    const int i = n * NT + tid;
    if (i < nvec) {
      const int f0 = i * VEC;
      if (f0 >= misO && f0 + VEC <= misO + n_el) {
        using F4 = float __attribute__((ext_vector_type(4)));  // non-temporal, as in chord_bwd_fused_k
        __builtin_nontemporal_store(*reinterpret_cast<const F4*>(&sOutV[i]), reinterpret_cast<F4*>(Oal + (int64_t)i * VEC));
      } else {
#pragma unroll
        for (int u = 0; u < VEC; ++u)
          if (f0 + u >= misO && f0 + u < misO + n_el) Oal[(int64_t)i * VEC + u] = sOutF[f0 + u];
      }
    }
  }
}

// LDS of the EDGE instance: two windows and two padded tile images (any misalignment)
template <int L, int TGS, int NT = kFusedThreads>
struct BwdFusedEdgeCfg {
  using B = BwdWinCfg<float, L, TGS, 1, NT>;
  static constexpr int lds_bytes = 2 * B::win_bytes + 2 * B::w_tile_bytes;
};

}  // namespace psf
