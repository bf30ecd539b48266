// bwd_window.h — LDS-window backward kernels for the chord pattern (f32, C a multiple of 4).
//
//   dW[b,p,k] = sum_c dZ[b,p,c] * V[b,(p+off_k) mod N,c]                       spmul/spmul_cuda.cu:102-111
//   dV[b,q,:] = sum_k W[b,(q-off_k) mod N,k] * dZ[b,(q-off_k) mod N,:]          spmul/spmul_cuda.cu:75-84
//
// Both reuse the forward kernel's structure (fwd_window.h): a tile of TR = RS*R consecutive rows, near links
// served from an LDS window staged once by LDS-DMA, far links streamed from L2 into registers, everything
// issued before one barrier, host-selected EDGE instance for ragged / irregular launches.
//
//   dW  gathers V exactly like the forward pass (window [p0, p0+2TR) + far rows), keeps the tile's dZ rows in
//       registers, forms each row dot per lane over its 4 channels (c ascending, uncontracted) and combines
//       the TG lanes of a row with a wave-shuffle butterfly. The [TR, L] result tile is assembled in LDS and
//       written out flat in 16-byte chunks (rows are L*4 bytes, e.g. 60 B: per-row stores would be 4-byte
//       scatters).
//   dV  is the transpose gather: sources are rows q-off_k, so the window runs BACKWARD, [q0-TR, q0+TR).
//       dZ window rows and the two W tiles that cover them (the previous tile's rows and this tile's rows: two
//       flat 16-byte-chunk images, each with its own alignment shift) are staged by LDS-DMA; near links read
//       W[(q-off_k), k] and dZ[(q-off_k), :] from LDS. Far links load dZ rows (coalesced) and the single
//       column k of the far W tile (a 4-byte load with a 60-byte lane stride: the one place where a row-major
//       W costs L2->L1 bandwidth; a k-major W from the producer would remove it — DESIGN.md §4.3).
//       Links ascending with uncontracted mul/add: bit-identical to the oracle.
#pragma once

#include "fwd_window.h"

namespace psf {

// ---- cross-lane sums on DPP (one VALU instruction per stage; __shfl_xor compiles to ds_bpermute_b32, an LDS-crossbar
//      round trip per stage) ----
template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}

// Sum over the TG adjacent lanes that share a row. TG <= 16: a row group never straddles a DPP row of 16 lanes and the
// whole reduction is DPP; wider groups do their upper stages with wave shuffles first. Every lane of the group ends up
// with the total.
template <int TG>
__device__ __forceinline__ float row_group_sum(float v) {
#pragma unroll
  for (int s = TG >> 1; s >= 16; s >>= 1) v = add_rn(v, __shfl_xor(v, s, 64));
  if constexpr (TG >= 16) v = add_rn(v, dpp_f32<0x140>(v));  // row_mirror: lane i <-> 15 - i
  if constexpr (TG >= 8) v = add_rn(v, dpp_f32<0x141>(v));   // row_half_mirror: i <-> 7 - i
  if constexpr (TG >= 4) v = add_rn(v, dpp_f32<0x4E>(v));    // quad_perm [2,3,0,1]
  if constexpr (TG >= 2) v = add_rn(v, dpp_f32<0xB1>(v));    // quad_perm [1,0,3,2]
  return v;
}

template <typename T, int L, int TGS, int R, int NT>
struct BwdWinCfg {
  using F = FwdWinCfg<T, L, TGS, R, NT>;
  static constexpr int VEC = F::VEC, TG = F::TG, RS = F::RS, TR = F::TR, WR = F::WR, KN = F::KN, NF = F::NF;
  static constexpr int win_vecs = F::win_vecs, win_bytes = F::win_bytes;
  static constexpr int w_vecs = F::w_vecs, w_passes = F::w_passes;
  static constexpr int w_tile_bytes = w_passes * NT * 16;
  static constexpr int lds_dw = win_bytes + w_tile_bytes;
  static constexpr int lds_dv = win_bytes + 2 * w_tile_bytes;
};

// Flat copy of `count` elements starting at element e_lo of the global array `G` (total g_total elements) into
// an LDS image whose float index 0 corresponds to element e_lo - mis (so chunk boundaries agree). DMA for
// whole chunks; EDGE also copes with chunks that stick out of the buffer.
template <typename T, int VEC, int NT, int PASSES, bool EDGE>
__device__ __forceinline__ void stage_flat_tile(const T* __restrict__ G, int64_t g_total, int64_t e_lo, int count,
                                                Vec<T, VEC>* sImg, int& mis_out) {
  const int tid = threadIdx.x, lane = tid & 63, wave64 = tid & ~63;
  const int mis = (int)(((reinterpret_cast<uintptr_t>(G) / sizeof(T)) + (uint64_t)e_lo) & (VEC - 1));
  const int64_t e_al = e_lo - mis;
  const int nvec = (mis + count + VEC - 1) / VEC;
  const T* __restrict__ Gal = G + e_al;
#pragma unroll
  for (int n = 0; n < PASSES; ++n) {
    const int i = n * NT + tid;
    if (i < nvec) {
      bool whole = true;
      if constexpr (EDGE) {
        const int64_t e0 = e_al + (int64_t)i * VEC;
        whole = e0 >= 0 && e0 + VEC <= g_total;
        if (!whole) {
          T* se = reinterpret_cast<T*>(sImg + i);
#pragma unroll
          for (int u = 0; u < VEC; ++u)
            if (e0 + u >= 0 && e0 + u < g_total) se[u] = Gal[(int64_t)i * VEC + u];
        }
      }
      if (whole) stage16<T, VEC, true>(Gal + (int64_t)i * VEC, sImg + n * NT + wave64, lane);
    }
  }
  mis_out = mis;
}

// ------------------------------------------------------------------------------------------------------
// dW
// ------------------------------------------------------------------------------------------------------
template <typename T, int L, int TGS, int R, int NT, bool EDGE>
__global__ void __launch_bounds__(NT)
chord_dw_win_k(const T* __restrict__ dZ, const T* __restrict__ V, T* __restrict__ dW, const Geom gm,
               const Offsets offs, const int64_t w_total) {
  using Cfg = BwdWinCfg<T, L, TGS, R, NT>;
  constexpr int VEC = Cfg::VEC, TG = Cfg::TG, RS = Cfg::RS, TR = Cfg::TR, KN = Cfg::KN, NF = Cfg::NF;
  using V4 = Vec<T, VEC>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  V4* __restrict__ sWin = reinterpret_cast<V4*>(smem);
  V4* __restrict__ sOutV = reinterpret_cast<V4*>(smem + Cfg::win_bytes);
  T* __restrict__ sOutF = reinterpret_cast<T*>(smem + Cfg::win_bytes);

  int b, tile, chunk;
  decode_block(gm, b, tile, chunk);  // launched with chunks_c == 1: a workgroup sees whole rows (CG <= TG)
  const int tid = threadIdx.x, lane = tid & 63, wave64 = tid & ~63;
  const int g = tid & (TG - 1), rs = tid >> TGS;
  const int p0 = tile * TR, N = gm.N, C = gm.C;
  const bool cg_ok = !EDGE || g < gm.CG;
  const int cgc = cg_ok ? g : gm.CG - 1;
  const T* __restrict__ Vb = V + (int64_t)b * gm.v_bstride;

  // (1) V window [p0, p0+2TR) mod N
#pragma unroll
  for (int n = 0; n < Cfg::win_vecs / NT; ++n) {
    const int i = n * NT + tid;
    const int wr = i >> TGS, gg = i & (TG - 1);
    int src = p0 + wr;
    if (src >= N) src -= N;
    if (!EDGE || gg < gm.CG) stage16<T, VEC, true>(Vb + (int64_t)src * C + (int64_t)gg * VEC, sWin + n * NT + wave64, lane);
  }
  // (2) far V rows and this tile's dZ rows -> registers
  V4 far[R][NF > 0 ? NF : 1];
  V4 dz[R];
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
    dz[j] = ld<T, VEC>(dZ + ((int64_t)b * N + p) * C + (int64_t)cgc * VEC);
  }
  __syncthreads();

  // (3) row dots -> LDS tile. LDS float index (mis + pl*L + k) <-> global element e_lo + pl*L + k
  const int rows_here = EDGE ? imin(TR, N - p0) : TR;
  const int count = rows_here * L;
  const int64_t e_lo = ((int64_t)b * N + p0) * L;
  const int mis = (int)(((reinterpret_cast<uintptr_t>(dW) / sizeof(T)) + (uint64_t)e_lo) & (VEC - 1));
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const int pl = j * RS + rs;
#pragma unroll
    for (int k = 0; k < L; ++k) {
      V4 x;
      if (k < KN) x = sWin[((pl + chord_off(k)) << TGS) + g];
      else x = far[j][k - KN < NF ? k - KN : 0];
      T part = T(0);
#pragma unroll
      for (int i = 0; i < VEC; ++i) part = add_rn(part, mul_rn(dz[j].e[i], x.e[i]));
      // Lanes past the last channel group add nothing — by a select, not by a zero factor: their window slots were never
      // staged, and whatever an earlier kernel left in that LDS (a NaN or Inf pattern) times zero is NaN. (Until round 5 this
      // zeroed dz instead: dW rows came out NaN once in a while for C / 4 not a power of two, depending on what had run before.)
      if constexpr (EDGE) part = cg_ok ? part : T(0);
      if constexpr (sizeof(T) == 4) {
        part = row_group_sum<TG>(part);
      } else {
#pragma unroll
        for (int s = TG >> 1; s > 0; s >>= 1) part = add_rn(part, __shfl_xor(part, s, 64));
      }
      if (g == 0) sOutF[mis + pl * L + k] = part;
    }
  }
  __syncthreads();

  // (4) flat store of the tile: 16-byte chunks that lie wholly inside the tile, element-wise at its two ends
  const int64_t e_al = e_lo - mis;
  const int nvec = (mis + count + VEC - 1) / VEC;
  T* __restrict__ Oal = dW + e_al;
#pragma unroll
  for (int n = 0; n < Cfg::w_passes; ++n) {
    const int i = n * NT + tid;
    if (i < nvec) {
      const int f0 = i * VEC;  // LDS float index of the chunk
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

// ------------------------------------------------------------------------------------------------------
// dV
// ------------------------------------------------------------------------------------------------------
template <typename T, int L, int TGS, int R, int NT, bool EDGE>
__global__ void __launch_bounds__(NT)
chord_dv_win_k(const T* __restrict__ dZ, const T* __restrict__ W, T* __restrict__ dV, const Geom gm,
               const Offsets offs, const int64_t w_total) {
  using Cfg = BwdWinCfg<T, L, TGS, R, NT>;
  constexpr int VEC = Cfg::VEC, TG = Cfg::TG, RS = Cfg::RS, TR = Cfg::TR, KN = Cfg::KN, NF = Cfg::NF;
  using V4 = Vec<T, VEC>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  V4* __restrict__ sWin = reinterpret_cast<V4*>(smem);
  V4* __restrict__ sWpV = reinterpret_cast<V4*>(smem + Cfg::win_bytes);
  V4* __restrict__ sWcV = reinterpret_cast<V4*>(smem + Cfg::win_bytes + Cfg::w_tile_bytes);
  const T* __restrict__ sWpF = reinterpret_cast<const T*>(smem + Cfg::win_bytes);
  const T* __restrict__ sWcF = reinterpret_cast<const T*>(smem + Cfg::win_bytes + Cfg::w_tile_bytes);

  int b, tile, chunk;
  decode_block(gm, b, tile, chunk);
  const int tid = threadIdx.x, lane = tid & 63, wave64 = tid & ~63;
  const int g = tid & (TG - 1), rs = tid >> TGS;
  const int q0 = tile * TR, N = gm.N, C = gm.C;
  const int cg = chunk * TG + g;
  const bool cg_ok = !EDGE || cg < gm.CG;
  const int cgc = cg_ok ? cg : gm.CG - 1;
  const T* __restrict__ Zb = dZ + (int64_t)b * N * C;
  const T* __restrict__ Wb = W + (int64_t)b * N * L;

  // (1) the two W tiles under the backward window: previous TR rows (wrapping for the first tile) and own rows
  int prev0 = q0 - TR;
  if (prev0 < 0) prev0 += N;
  const int rows_here = EDGE ? imin(TR, N - q0) : TR;
  int misP, misC;
  stage_flat_tile<T, VEC, NT, Cfg::w_passes, EDGE>(W, w_total, ((int64_t)b * N + prev0) * L, TR * L, sWpV, misP);
  stage_flat_tile<T, VEC, NT, Cfg::w_passes, EDGE>(W, w_total, ((int64_t)b * N + q0) * L, rows_here * L, sWcV, misC);

  // (2) dZ window: slot wr <-> row (q0 - TR + wr) mod N
#pragma unroll
  for (int n = 0; n < Cfg::win_vecs / NT; ++n) {
    const int i = n * NT + tid;
    const int wr = i >> TGS, gg = i & (TG - 1);
    int src = q0 - TR + wr;
    if (src < 0) src += N;
    if (src >= N) src -= N;
    const int cgi = chunk * TG + gg;
    if (!EDGE || cgi < gm.CG) stage16<T, VEC, true>(Zb + (int64_t)src * C + (int64_t)cgi * VEC, sWin + n * NT + wave64, lane);
  }

  // (3) far links: dZ rows (coalesced) and one W column element per row, strided out of W's rows
  V4 farZ[R][NF > 0 ? NF : 1];
  T farW[R][NF > 0 ? NF : 1];
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const int qr = q0 + j * RS + rs;
    const int q = EDGE ? imin(qr, N - 1) : qr;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      int src = q - offs.v[KN + f];
      if (src < 0) src += N;
      farZ[j][f] = ld<T, VEC>(Zb + (int64_t)src * C + (int64_t)cgc * VEC);
      farW[j][f] = Wb[(int64_t)src * L + (KN + f)];
    }
  }
  __syncthreads();

  // (4) accumulate, links ascending
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const int pl = j * RS + rs;
    const int q = q0 + pl;
    V4 acc;
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc.e[i] = T(0);
#pragma unroll
    for (int k = 0; k < KN; ++k) {
      const int wr = TR + pl - chord_off(k);  // in [0, 2TR): near offsets are <= TR
      const T w = wr < TR ? sWpF[misP + wr * L + k] : sWcF[misC + (wr - TR) * L + k];
      axpy_rn<T, VEC>(acc, w, sWin[(wr << TGS) + g]);
    }
#pragma unroll
    for (int f = 0; f < NF; ++f) axpy_rn<T, VEC>(acc, farW[j][f], farZ[j][f]);
    // (Non-temporal dV / dW stores for rows of >= 64 channels: 4-7 % per step with rotating operands, -1 % in the ListOps
    // training step where the next kernel reads this dV at once; the dW tile alone non-temporal, as in the fused step:
    // +0.1 % there — profiles/r05x_bwd_wide_nt_ab.log, r05x_lra_step_nt_ab.log, r05x_step_wide_dw_nt_ab.log. Not taken.)
    if (!EDGE || (q < N && cg_ok)) st<T, VEC>(dV + ((int64_t)b * N + q) * C + (int64_t)cg * VEC, acc);
  }
}

}  // namespace psf
