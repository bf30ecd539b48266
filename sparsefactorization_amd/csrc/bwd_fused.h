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
// loads the far-link operands into registers (dZ rows and W column elements for dV — the latter from the link-major side
// copy when the producer left one, see psf_chord.h; V rows for dW), then computes dV (links ascending, uncontracted:
// bit-identical to chord_dv_win_k and the oracle), the row dots of dW (same order as chord_dw_win_k) and writes the dW
// tile flat through LDS (the image reuses the first W tile's bytes).
// Full tiles only: N a multiple of TR, C / 4 = TG exactly, chunk-clean W / dW buffers; the host sends anything else to
// the two-kernel path.
#pragma once

#include "bwd_window.h"

namespace psf {

constexpr int kFusedThreads = 512;

template <int L, int TGS, int NT = kFusedThreads>
struct BwdFusedCfg {
  using B = BwdWinCfg<float, L, TGS, 1, NT>;
  static constexpr int lds_bytes = 2 * B::win_bytes + 2 * B::w_tile_bytes;
};

// 256 threads at C <= 8: five workgroups per CU (32 KB of LDS each at C = 8) need <= 96 registers; unbounded hipcc takes 100.
// Wider rows have more far links to hold (8 at C = 32): under that bound the C = 32 instance spilled 22 registers and took
// 99 us per step where the 512-thread one takes 54 (genome shape, profiles/r03ap_bwd_fused_c32.log), so the bound is for the
// narrow instances only.
template <int L, int TGS, int NT>
__global__ void __launch_bounds__(NT, (NT == 256 && TGS <= 1) ? 5 : 2)
chord_bwd_fused_k(const float* __restrict__ dZ, const float* __restrict__ W, const float* __restrict__ V,
                  float* __restrict__ dW, float* __restrict__ dV, const Geom gm, const Offsets offs, const int64_t w_total,
                  const float* __restrict__ wfar, const int far_k0) {
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
  const T* __restrict__ Zb = dZ + (int64_t)b * N * C;
  const T* __restrict__ Wb = W + (int64_t)b * N * L;
  const T* __restrict__ Vb = V + (int64_t)b * gm.v_bstride;

  // (1) the two W tiles under the backward window
  int prev0 = q0 - TR;
  if (prev0 < 0) prev0 += N;
  int misP, misC;
  stage_flat_tile<T, VEC, NT, Cfg::w_passes, false>(W, w_total, ((int64_t)b * N + prev0) * L, TR * L, sWpV, misP);
  stage_flat_tile<T, VEC, NT, Cfg::w_passes, false>(W, w_total, ((int64_t)b * N + q0) * L, TR * L, sWcV, misC);
  // (2) dZ window: slot wr <-> row (q0 - TR + wr) mod N;  V window: slot wr <-> row (q0 + wr) mod N
#pragma unroll
  for (int n = 0; n < Cfg::win_vecs / NT; ++n) {
    const int i = n * NT + tid;
    const int wr = i >> TGS, gg = i & (TG - 1);
    int sz = q0 - TR + wr;
    if (sz < 0) sz += N;
    int sv = q0 + wr;
    if (sv >= N) sv -= N;
    stage16<T, VEC, true>(Zb + (int64_t)sz * C + (int64_t)gg * VEC, sZ + n * NT + wave64, lane);
    stage16<T, VEC, true>(Vb + (int64_t)sv * C + (int64_t)gg * VEC, sV + n * NT + wave64, lane);
  }
  // (3) far links -> registers
  V4 farZ[NF > 0 ? NF : 1], farV[NF > 0 ? NF : 1];
  T farW[NF > 0 ? NF : 1];
  const T* __restrict__ Wf = wfar ? wfar + ((int64_t)b * (L - far_k0) + (KN - far_k0)) * N : nullptr;  // wave-uniform
#pragma unroll
  for (int f = 0; f < NF; ++f) {
    int src = q - offs.v[KN + f];
    if (src < 0) src += N;
    farZ[f] = ld<T, VEC>(Zb + (int64_t)src * C + (int64_t)g * VEC);
    farW[f] = Wf ? Wf[(int64_t)f * N + src] : Wb[(int64_t)src * L + (KN + f)];
    int dst = q + offs.v[KN + f];
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
    st<T, VEC>(dV + ((int64_t)b * N + q) * C + (int64_t)g * VEC, acc);
  }
  // (5) dW row dots (the tile's dZ rows are the upper half of the dZ window)
  T dots[L];
  {
    const V4 dz = sZ[((TR + pl) << TGS) + g];
#pragma unroll
    for (int k = 0; k < L; ++k) {
      V4 x;
      if (k < KN) x = sV[((pl + chord_off(k)) << TGS) + g];
      else x = farV[k - KN < NF ? k - KN : 0];
      T part = T(0);
#pragma unroll
      for (int i = 0; i < VEC; ++i) part = add_rn(part, mul_rn(dz.e[i], x.e[i]));
      dots[k] = row_group_sum<TG>(part);
    }
  }
  __syncthreads();  // every thread is done with the W tiles: their first image becomes the dW tile
  const int64_t e_lo = ((int64_t)b * N + q0) * L;
  const int misO = (int)(((reinterpret_cast<uintptr_t>(dW) / sizeof(T)) + (uint64_t)e_lo) & (VEC - 1));
  if (g == 0) {
#pragma unroll
    for (int k = 0; k < L; ++k) sOutF[misO + pl * L + k] = dots[k];
  }
  __syncthreads();
  // (6) flat store of the dW tile in whole 16-byte chunks (full tiles of chunk-clean buffers only: host-checked)
  const int nvec = (misO + TR * L + VEC - 1) / VEC;
  T* __restrict__ Oal = dW + (e_lo - misO);
  const V4* __restrict__ sOutV = reinterpret_cast<const V4*>(sOutF);
#pragma unroll
  for (int n = 0; n < Cfg::w_passes; ++n) {
    const int i = n * NT + tid;
    if (i < nvec) {
      const int f0 = i * VEC;
      if (f0 >= misO && f0 + VEC <= misO + TR * L) {
        st<T, VEC>(Oal + (int64_t)i * VEC, sOutV[i]);
      } else {  // the two ends of a tile that does not start / end on a 16-byte boundary belong to its neighbours too
#pragma unroll
        for (int u = 0; u < VEC; ++u)
          if (f0 + u >= misO && f0 + u < misO + TR * L) Oal[(int64_t)i * VEC + u] = sOutF[f0 + u];
      }
    }
  }
}

}  // namespace psf
