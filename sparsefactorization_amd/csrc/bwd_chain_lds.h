// bwd_chain_lds.h — the WHOLE backward chain in one launch for short sequences of narrow rows (N <= 1024, C = 4 or 8: the
// synthetic tasks up to N = 1024, BASELINE configs[0] among them), the counterpart of fwd_chain_lds.h.
//
//     g_M = dOut;   for m = M-1 .. 0:
//         dW_m[p,k] = sum_d g_{m+1}[p,d] * X_m[(p+off_k) mod N, d]                      (spmul/spmul_cuda.cu:102-111)
//         g_m[q,:]  = sum_k W_m[(q-off_k) mod N, k] * g_{m+1}[(q-off_k) mod N, :]        (spmul/spmul_cuda.cu:75-84)
//     dV0 = g_0                                        without the residual
//     dV0 = ((g_M + g_{M-1}) + ... + g_1) + g_0        with it (X_{m+1} = W_m X_m + V0 sends every g to V0 as well;
//                                                      the order is the per-step path's: chord.py, psf_sum_tensors_f32)
//
// Per step the per-step path launches a kernel of 4-5 us for a few microseconds of work (N = 128: two, the generic dV
// and dW kernels): 25 % of the kernel time of a cfg1 training step, 13-15 % at N = 1024 (profiles/r06x_short_step_kernels.log).
// Here ONE workgroup owns a sequence and all its channels, so nothing crosses workgroups: the running gradient lives in LDS
// (every transposed link of the dV sum is an LDS read), X_m and W_m are staged per step (W rows at an odd stride: the
// column reads W_m[(q-off_k), k] are conflict-free), and a thread owns one row: its dW row (L dot products over the C
// channels) and its g row. No g_m is ever written to memory; the residual sum is kept in registers.
// Arithmetic: products and sums rounded separately, links / channels ascending from zero — bit-identical to the oracle for
// dV0 AND dW (the per-step dW kernels reduce across lanes in another order and match it to 1e-6).
#pragma once

#include "fwd_chain_lds.h"  // kChainMaxSteps, WRow
#include "psf_common.h"

namespace psf {

constexpr int kChainBwdRows = 1024;  // rows a workgroup can hold (one row per thread); also the channel-group stride in LDS

struct ChainBwdArgs {
  const float* W[kChainMaxSteps];  // W_m [B, N, L]
  const float* X[kChainMaxSteps];  // X_m [B, N, C]: X[0] = V0, X[m] = the forward's result of step m-1
  float* dW[kChainMaxSteps];       // dW_m [B, N, L]
  const float* dOut;               // [B, N, C]
  float* dV0;                      // [B, N, C]
  int32_t M, N, C;
};

template <int L>
constexpr int chain_bwd_wstride() {
  return L | 1;
}

template <int L, int G, bool RES>
__global__ void __launch_bounds__(1024)
chord_chain_bwd_lds_k(const ChainBwdArgs a, const Offsets offs) {
  using V4 = Vec<float, 4>;
  using F4 = float __attribute__((ext_vector_type(4)));
  constexpr int GS = kChainBwdRows, LP = chain_bwd_wstride<L>();
  extern __shared__ __attribute__((aligned(16))) F4 lds[];
  F4* const gs = lds;                                         // [G][GS] the gradient entering the step
  F4* const xs = lds + G * GS;                                // [G][GS] X_m
  float* const ws = reinterpret_cast<float*>(lds + 2 * G * GS);  // [N][LP] W_m
  const int N = a.N, C = a.C;
  const int tid = threadIdx.x, b = blockIdx.x;
  const bool pok = tid < N;
  const int p = pok ? tid : N - 1;
  const int64_t row = (int64_t)b * N + p;

  V4 g[G], racc[G];
#pragma unroll
  for (int gi = 0; gi < G; ++gi) {
    g[gi] = ld<float, 4>(a.dOut + row * C + gi * 4);
    racc[gi] = g[gi];
  }

  for (int m = a.M - 1; m >= 0; --m) {
    // stage: this row of the gradient, of X_m and of W_m
    const WRow<L> w = *reinterpret_cast<const WRow<L>*>(a.W[m] + row * L);
    if (pok) {
#pragma unroll
      for (int gi = 0; gi < G; ++gi) {
        const V4 x = ld<float, 4>(a.X[m] + row * C + gi * 4);
        xs[gi * GS + p] = F4{x.e[0], x.e[1], x.e[2], x.e[3]};
        gs[gi * GS + p] = F4{g[gi].e[0], g[gi].e[1], g[gi].e[2], g[gi].e[3]};
      }
#pragma unroll
      for (int k = 0; k < L; ++k) ws[p * LP + k] = w.e[k];
    }
    __syncthreads();

    // dW_m[p, :]: L dot products of the row's gradient with the linked rows of X_m, channels ascending
    WRow<L> dw;
#pragma unroll
    for (int k = 0; k < L; ++k) {
      int src = p + offs.v[k];
      if (src >= N) src -= N;
      float acc = 0.f;
#pragma unroll
      for (int gi = 0; gi < G; ++gi) {
        const F4 x = xs[gi * GS + src];
        acc = add_rn(acc, mul_rn(g[gi].e[0], x.x));
        acc = add_rn(acc, mul_rn(g[gi].e[1], x.y));
        acc = add_rn(acc, mul_rn(g[gi].e[2], x.z));
        acc = add_rn(acc, mul_rn(g[gi].e[3], x.w));
      }
      dw.e[k] = acc;
    }
    if (pok) *reinterpret_cast<WRow<L>*>(a.dW[m] + row * L) = dw;

    // g_m[p, :]: the transposed links, ascending
    V4 acc[G];
#pragma unroll
    for (int gi = 0; gi < G; ++gi)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[gi].e[i] = 0.f;
#pragma unroll
    for (int k = 0; k < L; ++k) {
      int src = p - offs.v[k];
      if (src < 0) src += N;
      const float wk = ws[src * LP + k];
#pragma unroll
      for (int gi = 0; gi < G; ++gi) {
        const F4 z = gs[gi * GS + src];
        axpy_rn<float, 4>(acc[gi], wk, V4{{z.x, z.y, z.z, z.w}});
      }
    }
    __syncthreads();  // everyone has read this step's gs / xs / ws
#pragma unroll
    for (int gi = 0; gi < G; ++gi) {
      g[gi] = acc[gi];
      if constexpr (RES) {
#pragma unroll
        for (int i = 0; i < 4; ++i) racc[gi].e[i] = add_rn(racc[gi].e[i], g[gi].e[i]);
      }
    }
  }
  if (pok) {
#pragma unroll
    for (int gi = 0; gi < G; ++gi) st<float, 4>(a.dV0 + row * C + gi * 4, RES ? racc[gi] : g[gi]);
  }
}

// Host side (bwd_chain_lds_inst.hip)
bool chain_bwd_lds_fits(int64_t N, int64_t C, int32_t L, int32_t M);
hipError_t launch_chain_bwd_lds(int L, int G, bool res, const ChainBwdArgs& a, const Offsets& offs, int B, hipStream_t stream);

}  // namespace psf
