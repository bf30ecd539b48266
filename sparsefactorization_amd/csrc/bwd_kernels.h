// bwd_kernels.h — gradients of the forward step (spmul/spmul_cuda.cu:61-112).
//
//   dV[b,q,:] = sum_k W[b,(q-off_k) mod N,k] * dZ[b,(q-off_k) mod N,:]        (transpose gather, .cu:79-80)
//   dW[b,p,k] = sum_c dZ[b,p,c] * V[b,(p+off_k) mod N,c]                      (row dots,        .cu:105-108)
//
// Like the forward pass both are affine gathers: for a tile of consecutive rows and a fixed link the
// sources are consecutive rows, so every read is a coalesced burst and nothing is scattered (the PyTorch
// autograd path of the reference scatters a [B, N*L, C] intermediate with atomics instead).
#pragma once

#include "psf_common.h"

namespace psf {

// ------------------------------------------------------------------------------------------------------
// dV, generic: same thread layout as the generic forward kernel; links ascending, uncontracted mul/add.
// ------------------------------------------------------------------------------------------------------
template <typename T, int VEC>
__global__ void __launch_bounds__(kBlock)
chord_dv_generic_k(const T* __restrict__ dZ, const T* __restrict__ W, T* __restrict__ dV, const Geom gm,
                   const Offsets offs) {
  int b, tile, chunk;
  decode_block(gm, b, tile, chunk);
  const int TG = 1 << gm.tg_shift;
  const int g = threadIdx.x & (TG - 1);
  const int r = threadIdx.x >> gm.tg_shift;
  const int q = tile * gm.TR + r;
  const int cg = chunk * TG + g;
  if (q >= gm.N || cg >= gm.CG) return;

  const T* __restrict__ Wb = W + (int64_t)b * gm.N * gm.L;
  const T* __restrict__ Zb = dZ + (int64_t)b * gm.N * gm.C + (int64_t)cg * VEC;

  Vec<T, VEC> acc;
#pragma unroll
  for (int i = 0; i < VEC; ++i) acc.e[i] = T(0);

#pragma unroll 4
  for (int k = 0; k < gm.L; ++k) {
    int src = q - offs.v[k];
    if (src < 0) src += gm.N;
    const T w = Wb[(int64_t)src * gm.L + k];
    const Vec<T, VEC> x = ld<T, VEC>(Zb + (int64_t)src * gm.C);
    axpy_rn<T, VEC>(acc, w, x);
  }
  st<T, VEC>(dV + ((int64_t)b * gm.N + q) * gm.C + (int64_t)cg * VEC, acc);
}

// ------------------------------------------------------------------------------------------------------
// dW, generic: TG lanes share a row (TG <= 64, lanes of a row are adjacent in one wave). Each lane sums
// its own channel groups (channels ascending, uncontracted), then the TG partials are combined with a
// butterfly of wave shuffles; lane 0 of the row stores. Here the tile covers whole rows (chunks_c == 1):
// the reduction runs over all of C.
// ------------------------------------------------------------------------------------------------------
template <typename T, int VEC>
__global__ void __launch_bounds__(kBlock)
chord_dw_generic_k(const T* __restrict__ dZ, const T* __restrict__ V, T* __restrict__ dW, const Geom gm,
                   const Offsets offs) {
  int b, tile, chunk;
  decode_block(gm, b, tile, chunk);
  (void)chunk;
  const int TG = 1 << gm.tg_shift;
  const int g = threadIdx.x & (TG - 1);
  const int r = threadIdx.x >> gm.tg_shift;
  const int p = tile * gm.TR + r;
  const bool row_ok = p < gm.N;
  const int pc = row_ok ? p : gm.N - 1;  // clamp: every lane of a wave must reach the shuffles

  const T* __restrict__ Zrow = dZ + ((int64_t)b * gm.N + pc) * gm.C;
  const T* __restrict__ Vb = V + (int64_t)b * gm.v_bstride;
  T* __restrict__ dWrow = dW + ((int64_t)b * gm.N + pc) * gm.L;

  for (int k = 0; k < gm.L; ++k) {
    int src = pc + offs.v[k];
    if (src >= gm.N) src -= gm.N;
    const T* __restrict__ Vrow = Vb + (int64_t)src * gm.C;
    T part = T(0);
    for (int cg = g; cg < gm.CG; cg += TG) {
      const Vec<T, VEC> z = ld<T, VEC>(Zrow + (int64_t)cg * VEC);
      const Vec<T, VEC> x = ld<T, VEC>(Vrow + (int64_t)cg * VEC);
#pragma unroll
      for (int i = 0; i < VEC; ++i) part = add_rn(part, mul_rn(z.e[i], x.e[i]));
    }
    for (int s = TG >> 1; s > 0; s >>= 1) part = add_rn(part, __shfl_xor(part, s, 64));
    if (g == 0 && row_ok) dWrow[k] = part;
  }
}

}  // namespace psf
