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
//
// This file: the generic direct-gather kernel (any shape). The fast path is fwd_window.h.
#pragma once

#include "psf_common.h"

namespace psf {

// Any N, any L <= 64, any C (VEC = 1 when C is not a multiple of the 16-byte vector width), any offsets.
// Thread (r, g) owns VEC channels of one row and walks the links in order; the TG lanes of a row read the
// same W element (a wave-level broadcast).
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

}  // namespace psf
