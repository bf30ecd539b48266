// mlp_x3_store.h — how a finished Y^T accumulator tile of the split-bf16 producer kernels leaves the registers: shared by
// x3_fwd_k (mlp_fwd_x3.hip) and the lab kernel x3w_fwd_k (profiles/r06b_x3w_woven_kernel.patch). Everything in an unnamed namespace (one copy per unit).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mlp_x3_common.h"
#include "psf_common.h"

namespace {

using psf_x3::f32x16;

// 4-byte-aligned vectors: hipcc emits global_store_dwordx4 / x3 / x2 for them (rows of O floats are 16-byte aligned only
// when O is a multiple of 4; gfx950 stores unaligned vectors)
// 4-byte-aligned vectors of 4, 3, 2 floats (built-in vector types: they can be stored through an address-space pointer)
typedef float V4a __attribute__((ext_vector_type(4)));
typedef float V3a __attribute__((ext_vector_type(3)));
typedef float V2a __attribute__((ext_vector_type(2)));
typedef V4a V4u __attribute__((aligned(4)));
typedef V3a V3u __attribute__((aligned(4)));
typedef V2a V2u __attribute__((aligned(4)));
struct __attribute__((packed, aligned(4))) F4u { float x, y, z, w; };
struct __attribute__((packed, aligned(4))) F3u { float x, y, z; };
struct __attribute__((packed, aligned(4))) F2u { float x, y; };

// One token's Y row from the lane's registers (store_direct in x3_fwd_k): group q holds outputs 8 q + 4 half + (0..3) at
// yb + 32 q. OO > 0: the row length is a compile-time constant, so a group's width differs between lanes by `half` only;
// OO = 0: any length, decided per lane.
template <int OO>
__device__ __forceinline__ void store_row_groups(PSF_GLOBAL char* yb, const f32x16& y, int half, int O_rt = 0) {
  auto put = [&](int q, int n) {
    if (n >= 4) *reinterpret_cast<PSF_GLOBAL V4u*>(yb + 32 * q) = V4u{y[4 * q], y[4 * q + 1], y[4 * q + 2], y[4 * q + 3]};
    else if (n == 3) *reinterpret_cast<PSF_GLOBAL V3u*>(yb + 32 * q) = V3u{y[4 * q], y[4 * q + 1], y[4 * q + 2]};
    else if (n == 2) *reinterpret_cast<PSF_GLOBAL V2u*>(yb + 32 * q) = V2u{y[4 * q], y[4 * q + 1]};
    else if (n == 1) *reinterpret_cast<PSF_GLOBAL float*>(yb + 32 * q) = y[4 * q];
  };
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if constexpr (OO > 0) {
      constexpr int dummy = 0;
      (void)dummy;
      const int n0 = OO - 8 * q, n1 = OO - 8 * q - 4;  // floats of the group that exist for half 0 / half 1
      const int w0 = n0 >= 4 ? 4 : (n0 > 0 ? n0 : 0), w1 = n1 >= 4 ? 4 : (n1 > 0 ? n1 : 0);
      if (w0 == w1) {
        put(q, w0);
      } else if (half == 0) {
        put(q, w0);
      } else {
        put(q, w1);
      }
    } else {
      put(q, O_rt - (8 * q + 4 * half));
    }
  }
}


}  // namespace
