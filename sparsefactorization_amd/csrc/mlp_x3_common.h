// mlp_x3_common.h — the exact three-way bf16 split of f32 operands for the bf16 matrix pipe (shared by the fused MLP
// kernels mlp_fwd_x3.hip and mlp_bwd.hip).
//
// Every f32 operand v is split EXACTLY into three bf16 terms by truncation, v = v1 + v2 + v3 (8 + 8 + 8 significant
// bits: v1 = v & 0xffff0000, v2 = (v - v1) & 0xffff0000, v3 = v - v1 - v2; each subtraction is exact), and a product
// x*w is accumulated in f32 as the six terms of magnitude >= 2^-16 |x w|: x3 w1 + x2 w2 + x1 w3 + x2 w1 + x1 w2 + x1 w1
// (small terms first). The three dropped terms are <= 2^-23 |x w| together — one f32 rounding of the product.
// v_mfma_f32_32x32x16_bf16: A-operand lane l = (row i = l & 31, k = 8 (l >> 5) + 0..7), B-operand lane l = (k = 8 (l >> 5)
// + 0..7, column j = l & 31), result register r of lane l = D[(r & 3) + 8 (r >> 2) + 4 (l >> 5)][l & 31].
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace psf_x3 {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

__device__ __forceinline__ int cd_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

// exact three-way truncation split; the results are f32 bit patterns whose low 16 bits are zero
__device__ __forceinline__ void split3(float v, uint32_t& t1, uint32_t& t2, uint32_t& t3) {
  t1 = __float_as_uint(v) & 0xffff0000u;
  const float r1 = v - __uint_as_float(t1);
  t2 = __float_as_uint(r1) & 0xffff0000u;
  t3 = __float_as_uint(r1 - __uint_as_float(t2));
}

// eight f32 bit patterns (low halves zero) -> their bf16 high halves, element i in bits [16 (i&1), +16) of dword i/2
__device__ __forceinline__ bf16x8 pack8(const uint32_t (&w)[8]) {
  uint4 d;
  d.x = __builtin_amdgcn_perm(w[1], w[0], 0x07060302u);
  d.y = __builtin_amdgcn_perm(w[3], w[2], 0x07060302u);
  d.z = __builtin_amdgcn_perm(w[5], w[4], 0x07060302u);
  d.w = __builtin_amdgcn_perm(w[7], w[6], 0x07060302u);
  return __builtin_bit_cast(bf16x8, d);
}

struct Frag3 {  // the three terms of one 8-element operand fragment
  bf16x8 t1, t2, t3;
};

__device__ __forceinline__ Frag3 split_pack8(const float (&v)[8]) {
  uint32_t a[8], b[8], c[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) split3(v[i], a[i], b[i], c[i]);
  Frag3 f;
  f.t1 = pack8(a);
  f.t2 = pack8(b);
  f.t3 = pack8(c);
  return f;
}

// The same split with the two exact subtractions done on PAIRS (v_pk_add_f32): 4.5 instead of 5.5 VALU instructions per
// value (2 ands, 2 half pk-subtractions, 1.5 perms) — the kernels that split per tile are VALU-issue-bound.
using f32x2 = __attribute__((ext_vector_type(2))) float;
__device__ __forceinline__ Frag3 split_pack8_pk(const float (&v)[8]) {
  uint32_t a[8], b[8], c[8];
#pragma unroll
  for (int i = 0; i < 8; i += 2) {
    a[i] = __float_as_uint(v[i]) & 0xffff0000u;
    a[i + 1] = __float_as_uint(v[i + 1]) & 0xffff0000u;
    const f32x2 r1 = f32x2{v[i], v[i + 1]} - f32x2{__uint_as_float(a[i]), __uint_as_float(a[i + 1])};
    b[i] = __float_as_uint(r1.x) & 0xffff0000u;
    b[i + 1] = __float_as_uint(r1.y) & 0xffff0000u;
    const f32x2 r2 = r1 - f32x2{__uint_as_float(b[i]), __uint_as_float(b[i + 1])};
    c[i] = __float_as_uint(r2.x);
    c[i + 1] = __float_as_uint(r2.y);
  }
  Frag3 f;
  f.t1 = pack8(a);
  f.t2 = pack8(b);
  f.t3 = pack8(c);
  return f;
}

// acc += sum of the six kept terms of (operand w) x (operand x), smallest first
__device__ __forceinline__ f32x16 mfma6(const Frag3& w, const Frag3& x, f32x16 acc) {
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.t1, x.t3, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.t2, x.t2, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.t3, x.t1, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.t1, x.t2, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.t2, x.t1, acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w.t1, x.t1, acc, 0, 0, 0);
  return acc;
}

// term k (0..5) of mfma6, for callers that place the six instructions one by one
__device__ __forceinline__ f32x16 mfma6_term(const Frag3& w, const Frag3& x, f32x16 acc, int k) {
  const bf16x8& a = (k == 0 || k == 3 || k == 5) ? w.t1 : (k == 1 || k == 4) ? w.t2 : w.t3;
  const bf16x8& b = k == 0 ? x.t3 : (k == 1 || k == 3) ? x.t2 : x.t1;
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
}

__device__ __forceinline__ uint16_t bf16_bits(uint32_t f32_pattern) { return (uint16_t)(f32_pattern >> 16); }

}  // namespace psf_x3
