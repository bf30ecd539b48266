// mlp_planes.h — 32x32 bf16 operand planes in LDS that serve BOTH orientations of a v_mfma_f32_32x32x16_bf16 operand
// (used by the split-bf16 backward of the fused MLPs, mlp_bwd.hip; checked in isolation by profiles/trlab.hip).
//
// A plane is [32 rows][32 columns] of bf16 in 64-byte rows; the 16-byte chunk ch (columns 8 ch .. 8 ch + 7) of row r
// lives at  plane_off(r, ch) = 64 r + 16 (ch ^ ((r ^ (r >> 2)) & 3)).  With that XOR
//   * the ROW read of an operand (lane (c, half) takes row c, chunk 2 s + half: one ds_read_b128) is conflict-free,
//   * the TRANSPOSED read (ds_read_b64_tr_b16: a 16-lane group fetches a block of 4 rows x 16 columns and lane i of the
//     group receives column i, the block's row q in element q) is conflict-free,
//   * the accumulator-order store (lane (row c, half) writes 4 consecutive columns, ds_write_b64) is 2-way, which costs
//     8 LDS cycles against the instruction's own 6
// (bank rules of MI355X_MICROARCH.md, LDS table; enumerated offline for all 32 linear swizzles).
//
// Operand fragments of k-step s (8 bf16 per lane; c = lane & 31, half = lane >> 5):
//   row_frag    operand[c][16 s + 8 half + i]                       = plane[c][16 s + 8 half + i]      (one b128)
//   tr_frag     operand[c][16 s + 8 half + i]                       = plane[16 s + 8 half + i][c]      (two tr reads)
//   tr_frag_acc operand[c][i] with the contraction index in the ACCUMULATOR's register order
//               rho(s, half, i) = (i & 3) + 16 s + 8 (i >> 2) + 4 half  = plane[rho(s, half, i)][c]     (two tr reads)
//               — the order in which registers 8 s .. 8 s + 7 of a 32x32 accumulator tile are a B fragment as they stand.
// ds_read_b64_tr_b16 needs EXEC all ones: call these only from wave-uniform control flow.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mlp_x3_common.h"

namespace psf_x3 {

constexpr int kPlaneBytes = 32 * 64;

__device__ __forceinline__ int plane_off(int row, int ch) { return 64 * row + 16 * (ch ^ ((row ^ (row >> 2)) & 3)); }

using s16x4 = __attribute__((ext_vector_type(4))) short;
using lds_s16x4_ptr = __attribute__((address_space(3))) s16x4*;

__device__ __forceinline__ s16x4 tr_read(const unsigned char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)(p));
}

// The lane's byte offsets into a plane, computed once per kernel.
struct PlaneLane {
  int row[2];     // row_frag / row-chunk store of k-step s: plane_off(c, 2 s + half)
  int tr[2];      // tr_frag, read t (elements 4 t .. 4 t + 3) of k-step 0; k-step 1 is + 1024
  int tr_acc[2];  // tr_frag_acc, read t of k-step 0; k-step 1 is + 1024
  int acc_st;     // accumulator-order store: 64 c + 8 half; chunk g goes to acc_st + 16 (g ^ acc_sw)
  int acc_sw;
};

__device__ __forceinline__ PlaneLane plane_lane(int lane) {
  const int c = lane & 31, half = lane >> 5;
  const int q = (lane >> 2) & 3, p = lane & 3, cg = (lane >> 4) & 1;
  PlaneLane L;
#pragma unroll
  for (int s = 0; s < 2; ++s) L.row[s] = plane_off(c, 2 * s + half);
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    L.tr[t] = plane_off(8 * half + 4 * t + q, 2 * cg + (p >> 1)) + 8 * (p & 1);
    L.tr_acc[t] = plane_off(8 * t + 4 * half + q, 2 * cg + (p >> 1)) + 8 * (p & 1);
  }
  L.acc_st = 64 * c + 8 * half;
  L.acc_sw = (c ^ (c >> 2)) & 3;
  return L;
}

__device__ __forceinline__ bf16x8 join8(s16x4 lo, s16x4 hi) {
  using s16x8 = __attribute__((ext_vector_type(8))) short;
  const s16x8 v = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
  return __builtin_bit_cast(bf16x8, v);
}

__device__ __forceinline__ bf16x8 row_frag(const unsigned char* plane, const PlaneLane& L, int s) {
  return *reinterpret_cast<const bf16x8*>(plane + L.row[s]);
}
__device__ __forceinline__ bf16x8 tr_frag(const unsigned char* plane, const PlaneLane& L, int s) {
  return join8(tr_read(plane + L.tr[0] + 1024 * s), tr_read(plane + L.tr[1] + 1024 * s));
}
__device__ __forceinline__ bf16x8 tr_frag_acc(const unsigned char* plane, const PlaneLane& L, int s) {
  return join8(tr_read(plane + L.tr_acc[0] + 1024 * s), tr_read(plane + L.tr_acc[1] + 1024 * s));
}

// Sixteen accumulator registers split into the three bf16 terms, packed in pairs: dword m of term t holds registers
// 2 m (low half) and 2 m + 1. Dwords 4 s .. 4 s + 3 are the B fragment of k-step s (accumulator-as-operand); dwords
// 2 g, 2 g + 1 are the four consecutive rows 8 g + 4 half .. + 3 that store_acc_plane writes as one ds_write_b64.
struct Split16 {
  uint32_t d[3][8];
};

// registers 2 m and 2 m + 1 of the tile -> dword m of the three terms
__device__ __forceinline__ void split16_pair(float v0, float v1, Split16& r, int m) {
  const uint32_t a0 = __float_as_uint(v0) & 0xffff0000u, a1 = __float_as_uint(v1) & 0xffff0000u;
  // scalar subtractions: packed f32 VALU is slow beside the sibling wave's MFMAs (see gelu_and_grad1 in mlp_bwd.hip)
  const float r1x = v0 - __uint_as_float(a0), r1y = v1 - __uint_as_float(a1);
  const uint32_t b0 = __float_as_uint(r1x) & 0xffff0000u, b1 = __float_as_uint(r1y) & 0xffff0000u;
  const f32x2 r2 = {r1x - __uint_as_float(b0), r1y - __uint_as_float(b1)};
  r.d[0][m] = __builtin_amdgcn_perm(a1, a0, 0x07060302u);
  r.d[1][m] = __builtin_amdgcn_perm(b1, b0, 0x07060302u);
  r.d[2][m] = __builtin_amdgcn_perm(__float_as_uint(r2.y), __float_as_uint(r2.x), 0x07060302u);
}

__device__ __forceinline__ Split16 split16(const float (&v)[16]) {
  Split16 r;
#pragma unroll
  for (int i = 0; i < 16; i += 2) split16_pair(v[i], v[i + 1], r, i >> 1);
  return r;
}

__device__ __forceinline__ bf16x8 acc_frag(const Split16& x, int t, int s) {
  uint4 d;
  d.x = x.d[t][4 * s];
  d.y = x.d[t][4 * s + 1];
  d.z = x.d[t][4 * s + 2];
  d.w = x.d[t][4 * s + 3];
  return __builtin_bit_cast(bf16x8, d);
}

// term t of a tile held in accumulator layout (lane = column c of the tile, registers = its rows) -> plane[c][row]:
// the transposed image, from which tr_frag delivers operand[row][k = c...].
__device__ __forceinline__ void store_acc_plane(unsigned char* plane, const PlaneLane& L, const Split16& x, int t) {
#pragma unroll
  for (int g = 0; g < 4; ++g)
    *reinterpret_cast<uint2*>(plane + L.acc_st + 16 * (g ^ L.acc_sw)) = uint2{x.d[t][2 * g], x.d[t][2 * g + 1]};
}

}  // namespace psf_x3
