// mlp_x3_image.h — the packed weight image of one 32-row hidden unit of a token-wise MLP Linear(E,h) -> GELU -> Linear(h,O)
// (MLPBlock, SyntheticExperiments/psf.py:35-60) for the split-bf16 matrix-pipe kernels, the kernel that writes it and the
// GELU those kernels share. Used by mlp_fwd_x3.hip (the producer forward) and fwd_mlp_step.h (the chain step that computes
// its own W tile). Everything is in an unnamed namespace: each translation unit gets its own copy of the pack kernel.
//
// Orientation (as mlp_fwd_x3.hip): GEMM1 H^T[j][tok] = A_u X^T, GEMM2 Y^T[o][tok] += B_u[o][rho] H^T with the accumulator of
// GEMM1 used as the B operand of GEMM2 as it stands (register r of lane (tok, half) is hidden row rho(r, half) = (r&3) +
// 8(r>>2) + 4 half, so registers 8s..8s+7 are the B fragment of k-step s when the second layer's weights are packed in that
// order).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mlp_x3_common.h"

namespace {

// unit image (bytes): A terms 3 x [32 j][80 B: 32 bf16 + pad] | sa 32 f32 | B' terms 3 x [2 s][2 half][32 o][8 bf16]
// | sb 32 f32
constexpr int kARow = 80;                  // bytes; 20-dword stride: ds_read_b128 conflict-free over 16 lanes
constexpr int kATerm = 32 * kARow;         // 2560
constexpr int kOffSa = 3 * kATerm;         // 7680
constexpr int kOffB = kOffSa + 128;        // 7808
constexpr int kBTerm = 2 * 2 * 32 * 16;    // 2048
constexpr int kOffSb = kOffB + 3 * kBTerm; // 13952
constexpr int kImgBytes = kOffSb + 128;    // 14080
constexpr int kImgVecs = kImgBytes / 16;   // 880

struct X3Mlp {
  const float* A;
  const float* a;
  const float* B;
  const float* b;
  float* Y;
  int32_t h, O;
};

struct X3Args {
  X3Mlp m[32];
  // unit -> MLP | hidden block << 8 | (last unit of its MLP) << 16. Dwords: a byte table indexed by the unit counter is read
  // with global_load_ubyte + s_waitcnt vmcnt(0) (a full memory round trip per unit); a dword table with one s_load_dword.
  uint32_t unit[128];
  const float* X;
  unsigned char* images;
  int64_t T;
  int32_t E, U;
};

// GELU(x) = x Phi(x) for a PAIR of values, on packed f32 math (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32: two
// elements per instruction — the kernel is VALU-issue-bound, rocprofv3: ~70 % VALU-busy, 22 % matrix-pipe-busy).
// Phi by Abramowitz & Stegun 26.2.17 (the normal-CDF form of 7.1.26, |error| <= 7.5e-8):
//   t = 1 / (1 + 0.2316419 |x|),  q = exp(-x^2/2)/sqrt(2 pi) * (b1 t + ... + b5 t^5),  Phi = x >= 0 ? 1 - q : q
// with 1/sqrt(2 pi) folded into the b's and exp as exp2(x^2 * -0.5 log2 e): 19 instructions per pair. (The same arithmetic on
// scalar f32 instructions with -fno-slp-vectorize, in the step kernel of fwd_mlp_step.h: 821.7 against 809.6 us per mixer
// forward at cfg2, one process, seven rounds — packed stays.)
using f32x2 = __attribute__((ext_vector_type(2))) float;
__device__ __forceinline__ f32x2 gelu2(f32x2 x) {
  f32x2 t;
  t.x = __builtin_amdgcn_rcpf(fmaf(fabsf(x.x), 0.2316419f, 1.0f));
  t.y = __builtin_amdgcn_rcpf(fmaf(fabsf(x.y), 0.2316419f, 1.0f));
  const f32x2 B5 = {0.53070271f, 0.53070271f}, B4 = {-0.72657602f, -0.72657602f}, B3 = {0.71070687f, 0.71070687f},
              B2 = {-0.14224837f, -0.14224837f}, B1 = {0.12741479f, 0.12741479f};
  f32x2 p = __builtin_elementwise_fma(B5, t, B4);
  p = __builtin_elementwise_fma(p, t, B3);
  p = __builtin_elementwise_fma(p, t, B2);
  p = __builtin_elementwise_fma(p, t, B1);
  p = p * t;
  const f32x2 c2 = {-0.72134752044448170368f, -0.72134752044448170368f};
  const f32x2 arg = (x * x) * c2;
  f32x2 e;
  e.x = __builtin_amdgcn_exp2f(arg.x);
  e.y = __builtin_amdgcn_exp2f(arg.y);
  const f32x2 half2 = {0.5f, 0.5f};
  f32x2 dlt = half2 - p * e;  // 0.5 - q >= 0
  dlt.x = copysignf(dlt.x, x.x);
  dlt.y = copysignf(dlt.y, x.y);
  return x * (half2 + dlt);  // Phi = 0.5 + sign(x) (0.5 - q)
}

// One workgroup per unit: split the weights and write them in operand order.
__attribute__((unused)) __global__ void __launch_bounds__(256) x3_pack_k(const X3Args a) {  // (not every includer packs)
  using namespace psf_x3;
  const int u = blockIdx.x;
  const X3Mlp d = a.m[a.unit[u] & 0xff];
  const int ht = 32 * (int)((a.unit[u] >> 8) & 0xff), E = a.E;
  unsigned char* img = a.images + (size_t)u * kImgBytes;
  uint16_t* img16 = reinterpret_cast<uint16_t*>(img);
  float* img32 = reinterpret_cast<float*>(img);
  // A terms: [j][e]
  for (int i = threadIdx.x; i < 32 * 40; i += 256) {
    const int j = i / 40, e = i - j * 40;
    const float v = (e < E && ht + j < d.h) ? d.A[(ht + j) * E + e] : 0.f;
    uint32_t t1, t2, t3;
    split3(v, t1, t2, t3);
    img16[(0 * kATerm + j * kARow) / 2 + e] = bf16_bits(t1);
    img16[(1 * kATerm + j * kARow) / 2 + e] = bf16_bits(t2);
    img16[(2 * kATerm + j * kARow) / 2 + e] = bf16_bits(t3);
  }
  for (int j = threadIdx.x; j < 32; j += 256) {
    img32[kOffSa / 4 + j] = ht + j < d.h ? d.a[ht + j] : 0.f;
    img32[kOffSb / 4 + j] = j < d.O ? d.b[j] : 0.f;
  }
  // B' terms: [s][half][o][i] = B[o][ht + rho], rho = (i&3) + 16 s + 8 (i>>2) + 4 half
  for (int q = threadIdx.x; q < 2 * 2 * 32 * 8; q += 256) {
    const int i = q & 7, o = (q >> 3) & 31, hf = (q >> 8) & 1, s = q >> 9;
    const int rho = (i & 3) + 16 * s + 8 * (i >> 2) + 4 * hf;
    const float v = (o < d.O && ht + rho < d.h) ? d.B[o * d.h + ht + rho] : 0.f;
    uint32_t t1, t2, t3;
    split3(v, t1, t2, t3);
    img16[(kOffB + 0 * kBTerm) / 2 + q] = bf16_bits(t1);
    img16[(kOffB + 1 * kBTerm) / 2 + q] = bf16_bits(t2);
    img16[(kOffB + 2 * kBTerm) / 2 + q] = bf16_bits(t3);
  }
}

// unit table of K MLPs: every MLP is cut into ceil(h/32) units of 32 hidden rows, MLP by MLP
struct X3Plan {
  int U;
  uint8_t unit_k[128], unit_hb[128], unit_last[128];
  int first_unit[33];  // first unit of MLP k; first_unit[K] = U
};

inline bool x3_make_plan(int32_t E, int32_t K, const int32_t* h, const int32_t* O, X3Plan* p) {
  if (E < 4 || E > 32 || (E & 3) || K < 1 || K > 32 || !h || !O) return false;
  p->U = 0;
  for (int k = 0; k < K; ++k) {
    if (h[k] < 1 || h[k] > 128 || O[k] < 1 || O[k] > 32) return false;
    const int nb = (h[k] + 31) / 32;
    p->first_unit[k] = p->U;
    for (int hb = 0; hb < nb; ++hb) {
      p->unit_k[p->U] = (uint8_t)k;
      p->unit_hb[p->U] = (uint8_t)hb;
      p->unit_last[p->U] = hb == nb - 1;
      ++p->U;
    }
  }
  p->first_unit[K] = p->U;
  return true;
}

inline void x3_fill_args(const X3Plan& p, const float* X, int64_t T, int32_t E, int32_t K, const float* const* A,
                         const float* const* a, const float* const* B, const float* const* b, const int32_t* h,
                         const int32_t* O, float* const* Y, void* workspace, X3Args* args) {
  for (int k = 0; k < 32; ++k) args->m[k] = X3Mlp{nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0};
  for (int k = 0; k < K; ++k) args->m[k] = X3Mlp{A[k], a[k], B[k], b[k], Y ? Y[k] : nullptr, h[k], O[k]};
  for (int u = 0; u < 128; ++u)
    args->unit[u] = u < p.U ? ((uint32_t)p.unit_k[u] | ((uint32_t)p.unit_hb[u] << 8) | ((uint32_t)p.unit_last[u] << 16)) : 0u;
  args->X = X;
  args->images = reinterpret_cast<unsigned char*>(workspace);
  args->T = T;
  args->E = E;
  args->U = p.U;
}

}  // namespace
