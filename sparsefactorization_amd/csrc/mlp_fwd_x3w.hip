// mlp_fwd_x3w.hip — x3_fwd_k (mlp_fwd_x3.hip: all M + 1 producer MLPs of a PSFNet from one read of `data`, split-bf16 matrix
// pipe at f32 accuracy; MLPBlock, SyntheticExperiments/psf.py:35-60, calls :165,175) with the units of a tile SOFTWARE-PIPELINED
// INSIDE ONE WAVE, so that a wave's matrix instructions sit in its own vector instruction stream.
//
// Why (profiles/r05a_mlp_arithmetic.md): in x3_fwd_k a wave runs, per unit, 12 dependent MFMAs, then the GELU and three-way
// split of 8 results (~130 vector instructions), 6 MFMAs, the other 8 results, 6 MFMAs. Vector time (58-61 % of the SIMD) and
// matrix time (34 %) ADD (both at once 9-11 %) whatever the occupancy or the phase of the SIMD's waves: waves do not cover for
// each other on this chip; what overlaps is a wave's own INDEPENDENT vector instructions issued behind its own MFMAs
// (coexec_lab: 43 % of the matrix time hides under independent scalar f32 work, none under work that depends on the chain or
// whose MFMA operands arrive through LDS waits, none under packed-f32 instructions).
//
// The units of a tile are independent MLPs. So unit u's work is cut into
//     M1(u)   12 MFMAs  H^T = A_u X^T           (two k-steps s = 0, 1 of six product terms each)
//     G0(u), G1(u)      GELU + split of accumulator registers 0..7 / 8..15  (scalar f32 instructions only: this unit is built
//                       with -fno-slp-vectorize)
//     M2(u)   12 MFMAs  Y^T += B_u GELU(H^T)    (s = 0 consumes G0's fragment, s = 1 G1's)
// and slot j of the wave's loop runs, in two phases of 12 MFMAs and ~160 vector instructions each,
//     phase A:   G0(j)   woven with   M2(j-1, s=1) | M1(j+1, s=0)      (two independent accumulator chains, alternating)
//     phase B:   G1(j)   woven with   M2(j,   s=0) | M1(j+1, s=1)
// every MFMA operand in registers a phase ahead (weight fragments of unit j+1 / j+2 are read from the LDS images while the
// phase before runs), one MFMA in front of every value's GELU + split, fenced with sched_barrier so that hipcc keeps the
// order. Summation orders are x3_fwd_k's: outputs are bit-identical (tests/test_gpu_producer.py, test_gpu_mixer.py).
//
// LDS: a ring of four unit images (image u is read in slots u-2 .. u and fetched by LDS-DMA in slot u-3), one workgroup
// barrier per slot as in x3_fwd_k; 75 KB per workgroup, two workgroups (two waves per SIMD) per CU.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/psf_chord.h"
#include "mlp_fwd_x3.h"
#include "mlp_x3_common.h"
#include "mlp_x3_image.h"
#include "mlp_x3_store.h"
#include "psf_common.h"

namespace {

using namespace psf_x3;

constexpr int kScrW = 32 * 36;  // per-wave scratch floats: X staging [tok][36]
constexpr int kRing = 4;

#define PSF_SB() __builtin_amdgcn_sched_barrier(0)

// GELU(x) = x Phi(x), the arithmetic of mlp_x3_image.h: gelu2 value by value (same operations, same roundings)
__device__ __forceinline__ float gelu1(float x) {
  const float t = __builtin_amdgcn_rcpf(fmaf(fabsf(x), 0.2316419f, 1.0f));
  float p = fmaf(0.53070271f, t, -0.72657602f);
  p = fmaf(p, t, 0.71070687f);
  p = fmaf(p, t, -0.14224837f);
  p = fmaf(p, t, 0.12741479f);
  p = p * t;
  const float arg = (x * x) * -0.72134752044448170368f;
  const float e = __builtin_amdgcn_exp2f(arg);
  float dlt = 0.5f - p * e;
  dlt = copysignf(dlt, x);
  return x * (0.5f + dlt);
}

struct Half3 {  // the three split terms of eight values, before packing
  uint32_t a[8], b[8], c[8];
};

__device__ __forceinline__ void gelu_split1(float x, Half3& h, int i) {
  split3(gelu1(x), h.a[i], h.b[i], h.c[i]);
}

// one MFMA of the six-term product (mlp_x3_common.h: mfma6_term), k a compile-time constant
template <int K>
__device__ __forceinline__ void mf(f32x16& acc, const Frag3& w, const Frag3& x) {
  acc = mfma6_term(w, x, acc, K);
}

__device__ __forceinline__ Frag3 ld_wa(const unsigned char* img, int c, int half, int s) {
  const unsigned char* pa = img + c * kARow + 32 * s + 16 * half;
  Frag3 f;
  f.t1 = *reinterpret_cast<const bf16x8*>(pa);
  f.t2 = *reinterpret_cast<const bf16x8*>(pa + kATerm);
  f.t3 = *reinterpret_cast<const bf16x8*>(pa + 2 * kATerm);
  return f;
}
__device__ __forceinline__ Frag3 ld_wb(const unsigned char* img, int c, int half, int s) {
  const unsigned char* pb = img + kOffB + ((s * 2 + half) * 32 + c) * 16;
  Frag3 f;
  f.t1 = *reinterpret_cast<const bf16x8*>(pb);
  f.t2 = *reinterpret_cast<const bf16x8*>(pb + kBTerm);
  f.t3 = *reinterpret_cast<const bf16x8*>(pb + 2 * kBTerm);
  return f;
}
// bias rows of the lane's 16 accumulator registers: registers 4 q .. 4 q + 3 are rows 8 q + 4 half + (0..3)
__device__ __forceinline__ f32x16 ld_bias(const float* sv, int half) {
  f32x16 r;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float4 v = *reinterpret_cast<const float4*>(sv + 8 * q + 4 * half);
    r[4 * q] = v.x, r[4 * q + 1] = v.y, r[4 * q + 2] = v.z, r[4 * q + 3] = v.w;
  }
  return r;
}

// One phase: the GELU + split of eight accumulator registers (acc[8 S .. 8 S + 7]) into a packed operand fragment, with twelve
// MFMAs — six of chain P (accP += wP (x) xP) and six of chain Q — placed one in front of every value and every packing step.
template <int S>
__device__ __forceinline__ Frag3 phase(const f32x16& accG, f32x16& accP, const Frag3& wP, const Frag3& xP, f32x16& accQ,
                                       const Frag3& wQ, const Frag3& xQ) {
  Half3 h;
  PSF_SB();
  mf<0>(accP, wP, xP); PSF_SB(); gelu_split1(accG[8 * S + 0], h, 0); PSF_SB();
  mf<0>(accQ, wQ, xQ); PSF_SB(); gelu_split1(accG[8 * S + 1], h, 1); PSF_SB();
  mf<1>(accP, wP, xP); PSF_SB(); gelu_split1(accG[8 * S + 2], h, 2); PSF_SB();
  mf<1>(accQ, wQ, xQ); PSF_SB(); gelu_split1(accG[8 * S + 3], h, 3); PSF_SB();
  mf<2>(accP, wP, xP); PSF_SB(); gelu_split1(accG[8 * S + 4], h, 4); PSF_SB();
  mf<2>(accQ, wQ, xQ); PSF_SB(); gelu_split1(accG[8 * S + 5], h, 5); PSF_SB();
  mf<3>(accP, wP, xP); PSF_SB(); gelu_split1(accG[8 * S + 6], h, 6); PSF_SB();
  mf<3>(accQ, wQ, xQ); PSF_SB(); gelu_split1(accG[8 * S + 7], h, 7); PSF_SB();
  Frag3 f;
  mf<4>(accP, wP, xP); PSF_SB(); f.t1 = pack8(h.a); PSF_SB();
  mf<4>(accQ, wQ, xQ); PSF_SB(); f.t2 = pack8(h.b); PSF_SB();
  mf<5>(accP, wP, xP); PSF_SB(); f.t3 = pack8(h.c); PSF_SB();
  mf<5>(accQ, wQ, xQ); PSF_SB();
  return f;
}

__global__ void __launch_bounds__(256, 2)
x3w_fwd_k(const X3Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int c = lane & 31, half = lane >> 5;
  float* sw = reinterpret_cast<float*>(lds_raw + kRing * kImgBytes) + wv * kScrW;
  const int E = a.E, U = a.U;
  const int64_t tiles = (a.T + 31) / 32;

  auto image = [&](int u) -> const unsigned char* { return lds_raw + (u & (kRing - 1)) * kImgBytes; };
  auto stage = [&](int u) {
    if (u >= U) return;
    const unsigned char* src = a.images + (size_t)u * kImgBytes;
    unsigned char* dst = lds_raw + (u & (kRing - 1)) * kImgBytes;
    for (int v0 = 0; v0 < kImgVecs; v0 += 256) {
      const int v = v0 + tid;
      if (v < kImgVecs)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 16 * v),
                                         (__attribute__((address_space(3))) void*)(dst + 16 * (v0 + (tid & ~63))), 16, 0, 0);
    }
  };

  for (int64_t blk = blockIdx.x; blk * 4 < tiles; blk += gridDim.x) {
    // ---- the wave's tile of X: staged through LDS, split once, kept as B-operand fragments for every unit ----
    Frag3 xf[2];
    const int64_t t0 = (blk * 4 + wv) * 32;
    {
      const int64_t rows_left = a.T - t0;
      const int nflt = (int)(rows_left >= 32 ? 32 : (rows_left > 0 ? rows_left : 0)) * E;
      const float* xt = a.X + t0 * E;
      for (int f = 4 * lane; f < 32 * E; f += 256) {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (f < nflt) v = *reinterpret_cast<const float4*>(xt + f);
        const int tok = f / E, col = f - tok * E;
        *reinterpret_cast<float4*>(sw + tok * 36 + col) = v;
      }
      if (E < 32)
        for (int i = lane; i < 32 * 8; i += 64) {
          const int tok = i >> 3, col = 4 * (i & 7);
          if (col >= E) *reinterpret_cast<float4*>(sw + tok * 36 + col) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const float4 lo = *reinterpret_cast<const float4*>(sw + c * 36 + 16 * s + 8 * half);
        const float4 hi = *reinterpret_cast<const float4*>(sw + c * 36 + 16 * s + 8 * half + 4);
        const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        xf[s] = split_pack8(v);
      }
    }
    __syncthreads();  // the previous block's last slot is done with every image buffer
    stage(0);
    stage(1);
    stage(2);
    __syncthreads();  // images 0..2 have landed (hipcc drains vmcnt before the barrier)
    stage(3);

    // ---- prologue: M1(0) alone; the fragments phase A of slot 0 needs ----
    f32x16 accC = ld_bias(reinterpret_cast<const float*>(image(0) + kOffSa), half);  // H^T of the unit whose GELU runs
    {
      const Frag3 w0 = ld_wa(image(0), c, half, 0), w1 = ld_wa(image(0), c, half, 1);
      accC = mfma6(w0, xf[0], accC);
      accC = mfma6(w1, xf[1], accC);
    }
    Frag3 was0 = ld_wa(image(1), c, half, 0);  // A fragment, k-step 0, of unit j + 1 (garbage past the last unit: discarded)
    f32x16 accN = ld_bias(reinterpret_cast<const float*>(image(1) + kOffSa), half);  // H^T of unit j + 1, being accumulated
    f32x16 acc2;                                // Y^T of the current MLP
    Frag3 fprev, wbs1;                          // G1's fragment and the B' fragment, k-step 1, of unit j - 1
#pragma unroll
    for (int r = 0; r < 16; ++r) acc2[r] = 0.f;
    fprev.t1 = fprev.t2 = fprev.t3 = xf[0].t1;  // (slot 0 multiplies these into an acc2 that `first` then overwrites)
    wbs1 = was0;

    for (int j = 0; j < U; ++j) {
      if (j > 0) {
        __syncthreads();  // image j + 2 has landed; every wave is done with slot j - 1 (buffer (j + 3) & 3 is free)
        stage(j + 3);
      }
      const uint32_t uj = a.unit[j];
      const bool first = ((uj >> 8) & 0xff) == 0, last = (uj >> 16) != 0;
      const uint32_t ujp = a.unit[j > 0 ? j - 1 : 0];
      const bool last_prev = j > 0 && (ujp >> 16) != 0;

      // phase A: G0(j) with M2(j-1, s=1) and M1(j+1, s=0); meanwhile the fragments of phase B arrive
      const Frag3 was1 = ld_wa(image(j + 1), c, half, 1);
      const Frag3 wbs0 = ld_wb(image(j), c, half, 0);
      const Frag3 f0 = phase<0>(accC, acc2, wbs1, fprev, accN, was0, xf[0]);
      if (last_prev) {  // unit j - 1 closed its MLP: its Y^T tile leaves from the accumulator registers
        const X3Mlp& dp = a.m[ujp & 0xff];
        const int O = dp.O;
        if (t0 + c < a.T) {
          PSF_GLOBAL char* yb = psf::sbase(reinterpret_cast<char*>(dp.Y + t0 * O)) + ((uint32_t)c * (uint32_t)(O * 4) + (uint32_t)half * 16u);
          switch (O) {
#define PSF_O(OO) \
  case OO:        \
    store_row_groups<OO>(yb, acc2, half); \
    break;
            PSF_O(4) PSF_O(5) PSF_O(6) PSF_O(7) PSF_O(8) PSF_O(9) PSF_O(10) PSF_O(11) PSF_O(12) PSF_O(13) PSF_O(14) PSF_O(15)
            PSF_O(16) PSF_O(17) PSF_O(18) PSF_O(19) PSF_O(20) PSF_O(32)
#undef PSF_O
            default:
              store_row_groups<0>(yb, acc2, half, O);
          }
        }
      }
      if (first) acc2 = ld_bias(reinterpret_cast<const float*>(image(j) + kOffSb), half);

      // phase B: G1(j) with M2(j, s=0) and M1(j+1, s=1); meanwhile the fragments of the next phase A arrive
      was0 = ld_wa(image(j + 2), c, half, 0);
      wbs1 = ld_wb(image(j), c, half, 1);
      fprev = phase<1>(accC, acc2, wbs0, f0, accN, was1, xf[1]);
      accC = accN;
      accN = ld_bias(reinterpret_cast<const float*>(image(j + 2) + kOffSa), half);
      (void)last;
    }
    // ---- epilogue: M2(U-1, s=1) alone, then the last MLP's tile ----
    acc2 = mfma6(wbs1, fprev, acc2);
    {
      const uint32_t ul = a.unit[U - 1];
      const X3Mlp& dp = a.m[ul & 0xff];
      const int O = dp.O;
      if (t0 + c < a.T) {
        PSF_GLOBAL char* yb = psf::sbase(reinterpret_cast<char*>(dp.Y + t0 * O)) + ((uint32_t)c * (uint32_t)(O * 4) + (uint32_t)half * 16u);
        store_row_groups<0>(yb, acc2, half, O);
      }
    }
  }
}

}  // namespace

hipError_t psf_x3w_mlp_fwd_launch(const float* X, int64_t T, int32_t E, int32_t K, const float* const* A,
                                  const float* const* a, const float* const* B, const float* const* b, const int32_t* h,
                                  const int32_t* O, float* const* Y, void* workspace, hipStream_t s) {
  X3Plan p;
  if (!x3_make_plan(E, K, h, O, &p)) return hipErrorInvalidValue;
  X3Args args;
  x3_fill_args(p, X, T, E, K, A, a, B, b, h, O, Y, workspace, &args);
  hipLaunchKernelGGL(x3_pack_k, dim3(p.U), dim3(256), 0, s, args);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return e;
  const int64_t tiles = (T + 31) / 32;
  const int64_t blocks_needed = (tiles + 3) / 4;
  const int grid = (int)(blocks_needed < 4096 ? blocks_needed : 4096);
  const size_t lds = kRing * (size_t)kImgBytes + 4 * (size_t)kScrW * sizeof(float);
  e = hipFuncSetAttribute((const void*)x3w_fwd_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(x3w_fwd_k, dim3(grid), dim3(256), lds, s, args);
  return hipGetLastError();
}
