// mlp_fwd_x3.hip — the fused producer-MLP forward of mlp_fwd.hip on the bf16 matrix pipe, at f32 accuracy.
//
// Why: v_mfma_f32_32x32x2_f32 shares the vector ALU's datapath — MFMA cycles and VALU cycles ADD (profiles/
// r01_mfmalab.log: 64-cycle MFMA + 14 VALU = 130 cycles per SIMD at any occupancy), which pins mlp_fwd.hip at
// 32 x 64 + ~1500 cycles per tile-MLP. v_mfma_f32_32x32x16_bf16 runs on the separate matrix pipe (32 cycles, of
// which 8 hold the vector issue) and overlaps with VALU work (same log: 40-cycle MFMA + 16 VALU = 76 cycles).
//
// How f32 accuracy survives bf16 operands: every f32 operand v is split EXACTLY into three bf16 terms by
// truncation, v = v1 + v2 + v3 (8 + 8 + 8 significant bits: v1 = v & 0xffff0000, v2 = (v - v1) & 0xffff0000,
// v3 = v - v1 - v2; each subtraction is exact), and a product x*w is accumulated in f32 as the six terms of
// magnitude >= 2^-16 |x w|:  x3 w1 + x2 w2 + x1 w3 + x2 w1 + x1 w2 + x1 w1  (small terms first). The three
// dropped terms are <= 2^-23 |x w| together — the size of one f32 rounding of the product. Per GEMM of K = 32
// that is 2 k-steps x 6 = 12 MFMAs of 32 cycles instead of 16 of 64, on a pipe that leaves the VALU free for the
// GELU and the splitting (4 VALU + 1.5 v_perm per value).
//
// Structure (as mlp_fwd.hip): a wave owns one tile of 32 tokens and keeps X's split B-operand fragments in
// registers for all MLPs; a hidden layer wider than 32 is processed as 32-row units whose second-GEMM results
// accumulate in the same registers; weights are packed once per call into LDS images (already split, already in
// operand order) and stream through two LDS buffers by LDS-DMA, one barrier per unit.
//   GEMM1  H^T[j][tok] = A_u X^T: A operand = 8 consecutive e of row j (one ds_read_b128 per term and k-step)
//   GELU on the accumulator registers, then split + pack: register r of lane (tok, half) is hidden row
//          rho(r, half) = (r&3) + 8(r>>2) + 4 half, so registers 8s..8s+7 ARE the lane's B-operand fragment of
//          k-step s for the second GEMM if the weights are packed in the same order (accumulator-as-operand)
//   GEMM2  Y^T[o][tok] += B_u[o][rho] H^T: A operand prepacked as [s][half][o][8]
// Limits of this variant: E <= 32 (multiple of 4), h <= 128, O <= 32, K <= 32.
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

constexpr int kScr = 32 * 36;              // per-wave scratch floats: X staging [tok][36]

#ifdef PSF_X3F_TRACE  // profiles/x3flab.hip: shader-clock timestamps of one unit of one workgroup, per wave
__device__ unsigned long long psf_x3f_trace[4][16];
#define PSF_FTRACE(slot)                                                       \
  do {                                                                         \
    if (trace_on) {                                                            \
      const unsigned long long t_ = clock64();                                 \
      if (lane == 0) psf_x3f_trace[wv][(slot)] = t_;                           \
    }                                                                          \
  } while (0)
#else
#define PSF_FTRACE(slot) do { } while (0)
#endif

// Finished Y^T tiles leave straight from the accumulator registers at the end of their unit. (Rounds 2-3 parked them in LDS and
// stored them as contiguous bursts at the start of the next unit — equal bits, 1-10 % more time, profiles/r03ag_mlp_fwd_store_ab.log —
// and had a second instance with two software-pipelined tiles per wave at two workgroups per CU — 4-6 % slower at every length,
// r03ak_mlp_fwd_tpw_ab.log, and no faster with a scalar GELU, r05p_x3f_scalar_gelu_tpw.log. Both went in round 5 with their knobs.)
// One tile per wave: 146 registers, three workgroups (three waves per SIMD) per CU.
__global__ void __launch_bounds__(256, 2)
x3_fwd_k(const X3Args a) {
  constexpr int TPW = 1;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int c = lane & 31, half = lane >> 5;
  float* sw = reinterpret_cast<float*>(lds_raw + 2 * kImgBytes) + wv * (TPW * kScr);  // TPW tiles of [32][36]
  const int E = a.E, U = a.U;
  const int64_t tiles = (a.T + 31) / 32;
  const int64_t tiles_per_block = 4 * TPW;

  // (The inline-assembly form of this DMA that mlp_bwd.hip uses — hidden from hipcc's alias rule, explicit vmcnt(0) before the
  // barrier — was tried here too: its M0 save / restore per piece made the issue take 1050 instead of 420 clocks per unit and
  // the kernel 4 % slower.)
  auto stage = [&](int u) {
    const unsigned char* src = a.images + (size_t)u * kImgBytes;
    unsigned char* dst = lds_raw + (u & 1) * kImgBytes;
    for (int v0 = 0; v0 < kImgVecs; v0 += 256) {
      const int v = v0 + tid;
      if (v < kImgVecs)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 16 * v),
                                         (__attribute__((address_space(3))) void*)(dst + 16 * (v0 + (tid & ~63))), 16, 0, 0);
    }
  };

  for (int64_t blk = blockIdx.x; blk * tiles_per_block < tiles; blk += gridDim.x) {
    Frag3 xf[TPW][2];  // X fragments: k-step s covers e = 16 s + 8 half + (0..7) of the lane's token
    f32x16 acc2[TPW];
    int64_t t0[TPW];
#pragma unroll
    for (int tp = 0; tp < TPW; ++tp) {
      t0[tp] = (blk * tiles_per_block + wv * TPW + tp) * 32;
      const int64_t rows_left = a.T - t0[tp];
      const int nflt = (int)(rows_left >= 32 ? 32 : (rows_left > 0 ? rows_left : 0)) * E;
      const float* xt = a.X + t0[tp] * E;
      for (int f = 4 * lane; f < 32 * E; f += 256) {  // coalesced 16-byte loads of the contiguous tile
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (f < nflt) v = *reinterpret_cast<const float4*>(xt + f);
        const int tok = f / E, col = f - tok * E;
        *reinterpret_cast<float4*>(sw + tok * 36 + col) = v;  // E % 4 == 0: 16-byte aligned, never straddles a row
      }
      if (E < 32)
        for (int i = lane; i < 32 * 8; i += 64) {  // zero the columns >= E (groups of 4)
          const int tok = i >> 3, col = 4 * (i & 7);
          if (col >= E) *reinterpret_cast<float4*>(sw + tok * 36 + col) = make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const float4 lo = *reinterpret_cast<const float4*>(sw + c * 36 + 16 * s + 8 * half);
        const float4 hi = *reinterpret_cast<const float4*>(sw + c * 36 + 16 * s + 8 * half + 4);
        const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        xf[tp][s] = split_pack8(v);
      }
    }
    __syncthreads();  // the previous block's last unit is done with both image buffers
    stage(0);

    // The lane holds Y^T[o = 8 q + 4 half + (0..3)][tok = c] in registers 4 q .. 4 q + 3: up to four consecutive floats of its
    // token's row go out as one (unaligned) vector store per q — no LDS transposition, no parked tile, no flush. Addresses: the
    // tile's first row as a scalar base, the lane's token and half as one 32-bit offset, the q groups as immediates
    // (psf_common.h: sbase). How many of a group's four floats exist depends on O and on the lane's half only; O is
    // wave-uniform, so the widths PSFNet uses (C = 4, 8, 16, 32 for g; L = 4..20 for the link MLPs) are compiled in —
    // one scalar branch on O, then per group either one store for every lane or one per half — and only other widths take
    // the form that decides per lane and per group among four store widths (28 lane-masked blocks per unit).
    auto store_direct = [&](int k, const f32x16 (&y)[TPW]) {
      const X3Mlp& dp = a.m[k];
      const int O = dp.O;
      const uint32_t lo = (uint32_t)c * (uint32_t)(O * 4) + (uint32_t)half * 16u;
#pragma unroll
      for (int tp = 0; tp < TPW; ++tp) {
        if (t0[tp] + c < a.T) {
          PSF_GLOBAL char* yb = psf::sbase(reinterpret_cast<char*>(dp.Y + t0[tp] * O)) + lo;
          switch (O) {
#define PSF_O(OO) \
  case OO:        \
    store_row_groups<OO>(yb, y[tp], half); \
    break;
            PSF_O(4) PSF_O(5) PSF_O(6) PSF_O(7) PSF_O(8) PSF_O(9) PSF_O(10) PSF_O(11) PSF_O(12) PSF_O(13) PSF_O(14) PSF_O(15)
            PSF_O(16) PSF_O(17) PSF_O(18) PSF_O(19) PSF_O(20) PSF_O(32)
#undef PSF_O
            default:
              store_row_groups<0>(yb, y[tp], half, O);
          }
        }
      }
    };
    for (int u = 0; u < U; ++u) {
#ifdef PSF_X3F_TRACE
      const bool trace_on = blockIdx.x == 700 && u == 5;
#endif
      PSF_FTRACE(0);
      __syncthreads();  // image u has landed (hipcc drains vmcnt before the barrier); unit u-1 is finished
      PSF_FTRACE(1);
      if (u + 1 < U) stage(u + 1);
      PSF_FTRACE(6);
      PSF_FTRACE(2);
      const unsigned char* img = lds_raw + (u & 1) * kImgBytes;
      const float* sa = reinterpret_cast<const float*>(img + kOffSa);
      const float* sb = reinterpret_cast<const float*>(img + kOffSb);
      const bool first = ((a.unit[u] >> 8) & 0xff) == 0, last = (a.unit[u] >> 16) != 0;

      // weight fragments of this unit (shared by the wave's tiles)
      Frag3 wa[2], wb[2];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const unsigned char* pa = img + c * kARow + 32 * s + 16 * half;
        wa[s].t1 = *reinterpret_cast<const bf16x8*>(pa);
        wa[s].t2 = *reinterpret_cast<const bf16x8*>(pa + kATerm);
        wa[s].t3 = *reinterpret_cast<const bf16x8*>(pa + 2 * kATerm);
        const unsigned char* pb = img + kOffB + ((s * 2 + half) * 32 + c) * 16;
        wb[s].t1 = *reinterpret_cast<const bf16x8*>(pb);
        wb[s].t2 = *reinterpret_cast<const bf16x8*>(pb + kBTerm);
        wb[s].t3 = *reinterpret_cast<const bf16x8*>(pb + 2 * kBTerm);
      }

      if (first) {
#pragma unroll
        for (int tp = 0; tp < TPW; ++tp)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc2[tp][r] = sb[cd_row(r, half)];
      }
      PSF_FTRACE(3);
      auto half_gelu = [&](const f32x16& acc1, int s) {  // GELU + split of registers 8s..8s+7: one B fragment
        float g[8];
#pragma unroll
        for (int i = 0; i < 8; i += 2) {  // (scalar f32 instead of packed: no gain here, profiles/r02ai_x3flab.log)
          const f32x2 y = gelu2(f32x2{acc1[8 * s + i], acc1[8 * s + i + 1]});
          g[i] = y.x;
          g[i + 1] = y.y;
        }
        return split_pack8(g);
      };

      {
#pragma unroll
        for (int tp = 0; tp < TPW; ++tp) {
          f32x16 acc1;
#pragma unroll
          for (int r = 0; r < 16; ++r) acc1[r] = sa[cd_row(r, half)];
          // (s_setprio by phase — 1 around the matrix chains, 0 around the GELU halves, what gained 3 % in the backward kernel —
          // costs 2 % here at three waves per SIMD: profiles/r03an_x3f_prio.log)
          acc1 = mfma6(wa[0], xf[tp][0], acc1);
          acc1 = mfma6(wa[1], xf[tp][1], acc1);
          acc2[tp] = mfma6(wb[0], half_gelu(acc1, 0), acc2[tp]);
          acc2[tp] = mfma6(wb[1], half_gelu(acc1, 1), acc2[tp]);
        }
      }

      PSF_FTRACE(4);
      if (last) store_direct((int)(a.unit[u] & 0xff), acc2);
      PSF_FTRACE(5);
    }
  }
}

static_assert(kX3ImageBytes == kImgBytes, "mlp_fwd_x3.h states the image size for callers outside this file");

}  // namespace

int64_t psf_x3_mlp_fwd_workspace(int32_t E, int32_t K, const int32_t* h, const int32_t* O) {
  X3Plan p;
  if (!x3_make_plan(E, K, h, O, &p)) return -1;
  return (int64_t)p.U * kImgBytes;
}

hipError_t psf_x3_pack_launch(int32_t E, int32_t K, const float* const* A, const float* const* a, const float* const* B,
                              const float* const* b, const int32_t* h, const int32_t* O, void* workspace,
                              int32_t* first_unit, hipStream_t s) {
  X3Plan p;
  if (!x3_make_plan(E, K, h, O, &p)) return hipErrorInvalidValue;
  X3Args args;
  x3_fill_args(p, nullptr, 0, E, K, A, a, B, b, h, O, nullptr, workspace, &args);
  if (first_unit)
    for (int k = 0; k <= K; ++k) first_unit[k] = p.first_unit[k];
  hipLaunchKernelGGL(x3_pack_k, dim3(p.U), dim3(256), 0, s, args);
  return hipGetLastError();
}

hipError_t psf_x3_mlp_fwd_launch(const float* X, int64_t T, int32_t E, int32_t K, const float* const* A,
                                 const float* const* a, const float* const* B, const float* const* b, const int32_t* h,
                                 const int32_t* O, float* const* Y, void* workspace, hipStream_t s, bool packed) {
  X3Plan p;
  if (!x3_make_plan(E, K, h, O, &p)) return hipErrorInvalidValue;
  X3Args args;
  x3_fill_args(p, X, T, E, K, A, a, B, b, h, O, Y, workspace, &args);
  hipError_t e = hipSuccess;
  if (!packed) {
    hipLaunchKernelGGL(x3_pack_k, dim3(p.U), dim3(256), 0, s, args);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
  }
  const int64_t tiles = (T + 31) / 32;
  const int64_t blocks_needed = (tiles + 3) / 4;  // four waves, one tile each
  const int grid = (int)(blocks_needed < 4096 ? blocks_needed : 4096);
  const size_t lds = 2 * (size_t)kImgBytes + 4 * (size_t)kScr * sizeof(float);
  if (lds > 48 * 1024) {
    e = hipFuncSetAttribute((const void*)x3_fwd_k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(x3_fwd_k, dim3(grid), dim3(256), lds, s, args);
  if (e != hipSuccess) return e;
  return hipGetLastError();
}
