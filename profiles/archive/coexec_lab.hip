// coexec_lab.hip — do matrix-pipe time and vector time of DIFFERENT waves on one SIMD overlap, and for which vector instructions?
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -fno-slp-vectorize profiles/coexec_lab.hip -o profiles/bin/coexec_lab && profiles/bin/coexec_lab
// (round 5: the producer-MLP kernels show vector issue busy 58-61 %, matrix pipe busy 34 %, both at once 9-11 % — whatever
// the kernel's structure, profiles/r05a_mlp_arithmetic.md.)
//
// Every wave alternates a matrix phase (KM dependent v_mfma_f32_32x32x16_bf16) and a vector phase (KV instructions of one
// class on eight independent registers). PHASED: waves in an odd hardware wave slot of their SIMD run the vector phase first, so
// at any time half the SIMD's waves are in each phase; otherwise all waves run the phases in step. Reported: clocks per iteration per
// SIMD next to the phases run alone (sum = no overlap, max = full overlap).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

enum Cls { FMA = 0, AND_SUB = 1, PERM = 2, PKFMA = 3, TRANS = 4, GELU = 5, BFI = 6 };

template <int CLS>
__device__ __forceinline__ void vphase(float (&r)[8], int kv) {
  for (int i = 0; i < kv; i += 8) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if constexpr (CLS == FMA) r[j] = fmaf(r[j], 1.0001f, 0.5f);
      if constexpr (CLS == AND_SUB) {  // the split's pair: v_and + v_sub (counts as two)
        const float t = __uint_as_float(__float_as_uint(r[j]) & 0xffff0000u);
        r[j] = r[j] - t + 1.0f;
      }
      if constexpr (CLS == PERM) r[j] = __uint_as_float(__builtin_amdgcn_perm(__float_as_uint(r[j]), __float_as_uint(r[(j + 1) & 7]), 0x07060302u));
      if constexpr (CLS == PKFMA) {
        if (j & 1) continue;
        f32x2 v = {r[j], r[j + 1]};
        v = __builtin_elementwise_fma(v, f32x2{1.0001f, 1.0001f}, f32x2{0.5f, 0.5f});
        r[j] = v.x, r[j + 1] = v.y;
      }
      if constexpr (CLS == TRANS) r[j] = (j & 1) ? __builtin_amdgcn_exp2f(r[j]) : __builtin_amdgcn_rcpf(r[j]);
      if constexpr (CLS == BFI) r[j] = copysignf(r[j], r[(j + 3) & 7]);
      if constexpr (CLS == GELU) {  // the producers' GELU, one value (12 plain + 2 transcendental instructions)
        const float x = r[j];
        const float t = __builtin_amdgcn_rcpf(fmaf(fabsf(x), 0.2316419f, 1.0f));
        float p = fmaf(0.53070271f, t, -0.72657602f);
        p = fmaf(p, t, 0.71070687f);
        p = fmaf(p, t, -0.14224837f);
        p = fmaf(p, t, 0.12741479f);
        p = p * t;
        const float e = __builtin_amdgcn_exp2f((x * x) * -0.72134752f);
        r[j] = x * (0.5f + copysignf(0.5f - p * e, x));
      }
    }
  }
}

template <int CLS, bool PHASED, bool DO_M, bool DO_V>
__global__ void __launch_bounds__(256) lab_k(float* out, int iters, int km, int kv, float seed) {
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = seed * r;
  bf16x8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = (__bf16)(seed + threadIdx.x + i), b[i] = (__bf16)(seed * 2.f - i);
  float r[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = seed * (j + 1) + 0.25f;
  // the wave's slot on its SIMD (HW_REG_HW_ID bits 3:0) decides the phase order: co-resident waves of a SIMD sit in different
  // slots, whatever the dispatcher did with the workgroups
  const unsigned slot = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 4);
  const bool vfirst = PHASED && (slot & 1);
  for (int it = 0; it < iters; ++it) {
    if (DO_V && vfirst) vphase<CLS>(r, kv);
    if (DO_M) {
#ifdef PSF_ACC_VGPR  // accumulators in VGPRs, as in a kernel whose vector code reads them (the MLP kernels' GELU): the MFMA as
                     // inline assembly on "v" operands (timing only: the dependent chain relies on the hardware interlock)
      for (int m = 0; m < km; ++m) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
      asm volatile("s_nop 15\n\ts_nop 15");
#else
      for (int m = 0; m < km; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
#endif
    }
    if (DO_V && !vfirst) vphase<CLS>(r, kv);
    __builtin_amdgcn_sched_barrier(0);
  }
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < 16; ++q) s += acc[q];
#pragma unroll
  for (int j = 0; j < 8; ++j) s += r[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// The same work with the vector instructions placed INSIDE the wave's own MFMA chain: one MFMA, then kv / km vector
// instructions (a multiple of 8), every wave alike.
template <int CLS>
__global__ void __launch_bounds__(256) lab_inwave_k(float* out, int iters, int km, int kv, float seed) {
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = seed * r;
  bf16x8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = (__bf16)(seed + threadIdx.x + i), b[i] = (__bf16)(seed * 2.f - i);
  float r[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = seed * (j + 1) + 0.25f;
  const int per = kv / km;
  for (int it = 0; it < iters; ++it) {
    for (int m = 0; m < km; ++m) {
#ifdef PSF_ACC_VGPR
      asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
#else
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
#endif
      vphase<CLS>(r, per);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < 16; ++q) s += acc[q];
#pragma unroll
  for (int j = 0; j < 8; ++j) s += r[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CLS>
double run_inwave(int wps, int iters, int km, int kv, float* out) {
  const int blocks = 256 * wps;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((lab_inwave_k<CLS>), dim3(blocks), dim3(256), 0, 0, out, iters, km, kv, 0.001f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((lab_inwave_k<CLS>), dim3(blocks), dim3(256), 0, 0, out, iters, km, kv, 0.001f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e-3 * 2.4e9 / iters;
}

// What a producer-MLP wave does that the phases above do not: FLAGS bit 0 the vector phase starts from the accumulator (a data
// dependency on the MFMA chain), bit 1 every MFMA's operands come out of LDS (ds_read_b128 + wait), bit 2 a workgroup barrier
// per iteration. Scalar v_fma vector phase; waves in step.
template <int FLAGS, bool DO_M, bool DO_V>
__global__ void __launch_bounds__(256) lab_real_k(float* out, int iters, int km, int kv, float seed) {
  __shared__ bf16x8 sOp[2 * 256];
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = seed * r;
  bf16x8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = (__bf16)(seed + threadIdx.x + i), b[i] = (__bf16)(seed * 2.f - i);
  sOp[threadIdx.x] = a, sOp[256 + threadIdx.x] = b;
  __syncthreads();
  float r[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = seed * (j + 1) + 0.25f;
  for (int it = 0; it < iters; ++it) {
    if (DO_M)
      for (int m = 0; m < km; ++m) {
        if (FLAGS & 2) a = sOp[(threadIdx.x + m) & 255], b = sOp[256 + ((threadIdx.x + 2 * m) & 255)];
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
      }
    if (DO_V) {
      if (FLAGS & 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] += acc[j];
      }
      vphase<FMA>(r, kv);
      if ((FLAGS & 1) && DO_M) {
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j + 8] = r[j];  // ... and the next chain from the vector results
      }
    }
    if (FLAGS & 4) __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
  }
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < 16; ++q) s += acc[q];
#pragma unroll
  for (int j = 0; j < 8; ++j) s += r[j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int FLAGS, bool DO_M, bool DO_V>
double run_real(int wps, int iters, int km, int kv, float* out) {
  const int blocks = 256 * wps;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((lab_real_k<FLAGS, DO_M, DO_V>), dim3(blocks), dim3(256), 0, 0, out, iters, km, kv, 0.001f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((lab_real_k<FLAGS, DO_M, DO_V>), dim3(blocks), dim3(256), 0, 0, out, iters, km, kv, 0.001f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e-3 * 2.4e9 / iters;
}

template <int FLAGS>
void sweep_real(const char* name, int wps, int km, int kv, float* out) {
  const int iters = 400;
  const double m = run_real<FLAGS, true, false>(wps, iters, km, kv, out);
  const double v = run_real<FLAGS, false, true>(wps, iters, km, kv, out);
  const double both = run_real<FLAGS, true, true>(wps, iters, km, kv, out);
  printf("%-44s %d waves/SIMD  %2d MFMA + %3d v_fma:  matrix alone %7.0f  vector alone %7.0f  together %7.0f   (sum %7.0f, max %7.0f)\n",
         name, wps, km, kv, m, v, both, m + v, m > v ? m : v);
}

template <int CLS, bool PHASED, bool DO_M, bool DO_V>
double run(int wps, int iters, int km, int kv, float* out) {
  const int blocks = 256 * wps;  // 256-thread workgroups: one wave per SIMD each; blockIdx / 256 = which "layer" of waves
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((lab_k<CLS, PHASED, DO_M, DO_V>), dim3(blocks), dim3(256), 0, 0, out, iters, km, kv, 0.001f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((lab_k<CLS, PHASED, DO_M, DO_V>), dim3(blocks), dim3(256), 0, 0, out, iters, km, kv, 0.001f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms * 1e-3 * 2.4e9 / iters;  // clocks (at 2.4 GHz) per iteration of the whole SIMD (all its waves)
}

template <int CLS>
void sweep(const char* name, int wps, int km, int kv, float* out) {
  const int iters = 400;
  const double m = run<CLS, false, true, false>(wps, iters, km, kv, out);
  const double v = run<CLS, false, false, true>(wps, iters, km, kv, out);
  const double both = run<CLS, false, true, true>(wps, iters, km, kv, out);
  const double phased = run<CLS, true, true, true>(wps, iters, km, kv, out);
  const double inwave = (kv / km) % 8 == 0 ? run_inwave<CLS>(wps, iters, km, kv, out) : -1.0;
  printf("%-10s %d waves/SIMD  %2d MFMA + %3d vector per wave-iteration:  matrix alone %7.0f  vector alone %7.0f  in step %7.0f  "
         "half a period apart %7.0f  inside the chain %7.0f   (sum %7.0f, max %7.0f)\n",
         name, wps, km, kv, m, v, both, phased, inwave, m + v, m > v ? m : v);
}

int main(int argc, char** argv) {
  float* out;
  hipMalloc(&out, 4096 * 256 * sizeof(float));
  if (argc > 1) {  // "real": what a producer wave adds to the picture
    for (int wps : {2, 4}) {
      sweep_real<0>("independent phases", wps, 12, 192, out);
      sweep_real<1>("vector phase depends on the chain", wps, 12, 192, out);
      sweep_real<2>("MFMA operands from LDS", wps, 12, 192, out);
      sweep_real<4>("barrier per iteration", wps, 12, 192, out);
      sweep_real<3>("dependency + LDS operands", wps, 12, 192, out);
      sweep_real<7>("dependency + LDS operands + barrier", wps, 12, 192, out);
    }
    return 0;
  }
  for (int wps : {2, 4}) {
    sweep<FMA>("v_fma", wps, 12, 96, out);
    sweep<FMA>("v_fma", wps, 12, 192, out);
    sweep<AND_SUB>("and+sub", wps, 12, 96, out);
    sweep<PERM>("v_perm", wps, 12, 96, out);
    sweep<PKFMA>("v_pk_fma", wps, 12, 96, out);
    sweep<PKFMA>("v_pk_fma", wps, 12, 192, out);
    sweep<TRANS>("rcp/exp", wps, 12, 96, out);
    sweep<BFI>("v_bfi", wps, 12, 96, out);
    sweep<GELU>("gelu", wps, 12, 96, out);  // 8 values (112 instructions) behind every MFMA
  }
  return 0;
}
