// mfmalab.hip — what does v_mfma_f32_32x32x2_f32 actually sustain on this box?
//   hipcc -O3 --offload-arch=gfx950 profiles/mfmalab.hip -o gpurun_out/mfmalab && gpurun_out/mfmalab
// Variants: CH = independent accumulator chains per wave (1: every MFMA depends on the previous one),
// waves per SIMD via block count/size; optional VALU filler between MFMAs (the GELU of mlp_fwd.hip).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int CH, int VALU>
__global__ void __launch_bounds__(256) mfma_k(float* out, int iters, float seed) {
  f32x16 acc[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = seed * (c + r);
  float a = seed + threadIdx.x, b = seed * 2.f + threadIdx.x;
  float filler = seed;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
        if (VALU) {
#pragma unroll
          for (int v = 0; v < VALU; ++v) filler = fmaf(filler, 1.0001f, 0.5f);
        }
      }
    }
  }
  float s = filler;
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[c][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// bf16 matrix pipe: v_mfma_f32_32x32x16_bf16 (8 passes = 32 cycles per SIMD), same VALU filler
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
template <int CH, int VALU>
__global__ void __launch_bounds__(256) mfma_bf16_k(float* out, int iters, float seed) {
  f32x16 acc[CH];
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = seed * (c + r);
  bf16x8 a, b;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    a[i] = (__bf16)(seed + threadIdx.x + i);
    b[i] = (__bf16)(seed * 2.f + threadIdx.x - i);
  }
  float filler = seed;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
#pragma unroll
      for (int c = 0; c < CH; ++c) {
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
        if (VALU) {
#pragma unroll
          for (int v = 0; v < VALU; ++v) filler = fmaf(filler, 1.0001f, 0.5f);
        }
      }
    }
  }
  float s = filler;
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[c][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int CH, int VALU>
void run_bf16(const char* name, int blocks, int threads, int iters, float* out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((mfma_bf16_k<CH, VALU>), dim3(blocks), dim3(threads), 0, 0, out, iters, 0.001f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((mfma_bf16_k<CH, VALU>), dim3(blocks), dim3(threads), 0, 0, out, iters, 0.001f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double waves = (double)blocks * threads / 64;
  const double mfmas = waves * iters * 16.0 * CH;
  const double flops = mfmas * 32 * 32 * 16 * 2;
  const double per_simd = mfmas / 1024.0;
  printf("%-34s blocks=%5d thr=%4d  %8.3f ms  %7.1f TFLOP/s  %6.1f cyc/MFMA/SIMD @2.4GHz\n", name, blocks, threads, ms,
         flops / ms * 1e-9, ms * 1e-3 * 2.4e9 / per_simd);
}

template <int CH, int VALU>
void run(const char* name, int blocks, int threads, int iters, float* out) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((mfma_k<CH, VALU>), dim3(blocks), dim3(threads), 0, 0, out, iters, 0.001f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((mfma_k<CH, VALU>), dim3(blocks), dim3(threads), 0, 0, out, iters, 0.001f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double waves = (double)blocks * threads / 64;
  const double mfmas = waves * iters * 16.0 * CH;
  const double flops = mfmas * 32 * 32 * 2 * 2;
  // cycles per MFMA per SIMD at 2.4 GHz, assuming waves spread evenly over 1024 SIMDs
  const double per_simd = mfmas / 1024.0;
  printf("%-34s blocks=%5d thr=%4d  %8.3f ms  %7.1f TFLOP/s  %6.1f cyc/MFMA/SIMD @2.4GHz\n", name, blocks, threads, ms,
         flops / ms * 1e-9, ms * 1e-3 * 2.4e9 / per_simd);
}

int main() {
  float* out;
  hipMalloc(&out, 4096 * 1024 * sizeof(float));
  const int it = 2000;
  // one wave per SIMD: 256 CUs x 4 waves
  run<1, 0>("1 wave/SIMD dependent chain", 256, 256, it, out);
  run<2, 0>("1 wave/SIMD 2 chains", 256, 256, it, out);
  run<4, 0>("1 wave/SIMD 4 chains", 256, 256, it, out);
  run<1, 0>("2 waves/SIMD dependent chain", 512, 256, it, out);
  run<2, 0>("2 waves/SIMD 2 chains", 512, 256, it, out);
  run<1, 0>("4 waves/SIMD dependent chain", 1024, 256, it, out);
  run<1, 0>("8 waves/SIMD dependent chain", 2048, 256, it, out);
  run<1, 8>("1 wave/SIMD dep + 8 VALU", 256, 256, it, out);
  run<1, 14>("1 wave/SIMD dep + 14 VALU", 256, 256, it, out);
  run<1, 14>("2 waves/SIMD dep + 14 VALU", 512, 256, it, out);
  run<1, 24>("2 waves/SIMD dep + 24 VALU", 512, 256, it, out);
  run<1, 14>("4 waves/SIMD dep + 14 VALU", 1024, 256, it, out);
  printf("-- bf16 32x32x16 --\n");
  run_bf16<1, 0>("bf16 1 wave/SIMD dependent", 256, 256, it, out);
  run_bf16<2, 0>("bf16 1 wave/SIMD 2 chains", 256, 256, it, out);
  run_bf16<1, 0>("bf16 2 waves/SIMD dependent", 512, 256, it, out);
  run_bf16<2, 0>("bf16 2 waves/SIMD 2 chains", 512, 256, it, out);
  run_bf16<1, 4>("bf16 1 wave/SIMD dep + 4 VALU", 256, 256, it, out);
  run_bf16<1, 8>("bf16 1 wave/SIMD dep + 8 VALU", 256, 256, it, out);
  run_bf16<1, 8>("bf16 2 waves/SIMD dep + 8 VALU", 512, 256, it, out);
  run_bf16<1, 16>("bf16 2 waves/SIMD dep + 16 VALU", 512, 256, it, out);
  run_bf16<2, 16>("bf16 2 waves/SIMD 2ch + 16 VALU", 512, 256, it, out);
  run_bf16<1, 16>("bf16 4 waves/SIMD dep + 16 VALU", 1024, 256, it, out);
  hipFree(out);
  return 0;
}
