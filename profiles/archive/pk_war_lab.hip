// pk_war_lab.hip — does a 32-bit VALU write to the LOW register of a pair, issued right behind a packed-f32 instruction that
// READS that pair, reach the packed instruction's last lanes? Round 4: the mixer step kernel's accumulate phase came out
// wrong in dword 0 / dword 2 of lanes 48..63 of one wave, sporadically and only when hipcc had SLP-packed its multiply-adds
// into v_pk_mul_f32 / v_pk_add_f32; its ISA has   v_pk_add_f32 v[12:13], v[12:13], v[20:21] ; v_mov_b32 v20, v19   (the
// write-after-read this lab isolates).
//     hipcc -O3 --offload-arch=gfx950 -o /tmp/pk_war_lab profiles/pk_war_lab.hip && /tmp/pk_war_lab
//
// Every wave repeats ITER times, all in one asm statement on fixed registers:
//     pair = (1.0, 1.0) ; [an LDS read + s_waitcnt: the wave stalls and resumes] ; v_pk_add_f32 acc, acc, pair ;
//     NOPS x s_nop 0 ; v_mov_b32 pair.lo, 1000.0
// acc must end as (ITER, ITER) in every lane; a lane whose packed add saw 1000.0 ends high in the low half.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define STR2(x) #x
#define STR(x) STR2(x)

#define LAB_BODY(PAD)                                                                                          \
  asm volatile(                                                                                                \
      "v_mov_b32 v20, 0\n\tv_mov_b32 v21, 0\n\t"                                                               \
      "s_mov_b32 s20, %3\n\t"                                                                                  \
      "1:\n\t"                                                                                                 \
      "v_mov_b32 v22, 1.0\n\tv_mov_b32 v23, 1.0\n\t"                                                           \
      "ds_read_b32 v24, %2\n\t"                                                                                \
      "s_waitcnt lgkmcnt(0)\n\t"                                                                               \
      "v_pk_mul_f32 v[22:23], v[22:23], v[24:25] op_sel_hi:[1,0]\n\t"                                          \
      "s_nop 0\n\t"                                                                                            \
      "v_pk_add_f32 v[20:21], v[20:21], v[22:23]\n\t" PAD "v_mov_b32 v22, 0x447a0000\n\t"                      \
      "s_sub_u32 s20, s20, 1\n\t"                                                                              \
      "s_cmp_lg_u32 s20, 0\n\t"                                                                                \
      "s_cbranch_scc1 1b\n\t"                                                                                  \
      "s_nop 4\n\t"                                                                                            \
      "v_mov_b32 %0, v20\n\tv_mov_b32 %1, v21\n\t"                                                             \
      : "=v"(r0), "=v"(r1)                                                                                     \
      : "v"(lds_addr), "s"(iters)                                                                              \
      : "v20", "v21", "v22", "v23", "v24", "v25", "s20", "scc", "memory")

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;

// 512 threads: waves 0-3 run the packed-add loop, waves 4-7 (their partners on the four SIMDs) run back-to-back bf16 MFMAs
// when MFMA is set — in the kernel the packed arithmetic of one workgroup runs beside the matrix phase of the workgroup
// that shares its CU, which no other kernel of this library does.
template <int NOPS, bool MFMA>
__global__ void __launch_bounds__(512) lab(float* out, int iters, float* sink) {
  __shared__ float lds[512];
  lds[threadIdx.x] = 1.0f;  // the multiplier read back from LDS: pair * 1.0 = pair
  __syncthreads();
  if (threadIdx.x >= 256) {  // wave-uniform
    if (MFMA) {
      bf16x8 a, b;
      for (int i = 0; i < 8; ++i) a[i] = b[i] = (__bf16)(0.001f * (threadIdx.x & 7) + i);
      f32x16 c0 = {0}, c1 = {0};
      for (int it = 0; it < iters * 2; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b, a, c1, 0, 0, 0);
      }
      if (c0[0] + c1[3] == 12345.f) sink[0] = c0[1];
    }
    return;
  }
  const unsigned lds_addr = (unsigned)(size_t)(&lds[threadIdx.x]);
  float r0, r1;
  if constexpr (NOPS == 0) LAB_BODY("");
  if constexpr (NOPS == 1) LAB_BODY("s_nop 0\n\t");
  if constexpr (NOPS == 2) LAB_BODY("s_nop 1\n\t");
  out[(size_t)blockIdx.x * 512 + threadIdx.x * 2] = r0;
  out[(size_t)blockIdx.x * 512 + threadIdx.x * 2 + 1] = r1;
}

// a second stream of plain memory traffic, so that waves of the lab stall and resume at varying moments
__global__ void stream_k(float4* a, const float4* b, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = b[i];
}

template <int NOPS, bool MFMA>
void run(int blocks, int iters, float* d_out, bool with_stream, hipStream_t s2, float4* sa, float4* sb, size_t sn) {
  std::vector<float> h((size_t)blocks * 512);
  long bad_lo = 0, bad_hi = 0;
  long by_lane[64] = {0};
  for (int rep = 0; rep < 20; ++rep) {
    if (with_stream) hipLaunchKernelGGL(stream_k, dim3(1024), dim3(256), 0, s2, sa, sb, sn);
    hipLaunchKernelGGL((lab<NOPS, MFMA>), dim3(blocks), dim3(512), 0, 0, d_out, iters, (float*)sa);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d_out, h.size() * sizeof(float), hipMemcpyDeviceToHost);
    for (size_t i = 0; i < h.size(); i += 2) {
      if (h[i] != (float)iters) {
        ++bad_lo;
        ++by_lane[(i / 2) & 63];
      }
      if (h[i + 1] != (float)iters) ++bad_hi;
    }
  }
  printf("s_nop between: %d | MFMA partner waves: %d | memory stream beside: %d | lanes with a wrong LOW half %ld, HIGH half %ld of %zu | by lane:", NOPS,
         (int)MFMA, (int)with_stream, bad_lo, bad_hi, (size_t)20 * blocks * 256);
  for (int l = 0; l < 64; ++l) printf(" %ld", by_lane[l]);
  printf("\n");
  fflush(stdout);
}

int main() {
  const int blocks = 1024, iters = 4096;
  float* d_out;
  hipMalloc(&d_out, (size_t)blocks * 512 * sizeof(float));
  const size_t sn = (size_t)16 << 20;  // 256 MiB per buffer
  float4 *sa, *sb;
  hipMalloc(&sa, sn * sizeof(float4));
  hipMalloc(&sb, sn * sizeof(float4));
  hipMemset(sb, 0, sn * sizeof(float4));
  hipStream_t s2;
  hipStreamCreate(&s2);
  for (int ws = 0; ws < 2; ++ws) {
    run<0, false>(blocks, iters, d_out, ws, s2, sa, sb, sn);
    run<0, true>(blocks, iters, d_out, ws, s2, sa, sb, sn);
    run<1, true>(blocks, iters, d_out, ws, s2, sa, sb, sn);
    run<2, true>(blocks, iters, d_out, ws, s2, sa, sb, sn);
  }
  return 0;
}
