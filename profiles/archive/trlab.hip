// trlab.hip — checks csrc/mlp_planes.h (the dual-use LDS planes of the split-bf16 MLP backward) with exact data, one wave.
//   hipcc --offload-arch=gfx950 -O3 -o gpurun_out/trlab profiles/trlab.hip && gpurun_out/trlab
// 1. a matrix M[32][32] of distinct 16-bit patterns written row-wise by chunks -> row_frag, tr_frag, tr_frag_acc
// 2. a tile T[32][32] of f32 held in ACCUMULATOR layout -> split16 + store_acc_plane (3 terms) -> tr_frag; acc_frag
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../sparsefactorization_amd/csrc/mlp_planes.h"

using namespace psf_x3;

__global__ void __launch_bounds__(64) lab_k(const uint16_t* M, uint16_t* out_row, uint16_t* out_tr, uint16_t* out_tracc,
                                            const float* T, float* out_t_tr, float* out_t_acc) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[4 * kPlaneBytes];
  const int lane = threadIdx.x, c = lane & 31, half = lane >> 5;
  const PlaneLane L = plane_lane(lane);
  // 1.
  for (int s = 0; s < 2; ++s) {
    uint16_t v[8];
    for (int i = 0; i < 8; ++i) v[i] = M[c * 32 + 16 * s + 8 * half + i];
    *reinterpret_cast<uint4*>(lds + L.row[s]) = *reinterpret_cast<const uint4*>(v);
  }
  asm volatile("" ::: "memory");
  for (int s = 0; s < 2; ++s) {
    const bf16x8 r = row_frag(lds, L, s), t = tr_frag(lds, L, s), ta = tr_frag_acc(lds, L, s);
    *reinterpret_cast<bf16x8*>(out_row + (s * 64 + lane) * 8) = r;
    *reinterpret_cast<bf16x8*>(out_tr + (s * 64 + lane) * 8) = t;
    *reinterpret_cast<bf16x8*>(out_tracc + (s * 64 + lane) * 8) = ta;
  }
  // 2.
  float v[16];
  for (int r = 0; r < 16; ++r) v[r] = T[cd_row(r, half) * 32 + c];
  const Split16 x = split16(v);
  for (int t = 0; t < 3; ++t) store_acc_plane(lds + (1 + t) * kPlaneBytes, L, x, t);
  asm volatile("" ::: "memory");
  for (int s = 0; s < 2; ++s) {
    float sum[8] = {0, 0, 0, 0, 0, 0, 0, 0}, sum_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int t = 2; t >= 0; --t) {
      const bf16x8 f = tr_frag(lds + (1 + t) * kPlaneBytes, L, s), fa = acc_frag(x, t, s);
      for (int i = 0; i < 8; ++i) {
        sum[i] += (float)f[i];
        sum_acc[i] += (float)fa[i];
      }
    }
    for (int i = 0; i < 8; ++i) {
      out_t_tr[(s * 64 + lane) * 8 + i] = sum[i];
      out_t_acc[(s * 64 + lane) * 8 + i] = sum_acc[i];
    }
  }
}

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      std::printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); \
      return 2;                                                                 \
    }                                                                           \
  } while (0)

int main() {
  std::vector<uint16_t> M(1024);
  std::vector<float> T(1024);
  for (int i = 0; i < 1024; ++i) {
    M[i] = (uint16_t)(0x1000 + i * 7);
    T[i] = (float)(i * 37 % 1024) + (float)(i % 13) / 4096.f + 1.0f / 3.0f * (float)(i % 5);  // needs all three terms
  }
  uint16_t *dM, *d_row, *d_tr, *d_tracc;
  float *dT, *d_ttr, *d_tacc;
  CK(hipMalloc(&dM, 2048));
  CK(hipMalloc(&d_row, 2048));
  CK(hipMalloc(&d_tr, 2048));
  CK(hipMalloc(&d_tracc, 2048));
  CK(hipMalloc(&dT, 4096));
  CK(hipMalloc(&d_ttr, 4096));
  CK(hipMalloc(&d_tacc, 4096));
  CK(hipMemcpy(dM, M.data(), 2048, hipMemcpyHostToDevice));
  CK(hipMemcpy(dT, T.data(), 4096, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(lab_k, dim3(1), dim3(64), 0, 0, dM, d_row, d_tr, d_tracc, dT, d_ttr, d_tacc);
  CK(hipDeviceSynchronize());
  std::vector<uint16_t> o_row(1024), o_tr(1024), o_tracc(1024);
  std::vector<float> o_ttr(1024), o_tacc(1024);
  CK(hipMemcpy(o_row.data(), d_row, 2048, hipMemcpyDeviceToHost));
  CK(hipMemcpy(o_tr.data(), d_tr, 2048, hipMemcpyDeviceToHost));
  CK(hipMemcpy(o_tracc.data(), d_tracc, 2048, hipMemcpyDeviceToHost));
  CK(hipMemcpy(o_ttr.data(), d_ttr, 4096, hipMemcpyDeviceToHost));
  CK(hipMemcpy(o_tacc.data(), d_tacc, 4096, hipMemcpyDeviceToHost));
  int bad[5] = {0, 0, 0, 0, 0};
  for (int s = 0; s < 2; ++s)
    for (int lane = 0; lane < 64; ++lane)
      for (int i = 0; i < 8; ++i) {
        const int c = lane & 31, half = lane >> 5, k = 16 * s + 8 * half + i;
        const int rho = (i & 3) + 16 * s + 8 * (i >> 2) + 4 * half;
        const int at = (s * 64 + lane) * 8 + i;
        bad[0] += o_row[at] != M[c * 32 + k];
        bad[1] += o_tr[at] != M[k * 32 + c];
        bad[2] += o_tracc[at] != M[rho * 32 + c];
        // the tile is stored transposed (plane[c][row]), so tr_frag gives T[row = c][col = k]
        bad[3] += o_ttr[at] != T[c * 32 + k];
        bad[4] += o_tacc[at] != T[rho * 32 + c];
        if (bad[3] == 1 && o_ttr[at] != T[c * 32 + k]) std::printf("first t_tr mismatch s=%d lane=%d i=%d got %.9g want %.9g\n", s, lane, i, o_ttr[at], T[c * 32 + k]);
      }
  const char* names[5] = {"row_frag", "tr_frag", "tr_frag_acc", "acc store -> tr_frag (3-term sum)", "acc_frag (3-term sum)"};
  int total = 0;
  for (int i = 0; i < 5; ++i) {
    std::printf("%-40s %s (%d mismatches of 1024)\n", names[i], bad[i] ? "FAIL" : "ok", bad[i]);
    total += bad[i];
  }
  std::printf(total ? "TRLAB FAIL\n" : "TRLAB PASS\n");
  return total ? 1 : 0;
}
