// x3flab.hip — phase timing of x3_fwd_k (csrc/mlp_fwd_x3.hip built with PSF_X3F_TRACE): shader-clock timestamps that
// every wave of one workgroup takes at the phase boundaries of one unit, 15 MLPs (E = h = 32) over T tokens.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o /tmp/x3flab profiles/x3flab.hip && /tmp/x3flab
#ifndef PSF_X3F_NOTRACE  // -DPSF_X3F_NOTRACE: timing only, three readings of 30 calls after a warm-up
#define PSF_X3F_TRACE 1
#endif
#include <cstdio>
#include <vector>

#include <atomic>
#include "../sparsefactorization_amd/csrc/mlp_fwd_x3.hip"

#define CK(x)                                                                     \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      std::printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__);  \
      return 2;                                                                   \
    }                                                                             \
  } while (0)

int main(int argc, char** argv) {
  const int64_t T = argc > 1 ? atoll(argv[1]) : 40 * 16384;
  const int E = 32, K = 15;
  std::vector<int32_t> h(K, 32), O(K, 15);
  O[0] = 8;
  std::vector<float*> A(K), a(K), B(K), b(K), Y(K);
  auto dev_rand = [&](size_t n, float scale) {
    std::vector<float> v(n);
    uint32_t st = 12345u + (uint32_t)n;
    for (auto& x : v) {
      st = st * 1664525u + 1013904223u;
      x = scale * ((float)(st >> 8) / 8388608.f - 1.f);
    }
    float* d = nullptr;
    if (hipMalloc(&d, n * 4) != hipSuccess) return (float*)nullptr;
    (void)hipMemcpy(d, v.data(), n * 4, hipMemcpyHostToDevice);
    return d;
  };
  float* X = dev_rand((size_t)T * E, 1.f);
  for (int k = 0; k < K; ++k) {
    A[k] = dev_rand(32 * 32, 0.2f);
    a[k] = dev_rand(32, 0.1f);
    B[k] = dev_rand((size_t)O[k] * 32, 0.2f);
    b[k] = dev_rand(32, 0.1f);
    Y[k] = dev_rand((size_t)T * O[k], 0.f);
  }
  void* ws = nullptr;
  CK(hipMalloc(&ws, psf_x3_mlp_fwd_workspace(E, K, h.data(), O.data())));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
#ifdef PSF_X3F_NOTRACE
  for (int it = 0; it < 100; ++it)
    CK(psf_x3_mlp_fwd_launch(X, T, E, K, A.data(), a.data(), B.data(), b.data(), h.data(), O.data(), Y.data(), ws, nullptr));
  for (int rd = 0; rd < 3; ++rd) {
    CK(hipEventRecord(e0));
    for (int it = 0; it < 30; ++it)
      CK(psf_x3_mlp_fwd_launch(X, T, E, K, A.data(), a.data(), B.data(), b.data(), h.data(), O.data(), Y.data(), ws, nullptr));
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float msr = 0;
    CK(hipEventElapsedTime(&msr, e0, e1));
    std::printf("x3 forward, T = %lld: %.4f ms per call\n", (long long)T, msr / 30);
  }
  return 0;
#else
  for (int it = 0; it < 3; ++it)
    CK(psf_x3_mlp_fwd_launch(X, T, E, K, A.data(), a.data(), B.data(), b.data(), h.data(), O.data(), Y.data(), ws, nullptr));
  CK(hipEventRecord(e0));
  for (int it = 0; it < 10; ++it)
    CK(psf_x3_mlp_fwd_launch(X, T, E, K, A.data(), a.data(), B.data(), b.data(), h.data(), O.data(), Y.data(), ws, nullptr));
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::printf("x3 forward, T = %lld: %.3f ms per call (with the trace stores)\n", (long long)T, ms / 10);
  unsigned long long tr[4][16];
  CK(hipMemcpyFromSymbol(tr, HIP_SYMBOL(psf_x3f_trace), sizeof(tr)));
  const char* names[7] = {"unit top", "after barrier", "Y stores of the previous MLP issued", "weight fragments read",
                          "two tiles computed", "Y^T parked", "image DMA issued"};
  const int order[7] = {0, 1, 6, 2, 3, 4, 5};
  std::printf("%-44s", "slot (clocks since wave 0's unit top)");
  for (int w = 0; w < 4; ++w) std::printf("   wave%d", w);
  std::printf("\n");
  for (int oi = 0; oi < 7; ++oi) {
    const int sidx = order[oi];
    std::printf("%-44s", names[sidx]);
    for (int w = 0; w < 4; ++w) std::printf(" %7lld", (long long)(tr[w][sidx] - tr[0][0]));
    std::printf("\n");
  }
  return 0;
#endif
}
