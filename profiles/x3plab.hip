// x3plab.hip — phase timing of mlp_bwd_x3p_k (csrc/mlp_bwd.hip built with PSF_X3P_TRACE): shader-clock timestamps that
// every wave of one workgroup takes at the phase boundaries of one unit, at the Temporal-Order training shape.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -o /tmp/x3plab profiles/x3plab.hip && /tmp/x3plab
#define PSF_X3P_TRACE 1
#include <atomic>
#include <cstdio>
#include <vector>

std::atomic<int> psf_g_mlp_bwd_variant{3};
extern "C" int psf_internal_fail(int code, const char* message) {
  std::printf("psf_internal_fail(%d): %s\n", code, message);
  return code;
}
#include "../sparsefactorization_amd/csrc/mlp_bwd.hip"

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
  std::vector<float*> A(K), a(K), B(K), dY(K), dA(K), da(K), dB(K), db(K);
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
  float* dX = dev_rand((size_t)T * E, 0.f);
  for (int k = 0; k < K; ++k) {
    A[k] = dev_rand(32 * 32, 0.2f);
    a[k] = dev_rand(32, 0.1f);
    B[k] = dev_rand((size_t)O[k] * 32, 0.2f);
    dY[k] = dev_rand((size_t)T * O[k], 1.f);
    dA[k] = dev_rand(32 * 32, 0.f);
    da[k] = dev_rand(32, 0.f);
    dB[k] = dev_rand((size_t)O[k] * 32, 0.f);
    db[k] = dev_rand(32, 0.f);
  }
  const int64_t ws_bytes = psf_mlp_bwd_workspace(T, E, K, h.data(), O.data());
  void* ws = nullptr;
  CK(hipMalloc(&ws, ws_bytes));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int variant : {3, 2, 3}) {
    psf_g_mlp_bwd_variant.store(variant);
    for (int it = 0; it < 3; ++it)
      if (psf_mlp_bwd_f32(X, T, E, K, A.data(), a.data(), B.data(), h.data(), O.data(), dY.data(), dX, dA.data(), da.data(),
                          dB.data(), db.data(), ws, ws_bytes, nullptr) != 0)
        return 3;
    CK(hipEventRecord(e0));
    for (int it = 0; it < 10; ++it)
      psf_mlp_bwd_f32(X, T, E, K, A.data(), a.data(), B.data(), h.data(), O.data(), dY.data(), dX, dA.data(), da.data(), dB.data(),
                      db.data(), ws, ws_bytes, nullptr);
    CK(hipEventRecord(e1));
    CK(hipDeviceSynchronize());
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::printf("variant %d: %.3f ms per call\n", variant, ms / 10);
  }
  unsigned long long tr[8][32];
  CK(hipMemcpyFromSymbol(tr, HIP_SYMBOL(psf_x3p_trace), sizeof(tr)));
  const char* names[25] = {"unit top", "after B0", "t0 start", "t0 dY split+prefetch issued", "t0 steps 1,2 issued", "t0 GELU done",
                           "t0 step 4 issued", "", "", "", "t1 start", "t1 dY split+prefetch issued", "t1 steps 1,2 issued",
                           "t1 GELU done", "t1 step 4 issued", "", "", "", "tiles done (steps 5,6 issued)", "half sums + SCR writes",
                           "after B1", "sum8 + stores dA", "after B2", "after B3 (dBT written)", "sum8 + stores dBT"};
  std::printf("%-34s", "slot (clocks since wave 0's unit top)");
  for (int w = 0; w < 8; ++w) std::printf("   wave%d", w);
  std::printf("\n");
  for (int sidx = 0; sidx < 25; ++sidx) {
    if (!names[sidx][0]) continue;
    std::printf("%-34s", names[sidx]);
    for (int w = 0; w < 8; ++w) std::printf(" %7lld", (long long)(tr[w][sidx] - tr[0][0]));
    std::printf("\n");
  }
  return 0;
}
