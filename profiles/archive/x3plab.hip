// x3plab.hip — phase timing of mlp_bwd_x3p_k (csrc/mlp_bwd.hip built with PSF_X3P_TRACE): shader-clock timestamps that
// every wave of one workgroup takes at the phase boundaries of one unit, at the Temporal-Order training shape.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -o /tmp/x3plab profiles/x3plab.hip && /tmp/x3plab
#ifndef PSF_X3P_NOTRACE  // -DPSF_X3P_NOTRACE: timing only (the stamps themselves cost time), plus a checksum of the results
#define PSF_X3P_TRACE 1
#endif
#include <atomic>
#include <cstdio>
#include <vector>

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
  const int o_links = argc > 2 ? atoi(argv[2]) : 15;  // outputs of MLPs 1..K-1 (> 16: the kernels' wide-dY instances)
  std::vector<int32_t> h(K, 32), O(K, o_links);
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
  for (int rep = 0; rep < 4; ++rep) {
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
    std::printf("mlp_bwd_x3p_k: %.3f ms per call\n", ms / 10);
#ifdef PSF_PS_GUARD
    {
      unsigned int stuck = 0;
      CK(hipMemcpyFromSymbol(&stuck, HIP_SYMBOL(psf_ps_stuck), sizeof(stuck)));
      if (stuck) std::printf("STUCK: a counter wait gave up (code %u)\n", stuck);
    }
#endif
    {  // checksums of the results of the last call: equal across builds whose arithmetic is meant to be the same
      auto crc = [&](const float* d, size_t n) {
        std::vector<uint32_t> v(n);
        (void)hipMemcpy(v.data(), d, n * 4, hipMemcpyDeviceToHost);
        uint64_t hsum = 1469598103934665603ull;
        for (uint32_t x : v) hsum = (hsum ^ x) * 1099511628211ull;
        return hsum;
      };
      uint64_t hA = 0, ha = 0, hB = 0, hb = 0;
      for (int k = 0; k < K; ++k) hA ^= crc(dA[k], 1024) * (2 * k + 3), ha ^= crc(da[k], 32) * (2 * k + 3), hB ^= crc(dB[k], (size_t)O[k] * 32) * (2 * k + 3), hb ^= crc(db[k], O[k]) * (2 * k + 3);
      std::printf("checksums dX %016llx dA %016llx da %016llx dB %016llx db %016llx\n", (unsigned long long)crc(dX, (size_t)T * E),
                  (unsigned long long)hA, (unsigned long long)ha, (unsigned long long)hB, (unsigned long long)hb);
      std::vector<float> hdb(O[1]);
      (void)hipMemcpy(hdb.data(), db[1], O[1] * 4, hipMemcpyDeviceToHost);
      std::printf("db[1][0..3] = %.9g %.9g %.9g %.9g\n", hdb[0], hdb[1], hdb[2], hdb[3]);
    }
  }
#ifdef PSF_X3P_TRACE
  unsigned long long tr[8][32];
  CK(hipMemcpyFromSymbol(tr, HIP_SYMBOL(psf_x3p_trace), sizeof(tr)));
  const char* names[25] = {"unit top", "after B0", "t0 start", "t0 dY split+prefetch issued", "t0 steps 1,2 issued", "t0 GELU done",
                           "t0 step 4 issued", "", "", "", "t1 start", "t1 dY split+prefetch issued", "t1 steps 1,2 issued",
                           "t1 GELU done", "t1 step 4 issued", "", "", "", "tiles done (steps 5,6 issued)", "half sums + SCR writes",
                           "after B1", "sum8 + stores dA", "after B2", "after B3 (dBT written)", "sum8 + stores dBT"};
  std::printf("%-34s", "slot (clocks since wave 0's unit top)");
  for (int w = 0; w < 8; ++w) std::printf("   wave%d", w);
  std::printf("\n");
#ifdef PSF_LAB_ROLES  // the last traced launch was mlp_bwd_ps_k: waves 0-3 producers, 4-7 consumers
  const char* pn[7] = {"top", "next unit's image there", "A1: GELU, split y || step 1 of next", "dY handed over",
                       "Hpost handed over", "A2: split dY', split g || step 2 of next", "S slot free again (for G)"};
  const char* cn[21] = {"t0 top", "t0 dY ready", "t0 ys ready, read", "t0 step 4 issued", "t0 gs ready", "", "", "", "t1 top",
                        "t1 dY ready", "t1 ys ready, read", "t1 step 4 issued", "t1 gs ready", "", "", "", "tiles done (4)",
                        "sums in LDS", "all arrived", "sums stored", "all done"};
  std::printf("producers (clocks since wave 0's tile-unit top)  wave0   wave1   wave2   wave3\n");
  for (int t = 0; t < 2; ++t)
    for (int sidx = 0; sidx < 7; ++sidx) {
      std::printf("t%d %-39s", t, pn[sidx]);
      for (int w = 0; w < 4; ++w) std::printf(" %7lld", (long long)(tr[w][8 * t + sidx] - tr[0][0]));
      std::printf("\n");
    }
  {
    const char* an[3] = {"t1 A2: B^T fragments read", "t1 A2: dY' split", "t1 A2: next dY loads issued"};
    for (int sidx = 0; sidx < 3; ++sidx) {
      std::printf("%-42s", an[sidx]);
      for (int w = 0; w < 4; ++w) std::printf(" %7lld", (long long)(tr[w][16 + sidx] - tr[0][0]));
      std::printf("\n");
    }
  }
  std::printf("consumers                                       wave4   wave5   wave6   wave7\n");
  std::printf("%-42s", "unit top");
  for (int w = 4; w < 8; ++w) std::printf(" %7lld", (long long)(tr[w][30] - tr[0][0]));
  std::printf("\n");
  for (int sidx = 0; sidx < 21; ++sidx) {
    if (!cn[sidx][0]) continue;
    std::printf("%-42s", cn[sidx]);
    for (int w = 4; w < 8; ++w) std::printf(" %7lld", (long long)(tr[w][sidx] - tr[0][0]));
    std::printf("\n");
  }
#else
  for (int sidx = 0; sidx < 25; ++sidx) {
    if (!names[sidx][0]) continue;
    std::printf("%-34s", names[sidx]);
    for (int w = 0; w < 8; ++w) std::printf(" %7lld", (long long)(tr[w][sidx] - tr[0][0]));
    std::printf("\n");
  }
#endif
#endif
  return 0;
}
