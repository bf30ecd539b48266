// widelab.hip — lab for csrc/x3_gemm.h: the split-bf16 GEMM of the wide producer MLPs, checked against float64 on the
// host (sampled entries) and timed at the ListOps sizes (E = 512, 12 x h = 128, T = 64000).
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off -fno-slp-vectorize profiles/widelab.hip -o profiles/bin/widelab
//   profiles/bin/widelab [T] [E] [J] [reps]
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../sparsefactorization_amd/csrc/x3_gemm.h"

using namespace psf_wide;

#define CK(x)                                                                       \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      fprintf(stderr, "%s:%d %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      exit(2);                                                                      \
    }                                                                               \
  } while (0)

static int64_t up(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

struct Planes {
  unsigned char* base = nullptr;
  Operand op{};
  int64_t plane_bytes = 0;
};

static Planes make_planes(const float* dsrc, int64_t rows, int64_t cols, int64_t ld, hipStream_t s) {
  Planes P;
  const int64_t rows_pad = up(rows, 256);
  P.plane_bytes = (cols / 16) * rows_pad * 32;
  CK(hipMalloc(&P.base, 3 * P.plane_bytes));
  SplitArgs a;
  a.src = dsrc;
  for (int t = 0; t < 3; ++t) a.p[t] = P.base + t * P.plane_bytes, P.op.p[t] = a.p[t];
  a.rows = rows, a.rows_pad = rows_pad, a.ld = ld, a.blocks = (int)(cols / 16);
  P.op.rows_pad = rows_pad, P.op.blocks = a.blocks;
  dim3 grid((unsigned)(rows_pad / 64), (unsigned)std::min<int64_t>(8, (a.blocks + 3) / 4));
  hipLaunchKernelGGL(x3_split_planes_k, grid, dim3(256), 0, s, a);
  CK(hipGetLastError());
  return P;
}

int main(int argc, char** argv) {
  const int64_t T = argc > 1 ? atoll(argv[1]) : 64000;
  const int64_t E = argc > 2 ? atoll(argv[2]) : 512;
  const int64_t J = argc > 3 ? atoll(argv[3]) : 1536;
  const int reps = argc > 4 ? atoi(argv[4]) : 20;
  printf("widelab: T=%lld E=%lld J=%lld\n", (long long)T, (long long)E, (long long)J);
  hipStream_t s;
  CK(hipStreamCreate(&s));
  std::mt19937 rng(1234);
  std::normal_distribution<float> nd(0.f, 1.f);
  std::vector<float> hX((size_t)T * E), hW((size_t)J * E), hG((size_t)T * J);
  for (auto& v : hX) v = nd(rng);
  for (auto& v : hW) v = 0.05f * nd(rng);
  for (auto& v : hG) v = nd(rng) * 1e-3f;
  float *dX, *dW, *dG, *dWt;
  CK(hipMalloc(&dX, hX.size() * 4));
  CK(hipMalloc(&dW, hW.size() * 4));
  CK(hipMalloc(&dG, hG.size() * 4));
  CK(hipMalloc(&dWt, hW.size() * 4));
  CK(hipMemcpy(dX, hX.data(), hX.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dW, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dG, hG.data(), hG.size() * 4, hipMemcpyHostToDevice));
  std::vector<float> hWt((size_t)E * J);  // W^T [E][J]
  for (int64_t j = 0; j < J; ++j)
    for (int64_t e = 0; e < E; ++e) hWt[e * J + j] = hW[j * E + e];
  CK(hipMemcpy(dWt, hWt.data(), hWt.size() * 4, hipMemcpyHostToDevice));

  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto timeit = [&](auto fn, const char* what, double flop) {
    fn();
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < reps; ++i) fn();
    CK(hipEventRecord(e1, s));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    ms /= reps;
    printf("%-28s %8.3f ms  %7.1f TFLOP/s (f32-equivalent)  %7.1f TFLOP/s bf16\n", what, ms, flop / ms * 1e-9, 6 * flop / ms * 1e-9);
    return ms;
  };

  Planes XP, WP, GP, WTP;
  timeit([&] { if (XP.base) CK(hipFree(XP.base)); XP = make_planes(dX, T, E, E, s); }, "split X", 0);
  WP = make_planes(dW, J, E, E, s);     // rows j, cols e
  WTP = make_planes(dWt, E, J, J, s);   // rows e, cols j
  GP = make_planes(dG, T, J, J, s);     // rows tok, cols j
  CK(hipStreamSynchronize(s));
  const int64_t T_pad = up(T, 256), J_pad = up(J, 256), E_pad = up(E, 256);

  // ---------------- NT forward: Hpre^T[j][tok] = sum_e W[j][e] X[tok][e]  (row-major out for the check)
  {
    float* dOut;
    CK(hipMalloc(&dOut, (size_t)J_pad * T_pad * 4));
    GemmArgs g{};
    g.A = WP.op, g.B = XP.op;
    g.tiles_m = (int)(J_pad / 256), g.tiles_n = (int)(T_pad / 256), g.splits = 1, g.chunks = (int)(E / 16);
    g.n_fast = 0, g.epilogue = kEpiRowMajor, g.out = dOut, g.bias = nullptr, g.ld = T_pad, g.rows_alloc = J_pad;
    g.rows_valid = J, g.cols_valid = T;
    const unsigned grid = std::min(256u, (unsigned)(g.tiles_m * g.tiles_n));
    timeit([&] { hipLaunchKernelGGL(x3_gemm_k<false>, dim3(grid), dim3(512), 0, s, g); CK(hipGetLastError()); }, "NT fwd  (J x T, K = E)",
           2.0 * J * T * E);
    std::vector<float> hOut((size_t)J_pad * T_pad);
    CK(hipMemcpy(hOut.data(), dOut, hOut.size() * 4, hipMemcpyDeviceToHost));
    double maxerr = 0, maxref = 0;
    std::uniform_int_distribution<int64_t> rj(0, J - 1), rt(0, T - 1);
    for (int n = 0; n < 4000; ++n) {
      int64_t j = rj(rng), t = rt(rng);
      if (n < 8) j = n < 4 ? 0 : J - 1, t = (n & 1) ? T - 1 : 0;
      double ref = 0;
      for (int64_t e = 0; e < E; ++e) ref += (double)hW[j * E + e] * hX[t * E + e];
      maxerr = std::max(maxerr, std::fabs(ref - hOut[j * T_pad + t]));
      maxref = std::max(maxref, std::fabs(ref));
    }
    printf("  NT fwd check: max|err| %.3e  max|ref| %.3e  rel %.3e\n", maxerr, maxref, maxerr / maxref);
    // fragment epilogue with bias
    std::vector<float> hb(J_pad, 0.f);
    for (int64_t j = 0; j < J; ++j) hb[j] = 0.1f * nd(rng);
    float* dB;
    CK(hipMalloc(&dB, J_pad * 4));
    CK(hipMemcpy(dB, hb.data(), J_pad * 4, hipMemcpyHostToDevice));
    g.epilogue = kEpiFragBias, g.bias = dB, g.cols_valid = T_pad;
    timeit([&] { hipLaunchKernelGGL(x3_gemm_k<false>, dim3(grid), dim3(512), 0, s, g); CK(hipGetLastError()); }, "NT fwd, fragment epilogue",
           2.0 * J * T * E);
    CK(hipMemcpy(hOut.data(), dOut, hOut.size() * 4, hipMemcpyDeviceToHost));
    double e2 = 0;
    const int64_t units = J_pad / 32;
    for (int n = 0; n < 2000; ++n) {
      const int64_t j = rj(rng), t = rt(rng);
      double ref = hb[j];
      for (int64_t e = 0; e < E; ++e) ref += (double)hW[j * E + e] * hX[t * E + e];
      const int jr = (int)(j & 31), hf = (jr >> 2) & 1, r = (jr & 3) + 4 * (jr >> 3);
      const int64_t at = ((((t >> 5) * units + (j >> 5)) * 4 + (r >> 2)) * 64 + hf * 32 + (t & 31)) * 4 + (r & 3);
      e2 = std::max(e2, std::fabs(ref - hOut[at]));
    }
    printf("  fragment epilogue check: max|err| %.3e (rel %.3e)\n", e2, e2 / maxref);
    CK(hipFree(dOut));
    CK(hipFree(dB));
  }
  // ---------------- NT dX: dX[tok][e] = sum_j G[tok][j] W[j][e] = sum_j G[tok][j] Wt[e][j]
  {
    float* dOut;
    CK(hipMalloc(&dOut, (size_t)T * E * 4));
    GemmArgs g{};
    g.A = GP.op, g.B = WTP.op;
    g.tiles_m = (int)(T_pad / 256), g.tiles_n = (int)(E_pad / 256), g.splits = 1, g.chunks = (int)(J / 16);
    g.n_fast = 1, g.epilogue = kEpiRowMajor, g.out = dOut, g.ld = E, g.rows_alloc = T;
    g.rows_valid = T, g.cols_valid = E;
    const unsigned grid = std::min(256u, (unsigned)(g.tiles_m * g.tiles_n));
    timeit([&] { hipLaunchKernelGGL(x3_gemm_k<false>, dim3(grid), dim3(512), 0, s, g); CK(hipGetLastError()); }, "NT dX   (T x E, K = J)",
           2.0 * J * T * E);
    std::vector<float> hOut((size_t)T * E);
    CK(hipMemcpy(hOut.data(), dOut, hOut.size() * 4, hipMemcpyDeviceToHost));
    double maxerr = 0, maxref = 0;
    std::uniform_int_distribution<int64_t> re(0, E - 1), rt(0, T - 1);
    for (int n = 0; n < 3000; ++n) {
      const int64_t e = re(rng), t = n < 4 ? T - 1 : rt(rng);
      double ref = 0;
      for (int64_t j = 0; j < J; ++j) ref += (double)hG[t * J + j] * hW[j * E + e];
      maxerr = std::max(maxerr, std::fabs(ref - hOut[t * E + e]));
      maxref = std::max(maxref, std::fabs(ref));
    }
    printf("  NT dX check: max|err| %.3e  max|ref| %.3e  rel %.3e\n", maxerr, maxref, maxerr / maxref);
    CK(hipFree(dOut));
  }
  // ---------------- TN dA: dA[j][e] = sum_tok G[tok][j] X[tok][e]
  for (int splits : {8, 21, 42}) {
    float* dOut;
    CK(hipMalloc(&dOut, (size_t)splits * J_pad * E_pad * 4));
    GemmArgs g{};
    g.A = GP.op, g.B = XP.op;
    g.tiles_m = (int)(J_pad / 256), g.tiles_n = (int)(E_pad / 256), g.splits = splits, g.chunks = (int)(T_pad / 16);
    g.n_fast = 0, g.epilogue = kEpiRowMajor, g.out = dOut, g.ld = E_pad, g.rows_alloc = J_pad;
    g.rows_valid = J_pad, g.cols_valid = E_pad;
    const unsigned grid = std::min(256u, (unsigned)(g.tiles_m * g.tiles_n * splits));
    char name[64];
    snprintf(name, sizeof name, "TN dA   (J x E, K = T) /%d", splits);
    timeit([&] { hipLaunchKernelGGL(x3_gemm_k<true>, dim3(grid), dim3(512), 0, s, g); CK(hipGetLastError()); }, name, 2.0 * J * T * E);
    std::vector<float> hOut((size_t)splits * J_pad * E_pad);
    CK(hipMemcpy(hOut.data(), dOut, hOut.size() * 4, hipMemcpyDeviceToHost));
    double maxerr = 0, maxref = 0;
    std::uniform_int_distribution<int64_t> re(0, E - 1), rj(0, J - 1);
    for (int n = 0; n < 300; ++n) {
      const int64_t e = re(rng), j = n < 4 ? J - 1 : rj(rng);
      double ref = 0;
      for (int64_t t = 0; t < T; ++t) ref += (double)hG[t * J + j] * hX[t * E + e];
      double got = 0;
      for (int sp = 0; sp < splits; ++sp) got += hOut[((size_t)sp * J_pad + j) * E_pad + e];
      maxerr = std::max(maxerr, std::fabs(ref - got));
      maxref = std::max(maxref, std::fabs(ref));
    }
    printf("  TN dA check: max|err| %.3e  max|ref| %.3e  rel %.3e\n", maxerr, maxref, maxerr / maxref);
    CK(hipFree(dOut));
  }
  return 0;
}
