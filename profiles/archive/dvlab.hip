// dvlab.hip — ablation / design lab for the dV step at cfg2 (N=16384, L=15, C=8, B=64). NOT product code.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off profiles/dvlab.hip -o profiles/bin/dvlab && profiles/bin/dvlab
// Times the shipped chord_dv_win_k next to variants, interleaved in one process; variants that keep the arithmetic are
// compared bit for bit with the shipped kernel's output.
//   FLAGS bit 0  CHORD    near offsets are compile-time constants (0,1,2,4,...) instead of kernarg values
//         bit 1  SIDE     far-link W elements come from a link-major side copy Wfar[b][f][n] (coalesced dword loads)
//         bit 2  SIDEDMA  same side copy, staged by LDS-DMA: one 16-byte-per-lane wave instruction per 256 rows and link
//         bit 3  CONSTW   ablation: far-link W elements are a constant (wrong results)
//         bit 4  ONEIMG   the two W tiles under the window are ONE flat LDS image (needs N*L % 4 == 0)
//         bit 5  NOFARZ   ablation: far dZ rows are a constant (wrong results)
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../sparsefactorization_amd/csrc/bwd_window.h"

using namespace psf;

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

constexpr int near_off(int k) { return k == 0 ? 0 : 1 << (k - 1); }

template <int L, int TGS, int R, int NT, int FLAGS>
__global__ void __launch_bounds__(NT)
lab_dv_k(const float* __restrict__ dZ, const float* __restrict__ W, const float* __restrict__ Wfar,
         float* __restrict__ dV, const Geom gm, const Offsets offs, const int64_t w_total) {
  using T = float;
  using Cfg = BwdWinCfg<T, L, TGS, R, NT>;
  constexpr int VEC = 4, TG = Cfg::TG, RS = Cfg::RS, TR = Cfg::TR, KN = Cfg::KN, NF = Cfg::NF;
  constexpr bool CHORD = FLAGS & 1, SIDE = FLAGS & 2, SIDEDMA = FLAGS & 4, CONSTW = FLAGS & 8, ONEIMG = FLAGS & 16,
                 NOFARZ = FLAGS & 32;
  using V4 = Vec<T, VEC>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  V4* __restrict__ sWin = reinterpret_cast<V4*>(smem);
  V4* __restrict__ sWpV = reinterpret_cast<V4*>(smem + Cfg::win_bytes);
  V4* __restrict__ sWcV = reinterpret_cast<V4*>(smem + Cfg::win_bytes + Cfg::w_tile_bytes);
  const T* __restrict__ sWpF = reinterpret_cast<const T*>(smem + Cfg::win_bytes);
  const T* __restrict__ sWcF = reinterpret_cast<const T*>(smem + Cfg::win_bytes + Cfg::w_tile_bytes);
  // side-copy image: NF columns of TR floats behind the two W tiles
  V4* __restrict__ sFarV = reinterpret_cast<V4*>(smem + Cfg::win_bytes + 2 * Cfg::w_tile_bytes);
  const T* __restrict__ sFarF = reinterpret_cast<const T*>(smem + Cfg::win_bytes + 2 * Cfg::w_tile_bytes);

  int b, tile, chunk;
  decode_block(gm, b, tile, chunk);
  const int tid = threadIdx.x, lane = tid & 63, wave64 = tid & ~63;
  const int g = tid & (TG - 1), rs = tid >> TGS;
  const int q0 = tile * TR, N = gm.N, C = gm.C;
  const int cg = chunk * TG + g;
  const T* __restrict__ Zb = dZ + (int64_t)b * N * C;
  const T* __restrict__ Wb = W + (int64_t)b * N * L;

  int prev0 = q0 - TR;
  if (prev0 < 0) prev0 += N;
  int misP, misC;
  if constexpr (ONEIMG) {
    // one image: float index 0 <-> element (prev0 row, col 0) - misP; the current tile follows at + TR*L. With
    // N*L % 4 == 0 the wrap (tile 0: prev rows are the last TR rows of the sequence) keeps the chunk phase.
    stage_flat_tile<T, VEC, NT, Cfg::w_passes, false>(W, w_total, ((int64_t)b * N + prev0) * L, TR * L, sWpV, misP);
    // current tile lands at float offset misP + TR*L: chunk-aligned iff (misP + TR*L) % 4 == misC; stage it through
    // a pointer shifted by whole chunks
    const int64_t e_lo = ((int64_t)b * N + q0) * L;
    const int mis = (int)(((reinterpret_cast<uintptr_t>(W) / sizeof(T)) + (uint64_t)e_lo) & (VEC - 1));
    const int base_f = misP + TR * L - mis;  // float index of the chunk that holds element e_lo - mis (multiple of 4)
    V4* img = sWpV + base_f / VEC;
    const int nvec = (mis + TR * L + VEC - 1) / VEC;
    const T* Gal = W + (e_lo - mis);
#pragma unroll
    for (int n = 0; n < Cfg::w_passes; ++n) {
      const int i = n * NT + tid;
      if (i < nvec) stage16<T, VEC, true>(Gal + (int64_t)i * VEC, img + n * NT + wave64, lane);
    }
    misC = misP + TR * L;
  } else {
    stage_flat_tile<T, VEC, NT, Cfg::w_passes, false>(W, w_total, ((int64_t)b * N + prev0) * L, TR * L, sWpV, misP);
    stage_flat_tile<T, VEC, NT, Cfg::w_passes, false>(W, w_total, ((int64_t)b * N + q0) * L, TR * L, sWcV, misC);
  }

#pragma unroll
  for (int n = 0; n < Cfg::win_vecs / NT; ++n) {
    const int i = n * NT + tid;
    const int wr = i >> TGS, gg = i & (TG - 1);
    int src = q0 - TR + wr;
    if (src < 0) src += N;
    if (src >= N) src -= N;
    stage16<T, VEC, true>(Zb + (int64_t)src * C + (int64_t)(chunk * TG + gg) * VEC, sWin + n * NT + wave64, lane);
  }

  if constexpr (SIDEDMA) {
    // far link f of tile [q0, q0+TR): Wfar[b][f][q0 - off_f .. + TR) — TR contiguous floats (the source tile is aligned:
    // far offsets are multiples of TR), TR/4 lanes x 16 B
    constexpr int lanes = TR / VEC;  // 64 at TR = 256
    for (int f = tid / lanes; f < NF; f += NT / lanes) {
      int src0 = q0 - offs.v[KN + f];
      if (src0 < 0) src0 += N;
      const T* gsrc = Wfar + ((int64_t)(b * NF + f) * N + src0) + (tid % lanes) * VEC;
      stage16<T, VEC, true>(gsrc, sFarV + f * lanes + ((tid % lanes) & ~63), lane);
    }
  }

  V4 farZ[R][NF > 0 ? NF : 1];
  T farW[R][NF > 0 ? NF : 1];
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const int q = q0 + j * RS + rs;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      int src = q - offs.v[KN + f];
      if (src < 0) src += N;
      if constexpr (NOFARZ) farZ[j][f] = V4{{1.f, 2.f, 3.f, (float)f}};
      else farZ[j][f] = ld<T, VEC>(Zb + (int64_t)src * C + (int64_t)cg * VEC);
      if constexpr (CONSTW) farW[j][f] = 0.25f;
      else if constexpr (SIDE) farW[j][f] = Wfar[(int64_t)(b * NF + f) * N + src];
      else if constexpr (SIDEDMA) farW[j][f] = 0.f;  // read from LDS after the barrier
      else farW[j][f] = Wb[(int64_t)src * L + (KN + f)];
    }
  }
  __syncthreads();

#pragma unroll
  for (int j = 0; j < R; ++j) {
    const int pl = j * RS + rs;
    const int q = q0 + pl;
    V4 acc;
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc.e[i] = T(0);
#pragma unroll
    for (int k = 0; k < KN; ++k) {
      const int off = CHORD ? near_off(k) : offs.v[k];
      const int wr = TR + pl - off;
      T w;
      if constexpr (ONEIMG) w = sWpF[misP + wr * L + k];
      else w = wr < TR ? sWpF[misP + wr * L + k] : sWcF[misC + (wr - TR) * L + k];
      axpy_rn<T, VEC>(acc, w, sWin[(wr << TGS) + g]);
    }
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      T w = farW[j][f];
      if constexpr (SIDEDMA) w = sFarF[f * TR + pl];
      axpy_rn<T, VEC>(acc, w, farZ[j][f]);
    }
    st<T, VEC>(dV + ((int64_t)b * N + q) * C + (int64_t)cg * VEC, acc);
  }
}

struct Case {
  const char* name;
  int R;  // rows per 256 threads: TR = 128 * R (side copy selection)
  bool exact;
  void (*launch)(const float*, const float*, const float*, float*, const Geom&, const Offsets&, int64_t);
  std::vector<double> us;
};

constexpr int LL = 15, TGSS = 1, NTT = 256;

template <int R>
void launch_prod(const float* dZ, const float* W, const float*, float* dV, const Geom& gm, const Offsets& offs, int64_t wt) {
  using Cfg = BwdWinCfg<float, LL, TGSS, R, NTT>;
  auto k = chord_dv_win_k<float, LL, TGSS, R, NTT, false>;
  static bool once = (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::lds_dv), true);
  (void)once;
  hipLaunchKernelGGL(k, dim3(gm.nblocks), dim3(NTT), Cfg::lds_dv, 0, dZ, W, dV, gm, offs, wt, (const float*)nullptr, 0);
}

template <int R, int FLAGS, int NT = NTT>
void launch_lab(const float* dZ, const float* W, const float* Wfar, float* dV, const Geom& gm, const Offsets& offs, int64_t wt) {
  using Cfg = BwdWinCfg<float, LL, TGSS, R, NT>;
  auto k = lab_dv_k<LL, TGSS, R, NT, FLAGS>;
  constexpr int lds = Cfg::lds_dv + ((FLAGS & 4) ? Cfg::NF * Cfg::TR * 4 : 0);
  static bool once = (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds), true);
  (void)once;
  hipLaunchKernelGGL(k, dim3(gm.nblocks), dim3(NT), lds, 0, dZ, W, Wfar, dV, gm, offs, wt);
}

Geom geom(int B, int N, int L, int C, int tgs, int R, int NT) {
  Geom gm;
  gm.N = N; gm.L = L; gm.C = C; gm.CG = C / 4; gm.tg_shift = tgs; gm.TR = (NT >> tgs) * R;
  gm.tiles_n = (N + gm.TR - 1) / gm.TR; gm.tile0 = 0; gm.chunks_c = 1; gm.per_b = gm.tiles_n;
  gm.nblocks = B * gm.per_b; gm.xq = gm.nblocks / 8; gm.xr = gm.nblocks % 8; gm.remap = 1; gm.v_bstride = (int64_t)N * C;
  return gm;
}

int main() {
  const int B = 64, N = 16384, L = LL, C = 8;
  const size_t wn = (size_t)B * N * L, vn = (size_t)B * N * C;
  float *W, *dZ, *dV, *dVref, *Wfar;
  CK(hipMalloc(&W, wn * 4));
  CK(hipMalloc(&dZ, vn * 4));
  CK(hipMalloc(&dV, vn * 4));
  CK(hipMalloc(&dVref, vn * 4));
  std::vector<float> hW(wn), hZ(vn);
  for (size_t i = 0; i < wn; ++i) hW[i] = 0.1f * ((float)rand() / RAND_MAX - 0.5f);
  for (size_t i = 0; i < vn; ++i) hZ[i] = (float)rand() / RAND_MAX - 0.5f;
  CK(hipMemcpy(W, hW.data(), wn * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dZ, hZ.data(), vn * 4, hipMemcpyHostToDevice));
  Offsets offs{};
  for (int k = 1; k < L; ++k) offs.v[k] = 1 << (k - 1);

  // side copies for R = 2 (KN = 10, NF = 5) and R = 1 (KN = 9, NF = 6): Wfar[b][f][n] = W[b][n][KN + f]
  float* WfarR[3] = {nullptr, nullptr, nullptr};
  for (int R = 1; R <= 2; ++R) {
    const int TR = (NTT >> TGSS) * R;
    int KN = 2;
    for (int t = TR; t > 1; t >>= 1) ++KN;
    const int NF = L - KN;
    std::vector<float> h((size_t)B * NF * N);
    for (int b = 0; b < B; ++b)
      for (int f = 0; f < NF; ++f)
        for (int n = 0; n < N; ++n) h[((size_t)b * NF + f) * N + n] = hW[((size_t)b * N + n) * L + KN + f];
    CK(hipMalloc(&WfarR[R], h.size() * 4));
    CK(hipMemcpy(WfarR[R], h.data(), h.size() * 4, hipMemcpyHostToDevice));
  }
  (void)Wfar;

  std::vector<Case> cases = {
      {"prod R=2", 2, true, launch_prod<2>, {}},
      {"prod R=1", 1, true, launch_prod<1>, {}},
      {"lab  R=2 (same as prod)", 2, true, launch_lab<2, 0>, {}},
      {"lab  R=2 CHORD", 2, true, launch_lab<2, 1>, {}},
      {"lab  R=2 CHORD ONEIMG", 2, true, launch_lab<2, 1 | 16>, {}},
      {"lab  R=2 CHORD SIDE(regs)", 2, true, launch_lab<2, 1 | 2>, {}},
      {"lab  R=2 CHORD SIDEDMA", 2, true, launch_lab<2, 1 | 4>, {}},
      {"lab  R=2 CHORD ONEIMG SIDEDMA", 2, true, launch_lab<2, 1 | 4 | 16>, {}},
      {"lab  R=2 CHORD CONSTW (ablation)", 2, false, launch_lab<2, 1 | 8>, {}},
      {"lab  R=2 CHORD CONSTW NOFARZ (abl)", 2, false, launch_lab<2, 1 | 8 | 32>, {}},
      {"lab  R=1 CHORD", 1, true, launch_lab<1, 1>, {}},
      {"lab  R=1 CHORD ONEIMG SIDE(regs)", 1, true, launch_lab<1, 1 | 2 | 16>, {}},
      {"lab  R=1 CHORD CONSTW (ablation)", 1, false, launch_lab<1, 1 | 8>, {}},
      {"lab  NT=512 R=1 CHORD", 2, true, launch_lab<1, 1, 512>, {}},
      {"lab  NT=512 R=1 CHORD SIDE(regs)", 2, true, launch_lab<1, 1 | 2, 512>, {}},
      {"lab  NT=512 R=2 CHORD (TR=512)", 4, true, launch_lab<2, 1, 512>, {}},
      {"lab  NT=1024 R=1 CHORD (TR=512)", 4, true, launch_lab<1, 1, 1024>, {}},
  };
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  // reference output
  {
    Geom gm = geom(B, N, L, C, TGSS, 2, NTT);
    launch_prod<2>(dZ, W, nullptr, dVref, gm, offs, (int64_t)wn);
    CK(hipDeviceSynchronize());
  }
  std::vector<float> href(vn), hout(vn);
  CK(hipMemcpy(href.data(), dVref, vn * 4, hipMemcpyDeviceToHost));
  const int rounds = 5, iters = 30;
  for (int r = 0; r < rounds + 1; ++r) {
    for (auto& c : cases) {
      Geom gm = geom(B, N, L, C, TGSS, c.R, NTT);  // TR = 128 * c.R whatever the thread count
      const float* wf = c.R <= 2 ? WfarR[c.R] : nullptr;
      if (r == 0) {
        CK(hipMemset(dV, 0xff, vn * 4));
        c.launch(dZ, W, wf, dV, gm, offs, (int64_t)wn);
        CK(hipDeviceSynchronize());
        CK(hipGetLastError());
        if (c.exact) {
          CK(hipMemcpy(hout.data(), dV, vn * 4, hipMemcpyDeviceToHost));
          size_t bad = 0;
          for (size_t i = 0; i < vn; ++i) bad += memcmp(&hout[i], &href[i], 4) != 0;
          printf("%-40s %s\n", c.name, bad ? "MISMATCH" : "bit-exact");
          if (bad) printf("    %zu of %zu differ\n", bad, vn);
        }
        continue;
      }
      CK(hipEventRecord(e0));
      for (int it = 0; it < iters; ++it) c.launch(dZ, W, wf, dV, gm, offs, (int64_t)wn);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipGetLastError());
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      c.us.push_back(ms * 1e3 / iters);
    }
  }
  const double bytes = 4.0 * B * N * (L + 2 * C);
  for (auto& c : cases) {
    std::sort(c.us.begin(), c.us.end());
    const double med = c.us[c.us.size() / 2];
    printf("%-40s %7.2f us/launch (min %6.2f)  %5.2f TB/s  %.3f of 8 TB/s\n", c.name, med, c.us[0], bytes / med / 1e6,
           bytes / med / 8e6);
  }
  return 0;
}
