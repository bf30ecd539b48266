// fwdlab.hip — ablation lab for the forward step at cfg2 (N=16384, L=15, C=8, B=64, residual). NOT product code.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -ffp-contract=off profiles/fwdlab.hip -o /tmp/fwdlab && /tmp/fwdlab
// Times the shipped window kernel next to variants with one part removed (outputs of those are wrong on
// purpose; values are kept alive with empty asm so nothing is dead-code-eliminated), interleaved in one process.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../sparsefactorization_amd/csrc/fwd_window.h"

using namespace psf;

#define CK(x)                                                                   \
  do {                                                                          \
    hipError_t e_ = (x);                                                        \
    if (e_ != hipSuccess) {                                                     \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); \
      exit(1);                                                                  \
    }                                                                           \
  } while (0)

// ABL bit 0: no near accumulate (LDS reads skipped)   bit 1: no far loads   bit 2: no W/window staging
// ABL bit 3: no residual load                          bit 4: no accumulate arithmetic at all (store far[0])
template <int L, int TGS, int R, int NT, int ABL>
__global__ void __launch_bounds__(NT)
lab_k(const float* __restrict__ W, const float* __restrict__ V, const float* __restrict__ res,
      float* __restrict__ out, const Geom gm, const Offsets offs, const int64_t w_total) {
  using T = float;
  using Cfg = FwdWinCfg<T, L, TGS, R, NT>;
  constexpr int VEC = 4, TG = Cfg::TG, RS = Cfg::RS, TR = Cfg::TR, KN = Cfg::KN, NF = Cfg::NF;
  using V4 = Vec<T, VEC>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  V4* __restrict__ sWin = reinterpret_cast<V4*>(smem);
  V4* __restrict__ sWv = reinterpret_cast<V4*>(smem + Cfg::win_bytes);
  const T* __restrict__ sWf = reinterpret_cast<const T*>(smem + Cfg::win_bytes);
  int b, tile, chunk;
  decode_block(gm, b, tile, chunk);
  const int tid = threadIdx.x, lane = tid & 63, wave64 = tid & ~63;
  const int g = tid & (TG - 1), rs = tid >> TGS, p0 = tile * TR, N = gm.N, C = gm.C;
  const int cg = chunk * TG + g;
  const T* __restrict__ Vb = V + (int64_t)b * gm.v_bstride;
  const int64_t e_lo = ((int64_t)b * N + p0) * L;
  const int mis = (int)(((reinterpret_cast<uintptr_t>(W) / sizeof(T)) + (uint64_t)e_lo) & (VEC - 1));
  const int64_t e_al = e_lo - mis;
  const int nvec = (mis + TR * L + VEC - 1) / VEC;
  const T* __restrict__ Wal = W + e_al;
  if (!(ABL & 4)) {
#pragma unroll
    for (int n = 0; n < Cfg::w_passes; ++n) {
      const int i = n * NT + tid;
      if (i < nvec) stage16<T, VEC, true>(Wal + (int64_t)i * VEC, sWv + n * NT + wave64, lane);
    }
#pragma unroll
    for (int n = 0; n < Cfg::win_vecs / NT; ++n) {
      const int i = n * NT + tid;
      const int wr = i >> TGS, gg = i & (TG - 1);
      int src = p0 + wr;
      if (src >= N) src -= N;
      stage16<T, VEC, true>(Vb + (int64_t)src * C + (int64_t)(chunk * TG + gg) * VEC, sWin + n * NT + wave64, lane);
    }
  }
  V4 far[R][NF > 0 ? NF : 1];
  V4 rres[R];
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const int p = p0 + j * RS + rs;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
      int src = p + offs.v[KN + f];
      if (src >= N) src -= N;
      if (!(ABL & 2)) far[j][f] = ld<T, VEC>(Vb + (int64_t)src * C + (int64_t)cg * VEC);
      else far[j][f] = V4{{1.f, 2.f, 3.f, (float)f}};
    }
    if (!(ABL & 8)) rres[j] = ld<T, VEC>(res + ((int64_t)b * N + p) * C + (int64_t)cg * VEC);
    else rres[j] = V4{{0.f, 0.f, 0.f, 0.f}};
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const int pl = j * RS + rs;
    const int p = p0 + pl;
    V4 acc = V4{{0.f, 0.f, 0.f, 0.f}};
    if (ABL & 16) {
#pragma unroll
      for (int f = 0; f < NF; ++f) asm volatile("" ::"v"(far[j][f].e[0]), "v"(far[j][f].e[3]));
      acc = far[j][0];
    } else {
      const T* __restrict__ wrow = sWf + mis + pl * L;
      if (!(ABL & 1)) {
#pragma unroll
        for (int k = 0; k < KN; ++k) axpy_rn<T, VEC>(acc, wrow[k], sWin[((pl + offs.v[k]) << TGS) + g]);
      }
#pragma unroll
      for (int f = 0; f < NF; ++f) axpy_rn<T, VEC>(acc, (ABL & 1) ? 0.5f : wrow[KN + f], far[j][f]);
    }
#pragma unroll
    for (int i = 0; i < VEC; ++i) acc.e[i] = add_rn(acc.e[i], rres[j].e[i]);
    st<T, VEC>(out + ((int64_t)b * N + p) * C + (int64_t)cg * VEC, acc);
  }
}

// Persistent variant: gridDim.x workgroups (a multiple of 8) walk the tiles of their XCD's contiguous share.
// PF = 0: same schedule as the product kernel inside the loop. PF = 1: the NEXT tile's W tile and window are
// DMA'd into the other LDS buffer right after the barrier, so they fly during this tile's accumulate.
template <int L, int TGS, int R, int NT, int PF>
__global__ void __launch_bounds__(NT)
persist_k(const float* __restrict__ W, const float* __restrict__ V, const float* __restrict__ res,
          float* __restrict__ out, const Geom gm, const Offsets offs, const int64_t w_total) {
  using T = float;
  using Cfg = FwdWinCfg<T, L, TGS, R, NT>;
  constexpr int VEC = 4, TG = Cfg::TG, RS = Cfg::RS, TR = Cfg::TR, KN = Cfg::KN, NF = Cfg::NF;
  using V4 = Vec<T, VEC>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int BUF = Cfg::lds_bytes;
  const int tid = threadIdx.x, lane = tid & 63, wave64 = tid & ~63;
  const int g = tid & (TG - 1), rs = tid >> TGS, N = gm.N, C = gm.C;
  const int xcd = blockIdx.x & 7, local = blockIdx.x >> 3, gx = gridDim.x >> 3;
  const int per_xcd = (int)(gm.nblocks >> 3);  // lab: nblocks % 8 == 0

  auto stage = [&](int lb, char* buf, int& mis_out) {
    const int b = lb / gm.per_b, tile = lb - b * gm.per_b;
    const int p0 = tile * TR;
    V4* sWin = reinterpret_cast<V4*>(buf);
    V4* sWv = reinterpret_cast<V4*>(buf + Cfg::win_bytes);
    const T* Vb = V + (int64_t)b * gm.v_bstride;
    const int64_t e_lo = ((int64_t)b * N + p0) * L;
    const int mis = (int)(((reinterpret_cast<uintptr_t>(W) / sizeof(T)) + (uint64_t)e_lo) & (VEC - 1));
    const T* Wal = W + (e_lo - mis);
    const int nvec = (mis + TR * L + VEC - 1) / VEC;
#pragma unroll
    for (int n = 0; n < Cfg::w_passes; ++n) {
      const int i = n * NT + tid;
      if (i < nvec) stage16<T, VEC, true, 2>(Wal + (int64_t)i * VEC, sWv + n * NT + wave64, lane);
    }
#pragma unroll
    for (int n = 0; n < Cfg::win_vecs / NT; ++n) {
      const int i = n * NT + tid;
      const int wr = i >> TGS, gg = i & (TG - 1);
      int src = p0 + wr;
      if (src >= N) src -= N;
      stage16<T, VEC, true>(Vb + (int64_t)src * C + (int64_t)gg * VEC, sWin + n * NT + wave64, lane);
    }
    mis_out = mis;
  };

  int it = 0;
  int mis_cur = 0, mis_next = 0;
  if (PF && local < per_xcd) stage(xcd * per_xcd + local, smem, mis_cur);
  for (int i = local; i < per_xcd; i += gx, ++it) {
    const int lb = xcd * per_xcd + i;
    char* buf = smem + (PF ? (it & 1) * BUF : 0);
    if (!PF) stage(lb, buf, mis_cur);
    const int b = lb / gm.per_b, tile = lb - b * gm.per_b;
    const int p0 = tile * TR;
    const T* Vb = V + (int64_t)b * gm.v_bstride;
    V4 far[R][NF > 0 ? NF : 1];
    V4 rres[R];
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const int p = p0 + j * RS + rs;
#pragma unroll
      for (int f = 0; f < NF; ++f) {
        int src = p + offs.v[KN + f];
        if (src >= N) src -= N;
        far[j][f] = ld<T, VEC>(Vb + (int64_t)src * C + (int64_t)g * VEC);
      }
      rres[j] = ld<T, VEC>(res + ((int64_t)b * N + p) * C + (int64_t)g * VEC);
    }
    __syncthreads();
    if (PF && i + gx < per_xcd) stage(lb + gx, smem + ((it + 1) & 1) * BUF, mis_next);
    const V4* sWin = reinterpret_cast<const V4*>(buf);
    const T* sWf = reinterpret_cast<const T*>(buf + Cfg::win_bytes);
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const int pl = j * RS + rs;
      V4 acc = V4{{0.f, 0.f, 0.f, 0.f}};
      const T* wrow = sWf + mis_cur + pl * L;
#pragma unroll
      for (int k = 0; k < KN; ++k) axpy_rn<T, VEC>(acc, wrow[k], sWin[((pl + offs.v[k]) << TGS) + g]);
#pragma unroll
      for (int f = 0; f < NF; ++f) axpy_rn<T, VEC>(acc, wrow[KN + f], far[j][f]);
#pragma unroll
      for (int e = 0; e < VEC; ++e) acc.e[e] = add_rn(acc.e[e], rres[j].e[e]);
      st<T, VEC>(out + ((int64_t)b * N + p0 + pl) * C + (int64_t)g * VEC, acc);
    }
    if (PF) mis_cur = mis_next;
    else __syncthreads();
  }
  (void)w_total;
}

struct Case {
  const char* name;
  int R, NT;
  void (*launch)(const float*, const float*, const float*, float*, const Geom&, const Offsets&, int64_t);
  std::vector<double> us;
};

template <int R, int NT, int ABL>
void launch_lab(const float* W, const float* V, const float* res, float* out, const Geom& gm, const Offsets& offs, int64_t wt) {
  using Cfg = FwdWinCfg<float, 15, 1, R, NT>;
  auto k = lab_k<15, 1, R, NT, ABL>;
  static bool once = (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::lds_bytes), true);
  (void)once;
  hipLaunchKernelGGL(k, dim3(gm.nblocks), dim3(NT), Cfg::lds_bytes, 0, W, V, res, out, gm, offs, wt);
}
template <int R, int NT, bool DMA>
void launch_prod(const float* W, const float* V, const float* res, float* out, const Geom& gm, const Offsets& offs, int64_t wt) {
  using Cfg = FwdWinCfg<float, 15, 1, R, NT>;
  auto k = chord_fwd_win_k<float, 15, 1, R, NT, DMA, /*RES=*/true, /*EDGE=*/false>;
  static bool once = (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::lds_bytes), true);
  (void)once;
  hipLaunchKernelGGL(k, dim3(gm.nblocks), dim3(NT), Cfg::lds_bytes, 0, W, V, res, out, gm, offs, wt, (float*)nullptr, 0);
}

template <int R, int NT, int PF, int GRID>
void launch_persist(const float* W, const float* V, const float* res, float* out, const Geom& gm, const Offsets& offs, int64_t wt) {
  using Cfg = FwdWinCfg<float, 15, 1, R, NT>;
  auto k = persist_k<15, 1, R, NT, PF>;
  const int lds = Cfg::lds_bytes * (PF ? 2 : 1);
  static bool once = (hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds), true);
  (void)once;
  hipLaunchKernelGGL(k, dim3(GRID), dim3(NT), lds, 0, W, V, res, out, gm, offs, wt);
}

Geom geom(int B, int N, int L, int C, int tgs, int R, int NT) {
  Geom gm;
  gm.N = N; gm.L = L; gm.C = C; gm.CG = C / 4; gm.tg_shift = tgs; gm.TR = (NT >> tgs) * R;
  gm.tiles_n = (N + gm.TR - 1) / gm.TR; gm.chunks_c = 1; gm.per_b = gm.tiles_n;
  gm.nblocks = B * gm.per_b; gm.xq = gm.nblocks / 8; gm.xr = gm.nblocks % 8; gm.remap = 1; gm.v_bstride = (int64_t)N * C;
  return gm;
}

int main() {
  const int B = 64, N = 16384, L = 15, C = 8, M = 14;
  const size_t wn = (size_t)B * N * L, vn = (size_t)B * N * C;
  float *W, *V0, *P0, *P1;
  CK(hipMalloc(&W, wn * 4 * M));
  CK(hipMalloc(&V0, vn * 4));
  CK(hipMalloc(&P0, vn * 4));
  CK(hipMalloc(&P1, vn * 4));
  std::vector<float> h(wn);
  for (size_t i = 0; i < wn; ++i) h[i] = 0.1f * ((float)rand() / RAND_MAX - 0.5f);
  for (int m = 0; m < M; ++m) CK(hipMemcpy(W + (size_t)m * wn, h.data(), wn * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(V0, h.data(), vn * 4, hipMemcpyHostToDevice));
  CK(hipMemset(P0, 0, vn * 4));
  CK(hipMemset(P1, 0, vn * 4));
  Offsets offs{};
  for (int k = 1; k < L; ++k) offs.v[k] = 1 << (k - 1);

  std::vector<Case> cases = {
      {"prod NT=256  R=1 dma", 1, 256, launch_prod<1, 256, true>, {}},
      {"prod NT=256  R=2 dma", 2, 256, launch_prod<2, 256, true>, {}},
      {"lab  NT=256  R=2 full", 2, 256, launch_lab<2, 256, 0>, {}},
      {"persist R=2 PF=0 grid=1280", 2, 256, launch_persist<2, 256, 0, 1280>, {}},
      {"persist R=2 PF=0 grid=1024", 2, 256, launch_persist<2, 256, 0, 1024>, {}},
      {"persist R=2 PF=0 grid=2048", 2, 256, launch_persist<2, 256, 0, 2048>, {}},
      {"persist R=1 PF=0 grid=2048", 1, 256, launch_persist<1, 256, 0, 2048>, {}},
      {"persist R=2 PF=1 grid=512", 2, 256, launch_persist<2, 256, 1, 512>, {}},
      {"persist R=1 PF=1 grid=1280", 1, 256, launch_persist<1, 256, 1, 1280>, {}},
      {"persist R=1 PF=1 grid=1024", 1, 256, launch_persist<1, 256, 1, 1024>, {}},
      {"persist R=1 PF=1 grid=768", 1, 256, launch_persist<1, 256, 1, 768>, {}},
  };
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const int rounds = 5, chains = 10;
  for (int r = 0; r < rounds + 1; ++r) {
    for (auto& c : cases) {
      Geom gm = geom(B, N, L, C, 1, c.R, c.NT);
      CK(hipEventRecord(e0));
      for (int it = 0; it < chains; ++it)
        for (int m = 0; m < M; ++m) {
          const float* in = m == 0 ? V0 : ((m - 1) & 1 ? P1 : P0);
          float* o = (m & 1) ? P1 : P0;
          c.launch(W + (size_t)m * wn, in, V0, o, gm, offs, (int64_t)wn);
        }
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
      CK(hipGetLastError());
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (r > 0) c.us.push_back(ms * 1e3 / (chains * M));
    }
  }
  const double bytes = 4.0 * B * N * (L + 3 * C);
  for (auto& c : cases) {
    std::sort(c.us.begin(), c.us.end());
    const double med = c.us[c.us.size() / 2];
    printf("%-42s %7.2f us/launch (min %6.2f)  %6.0f GB/s-equivalent\n", c.name, med, c.us[0], bytes / med / 1e3);
  }
  return 0;
}
