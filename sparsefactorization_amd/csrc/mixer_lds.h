// mixer_lds.h — the WHOLE mixer of a short sequence in one launch: V0 = g(data), then M steps V <- W_m V (+ V0) with every
// W_m = fs[m](data) computed on chip, the sequence's V resident in LDS (SURVEY.md §8(f) row 3 on the kernel that already had
// the synchronisation: fwd_chain_lds.h). Reference: SyntheticExperiments/psf.py:165-188 (the loop the reference runs as
// 1 + M MLP calls and M spmm calls through memory).
//
// One workgroup = one sequence with all its C <= 8 channels. Per step: each wave evaluates the step's MLP on its token tiles
// (32 rows each; the data rows stay in registers, split per step — mlp_tile of fwd_mlp_step.h) and writes the W rows to LDS;
// after a barrier every thread accumulates its rows' L links from the LDS-resident X_m (links ascending, uncontracted
// multiply and add: the arithmetic of chord_chain_lds_k and of the oracle) into the other X buffer; the next step's weight
// image streams in by LDS-DMA meanwhile. HBM traffic of the whole mixer: the data rows (or their recipe), the weight images,
// and whatever step results the caller wants stored. No W_m, no `data`, no intermediate V in memory.
//
// Limits (plan_mixer_lds): N a multiple of 32, N * C / 4 <= 1024 slots (<= 512 threads), C in {4, 8}, at most two token
// tiles per wave, L <= 20, E <= 32, h <= 128. Longer sequences take the per-step kernels (fwd_mlp_step.h).
#pragma once

#include "fwd_mlp_step.h"
#include "mixer_lds_launch.h"

namespace psf {

constexpr int mixer_lds_ws(int L) { return L <= 12 ? 12 : 20; }

template <bool RES, int KIND>
__global__ void __launch_bounds__(512)
chord_mixer_lds_k(const MixerLdsArgs a, const Offsets offs) {
  using namespace psf_x3;
  using F4 = float __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int N = a.N, CG = a.CG, WS = a.WS, slots = N * CG;
  F4* __restrict__ xb = reinterpret_cast<F4*>(smem);                       // two X buffers of `slots` vectors
  float* __restrict__ sW = reinterpret_cast<float*>(smem + 2 * slots * 16);  // N rows of WS floats
  unsigned char* __restrict__ sImg = reinterpret_cast<unsigned char*>(smem + 2 * slots * 16 + N * WS * 4);
  float* __restrict__ sAff = reinterpret_cast<float*>(sImg + a.nu_max * kImgBytes);

  const int tid = threadIdx.x, lane = tid & 63, wave64 = tid & ~63, nthreads = blockDim.x;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6), nwaves = nthreads >> 6;
  const int c = lane & 31, half = lane >> 5;
  const int b = blockIdx.x;
  const int g = tid % CG, rs = tid / CG, RSN = nthreads / CG;  // accumulate phase: channel group, first row; rows rs + j RSN

  auto stage_image = [&](int k) {  // the units of MLP k -> sImg (LDS-DMA; the caller orders it against the readers)
    const unsigned char* src = a.images + (size_t)a.first_unit[k] * kImgBytes;
    const int vecs = (a.first_unit[k + 1] - a.first_unit[k]) * kImgVecs;
    for (int v0 = 0; v0 < vecs; v0 += nthreads) {
      const int v = v0 + tid;
      if (v < vecs)
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 16 * (size_t)v),
                                         (__attribute__((address_space(3))) void*)(sImg + 16 * (v0 + wave64)), 16, 0, 0);
    }
  };

  stage_image(0);
  if constexpr (KIND == 1) {
    stage_affine(a.in, sAff, a.E, tid);
    __syncthreads();
  }
  // this wave's token tiles t = wv, wv + nwaves (at most two): their data rows, kept for all M + 1 MLPs
  float xv[2][2][8];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int t = wv + i * nwaves;
    const int row = 32 * (t < a.TT ? t : 0) + c;
#pragma unroll
    for (int s = 0; s < 2; ++s) data_row8<KIND>(a.in, sAff, b, row, N, a.E, 16 * s + 8 * half, false, xv[i][s]);
  }
  __syncthreads();  // image of g landed

  // V0 = g(data): the tile's C <= 8 channels are registers 0..3 of the lane (channels 4 half .. 4 half + 3 of its token)
  {
    const int nu = a.first_unit[1] - a.first_unit[0];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int t = wv + i * nwaves;
      if (t >= a.TT) break;  // wave-uniform
      const f32x16 y = mlp_tile(sImg, 0, nu, true, xv[i], c, half);
      if (half < CG) xb[(32 * t + c) * CG + half] = F4{y[0], y[1], y[2], y[3]};
    }
  }
  __syncthreads();  // X_0 complete; every wave is done with g's image
  stage_image(1);
  F4 resv[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int p = rs + j * RSN;
    resv[j] = p < N ? xb[p * CG + g] : F4{0.f, 0.f, 0.f, 0.f};
    if (p < N && a.V0 != nullptr) *reinterpret_cast<F4*>(a.V0 + ((int64_t)b * N + p) * a.C + 4 * g) = resv[j];
  }
  __syncthreads();  // image of fs[0] landed (hipcc drains vmcnt before the barrier)

  int cur = 0;
  for (int m = 0; m < a.M; ++m) {
    // (1) W_m rows of this wave's tiles -> sW
    {
      const int nu = a.first_unit[m + 2] - a.first_unit[m + 1];
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int t = wv + i * nwaves;
        if (t >= a.TT) break;
        const f32x16 y = mlp_tile(sImg, 0, nu, true, xv[i], c, half);
        float* dst = sW + (32 * t + c) * WS;
#pragma unroll
        for (int q = 0; q < 3; ++q)
          if (8 * q + 4 * half < a.L)
            *reinterpret_cast<float4*>(dst + 8 * q + 4 * half) = make_float4(y[4 * q], y[4 * q + 1], y[4 * q + 2], y[4 * q + 3]);
      }
    }
    __syncthreads();  // W_m complete; every wave is done with this step's image
    if (m + 1 < a.M) stage_image(m + 2);  // lands during the accumulate phase

    // (2) X_{m+1}[p] = sum_k W_m[p,k] X_m[(p + off_k) mod N] (+ V0[p]), links ascending. Links are taken five at a time and
    //     every LDS operand of a group is waited for in full before its arithmetic (fwd_mlp_step.h, phase 2, says why).
    const bool store = (a.store_mask >> m) & 1;
    float* __restrict__ om = a.out[m];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int p = rs + j * RSN;
      if (p < N) {
        float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
        const float* __restrict__ wrow = sW + p * WS;
        for (int k0 = 0; k0 < a.L; k0 += 5) {
          F4 x[5];
          float w[5];
#pragma unroll
          for (int i = 0; i < 5; ++i) {
            const int k = k0 + i < a.L ? k0 + i : a.L - 1;
            int src = p + offs.v[k];
            if (src >= N) src -= N;
            x[i] = xb[cur + src * CG + g];
            w[i] = wrow[k];
          }
          lds_wait_all();
#pragma unroll
          for (int i = 0; i < 5; ++i) behind_wait(x[i]), behind_wait(w[i]);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int i = 0; i < 5; ++i)
            if (k0 + i < a.L) {
              acc0 = add_rn(acc0, mul_rn(w[i], x[i].x));
              acc1 = add_rn(acc1, mul_rn(w[i], x[i].y));
              acc2 = add_rn(acc2, mul_rn(w[i], x[i].z));
              acc3 = add_rn(acc3, mul_rn(w[i], x[i].w));
            }
        }
        if constexpr (RES) {
          acc0 = add_rn(acc0, resv[j].x), acc1 = add_rn(acc1, resv[j].y);
          acc2 = add_rn(acc2, resv[j].z), acc3 = add_rn(acc3, resv[j].w);
        }
        const F4 r = F4{acc0, acc1, acc2, acc3};
        xb[(slots - cur) + p * CG + g] = r;
        if (store) *reinterpret_cast<F4*>(om + ((int64_t)b * N + p) * a.C + 4 * g) = r;
      }
    }
    __syncthreads();  // X_{m+1} complete, the next image landed, sW free
    cur = slots - cur;
  }
}

}  // namespace psf
