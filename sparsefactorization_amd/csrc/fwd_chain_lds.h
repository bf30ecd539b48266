// fwd_chain_lds.h — the WHOLE forward chain in one launch for short sequences, with V resident in LDS.
//
//     X_0 = V0;  X_{m+1}[p,:] = sum_k W_m[p,k] * X_m[(p+off_k) mod N,:]  (+ V0[p,:])     m = 0 .. M-1
//                                                           (SyntheticExperiments/psf.py:167-188)
//
// The operator never mixes channels, so a workgroup that owns ONE sequence and ONE group of CC*4 channels can
// run all M steps by itself: its slice of X_m lives in LDS (two buffers, N*CC*16 bytes each), every link — near
// or far, any offset — is an LDS read, and the only HBM traffic is W_m (read once per step, 4-byte-aligned
// dwordx4 row loads, prefetched one step ahead) plus whatever step results the caller wants stored. That removes
// the per-step V / residual / output traffic (100 of 156 B per row-step at cfg2's mix) and the M-1 kernel
// boundaries, which dominate for short sequences (per-step kernels at N <= 2048 run 6-19 us of mostly launch and
// fill/drain). Fits N * CC <= 2112 (row, channel-group) slots, i.e. N <= 2112: the LRA ListOps / Pathfinder /
// CIFAR-10 configurations with or without a CLS-token column (N = 2^k + 1) and the synthetic tasks up to N = 2048. Launches
// of >= 256 workgroups at 1057 <= N <= 2048 (eight channels per workgroup) and 2113 <= N <= 4160 run chord_chain_rows_k
// (below: a thread owns whole rows); other sequences use the per-step kernels.
//
// Thread (rs, g): channel group g < CC of rows rs + j*RSN, j < R (RSN = blockDim / CC row slots). The same thread
// produces the same rows at every step, so the residual rows stay in registers. Summation order and arithmetic
// are those of the per-step kernels: results are bit-identical.
#pragma once

#include "psf_common.h"

namespace psf {

constexpr int kChainMaxSteps = 64;
constexpr int kChainBigRows = 2048;   // chord_chain_rows_k<G = 2>: rows a workgroup can hold (two channel groups each)
constexpr int kChainLongRows = 4160;  // chord_chain_rows_k<G = 1>: rows a workgroup can hold (one channel group)

struct ChainArgs {
  const float* W[kChainMaxSteps];  // W_m [B, N, L]
  float* out[kChainMaxSteps];      // X_{m+1} [B, N, C]; written when bit m of store_mask is set
  const float* V0;                 // [B, N, C] or [N, C] (v0_bstride == 0)
  uint64_t store_mask;
  int64_t v0_bstride;
  int32_t M, N, C, CG, chunks;     // CG = C / 4 channel groups, chunks = ceil(CG / CC) workgroups per sequence
  int32_t xcd_remap;               // 1: consecutive logical workgroups share an XCD (knob "xcd_remap")
};

template <int L>
struct __attribute__((packed, aligned(4))) WRow {
  float e[L];
};

// NTMAX (512 or 1024) is the launch bound: workgroups of <= 512 threads may use 256 VGPRs and run 2-4 per CU,
// which is what overlaps one workgroup's barrier / load latency with another's arithmetic.
template <int L, int CC, int R, bool RES, int NTMAX>
__global__ void __launch_bounds__(NTMAX)
chord_chain_lds_k(const ChainArgs a, const Offsets offs) {
  using V4 = Vec<float, 4>;
  // LDS is addressed as ONE array of 16-byte vectors with a buffer offset (0 / slots) per step: a true vector
  // type and a single base keep the accesses ds_read_b128 / ds_write_b128 (two swapped pointers made hipcc fall
  // back to ds_read2_b32 pairs, 4-way bank-conflicted at a 16-byte lane stride).
  using F4 = float __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) F4 xlds[];
  const int N = a.N, C = a.C;
  const int slots = N * CC;
  int cur = 0;  // buffer holding X_m; the other one receives X_{m+1}

  const int tid = threadIdx.x;
  const int g = tid & (CC - 1);
  const int rs = tid / CC;
  const int RSN = blockDim.x / CC;
  // The `chunks` workgroups of a sequence all stream the sequence's W: consecutive LOGICAL ids, which the XCD-aware map puts on
  // ONE XCD (hardware deals blockIdx.x round-robin over the eight XCDs), so that W crosses the fabric once per sequence and the
  // other chunks' reads hit that XCD's L2. Bijective for any grid (psf_common.h: logical_block).
  uint32_t lb = blockIdx.x;
  if (a.xcd_remap) {
    const uint32_t nb = gridDim.x, xq = nb / kXcds, xr = nb % kXcds, xcd = lb % kXcds, idx = lb / kXcds;
    lb = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + idx;
  }
  const int b = (int)(lb / (uint32_t)a.chunks);
  const int chunk = (int)(lb - (uint32_t)b * (uint32_t)a.chunks);
  const int cg = chunk * CC + g;
  const bool cg_ok = cg < a.CG;
  const int cgc = cg_ok ? cg : a.CG - 1;

  int prow[R];
  bool pok[R];
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const int p = rs + j * RSN;
    pok[j] = p < N;
    prow[j] = pok[j] ? p : N - 1;
  }

  // X_0 slice -> LDS (and the residual rows -> registers)
  const float* __restrict__ V0b = a.V0 + (int64_t)b * a.v0_bstride + (int64_t)cgc * 4;
  V4 resv[R];
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const V4 v = ld<float, 4>(V0b + (int64_t)prow[j] * C);
    if (pok[j]) xlds[prow[j] * CC + g] = F4{v.e[0], v.e[1], v.e[2], v.e[3]};
    resv[j] = v;
  }

  // W rows of step 0; from then on the rows of step m+1 are requested at the START of step m (W does not depend
  // on X), so they have a whole step — accumulate, LDS writes and the barrier — to land. 2*R*L registers, which
  // is why R stops at 2 for 1024-thread workgroups.
  WRow<L> w[R];
#pragma unroll
  for (int j = 0; j < R; ++j) w[j] = *reinterpret_cast<const WRow<L>*>(a.W[0] + ((int64_t)b * N + prow[j]) * L);
  __syncthreads();

  for (int m = 0; m < a.M; ++m) {
    const float* __restrict__ Wn = a.W[m + 1 < a.M ? m + 1 : m];
    WRow<L> wn[R];
#pragma unroll
    for (int j = 0; j < R; ++j) wn[j] = *reinterpret_cast<const WRow<L>*>(Wn + ((int64_t)b * N + prow[j]) * L);

    const bool store = (a.store_mask >> m) & 1;
    float* __restrict__ om = a.out[m];
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const int p = prow[j];
      V4 acc;
#pragma unroll
      for (int i = 0; i < 4; ++i) acc.e[i] = 0.f;
#pragma unroll
      for (int k = 0; k < L; ++k) {
        int src = p + offs.v[k];
        if (src >= N) src -= N;
        const F4 x = xlds[cur + src * CC + g];
        axpy_rn<float, 4>(acc, w[j].e[k], V4{{x.x, x.y, x.z, x.w}});
      }
      if constexpr (RES) {
#pragma unroll
        for (int i = 0; i < 4; ++i) acc.e[i] = add_rn(acc.e[i], resv[j].e[i]);
      }
      if (pok[j]) {
        xlds[(slots - cur) + p * CC + g] = F4{acc.e[0], acc.e[1], acc.e[2], acc.e[3]};
        if (store && cg_ok) st<float, 4>(om + ((int64_t)b * N + p) * C + (int64_t)cg * 4, acc);
      }
    }
    __syncthreads();
    cur = slots - cur;
#pragma unroll
    for (int j = 0; j < R; ++j) w[j] = wn[j];
  }
}

// The large instances (one workgroup per CU): a THREAD owns whole rows — G channel groups of each of its R rows.
//   G = 2, R = 2, 1057 <= N <= 2048: eight channels per workgroup. Half as many workgroups stream a sequence's W as with one
//          group each, and that stream's L2 requests are what bounds the one-launch chain at these lengths (ListOps,
//          N = 2000 x 128: 32 workgroups per sequence at 124 G requests/s, profiles/r06y_bwd_pmc.json).
//   G = 1, R = 5, 2113 <= N <= 4160: four channels per workgroup, the lengths of the LRA text task (N = 4096 + 1).
// One W row load per row instead of one per (row, group), one LDS address per link (group g sits CAP slots further: an
// immediate offset), and the W rows are requested one ROW ahead — row j+1's while row j accumulates, the next step's
// row 0 across the barrier — instead of a step ahead (2*R*L registers), which is what fits 128 registers on 1024
// threads. LDS: [buffer][group][CAP] vectors of 16 bytes. Arithmetic and summation order per channel are those of
// chord_chain_lds_k: bit-identical.
template <int L, int G, int R, int CAP, bool RES>
__global__ void __launch_bounds__(1024)
chord_chain_rows_k(const ChainArgs a, const Offsets offs) {
  using V4 = Vec<float, 4>;
  using F4 = float __attribute__((ext_vector_type(4)));
  static_assert(G == 1 || G == 2, "channel groups per thread");
  extern __shared__ __attribute__((aligned(16))) F4 xlds[];
  const int N = a.N, C = a.C;
  constexpr int GS = CAP;  // group stride in slots: a constant, so the second group is an immediate offset
  constexpr int slots = G * GS;
  int cur = 0;
  const int tid = threadIdx.x;
  const int RSN = blockDim.x;
  uint32_t lb = blockIdx.x;
  if (a.xcd_remap) {  // (as in chord_chain_lds_k: the workgroups of a sequence on one XCD)
    const uint32_t nb = gridDim.x, xq = nb / kXcds, xr = nb % kXcds, xcd = lb % kXcds, idx = lb / kXcds;
    lb = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + idx;
  }
  const int b = (int)(lb / (uint32_t)a.chunks);
  const int chunk = (int)(lb - (uint32_t)b * (uint32_t)a.chunks);
  const int cg0 = chunk * G;
  const bool g1_ok = G == 2 && cg0 + 1 < a.CG;          // (an odd number of channel groups: the last workgroup owns one)
  const int cg1 = g1_ok ? cg0 + 1 : cg0;

  int prow[R];
  bool pok[R];
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const int p = tid + j * RSN;
    pok[j] = p < N;
    prow[j] = pok[j] ? p : N - 1;
  }

  const float* __restrict__ V0b = a.V0 + (int64_t)b * a.v0_bstride;
  V4 resv[R][2];
#pragma unroll
  for (int j = 0; j < R; ++j) {
    const V4 v0 = ld<float, 4>(V0b + (int64_t)prow[j] * C + (int64_t)cg0 * 4);
    const V4 v1 = ld<float, 4>(V0b + (int64_t)prow[j] * C + (int64_t)cg1 * 4);
    if (pok[j]) {
      xlds[prow[j]] = F4{v0.e[0], v0.e[1], v0.e[2], v0.e[3]};
      if constexpr (G == 2) xlds[GS + prow[j]] = F4{v1.e[0], v1.e[1], v1.e[2], v1.e[3]};
    }
    resv[j][0] = v0;
    resv[j][1] = v1;
  }

  auto w_row = [&](int m, int j) { return *reinterpret_cast<const WRow<L>*>(a.W[m] + ((int64_t)b * N + prow[j]) * L); };
  WRow<L> wc = w_row(0, 0);
  __syncthreads();
  for (int m = 0; m < a.M; ++m) {
    const bool store = (a.store_mask >> m) & 1;
    float* __restrict__ om = a.out[m];
    const int mn = m + 1 < a.M ? m + 1 : m;  // (the last step re-requests one of its own rows: no branch in the pipeline)
#pragma unroll
    for (int j = 0; j < R; ++j) {
      const WRow<L> wnx = j + 1 < R ? w_row(m, j + 1) : w_row(mn, 0);
      int p = prow[j];
      if constexpr (R > 2) asm volatile("" : "+v"(p));  // (five rows x L hoisted link addresses would not fit: recomputed per step)
      V4 acc0, acc1;
#pragma unroll
      for (int i = 0; i < 4; ++i) acc0.e[i] = acc1.e[i] = 0.f;
#pragma unroll
      for (int k = 0; k < L; ++k) {
        int src = p + offs.v[k];
        if (src >= N) src -= N;
        const F4 x0 = xlds[cur + src];
        axpy_rn<float, 4>(acc0, wc.e[k], V4{{x0.x, x0.y, x0.z, x0.w}});
        if constexpr (G == 2) {
          const F4 x1 = xlds[cur + GS + src];
          axpy_rn<float, 4>(acc1, wc.e[k], V4{{x1.x, x1.y, x1.z, x1.w}});
        }
      }
      if constexpr (RES) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc0.e[i] = add_rn(acc0.e[i], resv[j][0].e[i]);
          if constexpr (G == 2) acc1.e[i] = add_rn(acc1.e[i], resv[j][1].e[i]);
        }
      }
      if (pok[j]) {
        xlds[(slots - cur) + p] = F4{acc0.e[0], acc0.e[1], acc0.e[2], acc0.e[3]};
        if constexpr (G == 2) xlds[(slots - cur) + GS + p] = F4{acc1.e[0], acc1.e[1], acc1.e[2], acc1.e[3]};
        if (store) {
          st<float, 4>(om + ((int64_t)b * N + p) * C + (int64_t)cg0 * 4, acc0);
          if constexpr (G == 2) {
            if (g1_ok) st<float, 4>(om + ((int64_t)b * N + p) * C + (int64_t)cg1 * 4, acc1);
          }
        }
      }
      wc = wnx;
    }
    __syncthreads();
    cur = slots - cur;
  }
}

}  // namespace psf
