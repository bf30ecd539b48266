// fwd_chain_lds_launch.h — host-side interface of the LDS-resident fused chain kernel (fwd_chain_lds.h).
#pragma once

#include "fwd_chain_lds.h"

namespace psf {

constexpr int kChainLdsLmin = 2, kChainLdsLmax = 20;  // compiled link counts
// N * CC: 2 rows per thread x 1024 threads (4 rows per thread would need > 128 VGPRs at 1024 threads and spill). A few slots
// beyond 2048 — N = 1025 with two channel groups, N = 2049 with one: LRA's CLS-token column makes N = 2^k + 1 — run 3 rows per
// thread on <= 768 threads (170 VGPRs) instead of falling to four more workgroups per sequence or to the per-step kernels.
constexpr int kChainLdsSlots2 = 2048;  // up to here 1 or 2 rows per thread
constexpr int kChainLdsMaxSlots = 2112;
constexpr int kChainLdsMaxBytes = 2 * kChainLdsMaxSlots * 16;  // two X buffers: 66 KiB, two workgroups per CU
// The large instances (chord_chain_rows_k, ONE workgroup per CU, a thread owns whole rows of its channel groups), taken when
// the launch still has >= kChainBigMinWgs workgroups:
//   big = 1: TWO channel groups per workgroup for 1057 <= N <= 2048 (128 KiB). Half as many workgroups stream a sequence's W,
//            and that stream's L2 requests are what bounds the one-launch chain at these lengths (profiles/r06y_bwd_pmc.json:
//            ListOps 124 G requests/s).
//   big = 2: one channel group, five rows per thread, for 2113 <= N <= 4160 (130 KiB): the LRA text task's N = 4096 + 1.
constexpr int kChainBigSlots = 2 * kChainBigRows;
constexpr int kChainBigBytes = 2 * kChainBigSlots * 16;
constexpr int kChainLongBytes = 2 * kChainLongRows * 16;
constexpr int kChainLongRowsPerThread = 5;
constexpr int kChainBigMinWgs = 256;

struct ChainLdsPlan {
  int cc;       // channel groups (of 4 channels) per workgroup: 1 or 2
  int rows;     // rows per thread: 1, 2 or 3
  int threads;  // workgroup size (multiple of 64, <= 1024)
  int chunks;   // workgroups per sequence
  int lds_bytes;
  int big;      // 1 / 2: chord_chain_rows_k with two groups x two rows / one group x five rows per thread
};

// false when the shape does not fit the kernel (N * cc > 4096, L outside 2..20, C not a multiple of 4, ...)
// cc_pref: 0 = automatic (2 channel groups per workgroup when the row count allows; beyond 1056 rows, and the one-group
// instance beyond 2112 rows, when the launch keeps >= kChainBigMinWgs workgroups: B sequences), 1 = force one (and no
// instance beyond 2112 rows), 2 = the large instances wherever they fit
bool plan_chain_lds(int64_t N, int64_t C, int32_t L, int32_t M, ChainLdsPlan* plan, int cc_pref = 0, int64_t B = 0);

hipError_t launch_chain_lds(const ChainLdsPlan& plan, int L, bool res, const ChainArgs& args, const Offsets& offs,
                            int B, hipStream_t stream);

}  // namespace psf
