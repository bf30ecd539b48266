// psf_chord.hip — C ABI (include/psf_chord.h) and kernel dispatch for libpsf_chord.so. gfx950 only.
//
// Host side of the drop-in boundary that replaces torch_sparse.spmm at SyntheticExperiments/psf.py:178-184
// (and its copies) and spmul_cuda.{forward_host,backward_host} (spmul/spmul_cuda.cu:31-59,114-159).
// Stateless apart from a thread-local error string and a few process-wide tuning integers.

#include <atomic>
#include <cstdarg>
#include <cstdio>
#include <cstring>

#include "bwd_kernels.h"
#include "bwd_window_launch.h"
#include "bwd_chain_lds.h"
#include "fwd_chain_lds_launch.h"
#include "fwd_kernels.h"
#include "fwd_mlp_step_launch.h"
#include "fwd_window_launch.h"
#include "mixer_lds_launch.h"
#include "mlp_fwd_x3.h"

using namespace psf;

namespace {

// ------------------------------------------------------------------------------------------------------
// errors, tuning
// ------------------------------------------------------------------------------------------------------
thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

int fail_hip(hipError_t e, const char* what) {
  snprintf(g_err, sizeof(g_err), "%s: %s (hipError_t %d)", what, hipGetErrorString(e), (int)e);
  return (int)e;
}

std::atomic<int> g_fwd_variant{0};  // 0 auto, 1 generic, 2 window
std::atomic<int> g_bwd_variant{0};  // 0 auto, 1 generic
std::atomic<int> g_xcd_remap{1};
// 1 = full tiles on the predicate-free kernel + the ragged last tiles in a second small launch, except on small
// problems (one predicated launch); 2 = always two launches; 0 = always one predicated launch
std::atomic<int> g_fwd_split{1};
// dV workgroup: 0 = auto (512 threads x 1 row for rows of <= 8 channels), 1 = 256 threads x 2 rows
std::atomic<int> g_dv_threads{0};
std::atomic<int> g_bwd_fused{1};  // 0 = never, 1 = auto, 2 = wherever the fused step kernel applies
std::atomic<int> g_dw_variant{0};   // dW: 0 = auto (chunk-looping kernel for C >= 32), 1 = whole-row window kernel, 2 = chunk forced
std::atomic<int> g_dw_tgs{0};       // chunk-looping dW: 0 = auto, 4 = 8 lanes per row chunk, 5 = 16 lanes
std::atomic<int> g_chain_fused{1};  // 1 = short sequences run the whole chain in one LDS-resident launch
std::atomic<int> g_chain_bwd_fused{1};  // psf_chord_chain_bwd_f32: 1 = the one-launch kernel where it fits, 0 = never (PSF_E_UNSUPPORTED)
std::atomic<int> g_chain_cc{0};     // fused chain: 0 = auto channel groups per workgroup, 1 = one, 2 = two wherever it fits
// Rows of >= 64 channels: 0 = one workgroup spans the whole row (default); 1 = 32-channel chunks on 1024-thread
// workgroups (256-row tiles, far links 6 -> 2 at L=12); 2 = 32-channel chunks on 256-thread workgroups.
// r01 sweep (us/launch): cfg3 C=128: 18.9 / 23.2 / 21.8; attention map C=1024: 19.2 / 20.5 / 20.9; C=64: 10.2 / 9.4 /
// 10.0 — contiguous whole-row bursts beat fewer far links, so 0 stays the default.
std::atomic<int> g_fwd_wide{0};
// Forward window kernel, workgroups per CU: 0 = auto (3 for narrow rows on large launches), 1 = no limit, 2..4 = cap
std::atomic<int> g_fwd_wg_limit{0};
std::atomic<int> g_bwd_fused_wg_limit{0};  // fused backward step: 0 = whatever fits (five of 256 threads at C = 8), n = at most n
// Per-step launches of a chain: 1 = alternate the direction in which each XCD walks its tile range
std::atomic<int> g_chain_zigzag{1};
std::atomic<int> g_bwd_fronts{0};     // fused backward step: interleaved fronts per batch element, 0 = auto (2 from N = 8192 on), 1, 2, 4, 8
std::atomic<int> g_fwd_rows{0};       // forward window kernel, rows per thread: 0 = auto, 2, 4 (4: where compiled, fwd_window_launch.h)
std::atomic<int> g_bwd_ablate{0};      // fused backward step, -DPSF_BWD_ABLATE_LAB builds only (bwd_fused.h: ABL); ignored otherwise
std::atomic<int> g_mixer_ablate{0};    // timing experiments on that kernel: bit 0 no MLP arithmetic, 1 no multiply-add chain, 2 no far rows, 3 no data rows
std::atomic<int> g_mixer_lds{1};       // psf_mixer_fwd_*: 1 = short sequences take the single-launch LDS-resident mixer (mixer_lds.h)
std::atomic<int> g_mixer_wg_limit{0};  // step kernel that computes its own W (fwd_mlp_step.h): 0 = whatever fits, n = at most n per CU

}  // namespace
// Fused producer MLPs: 0 = auto (split-bf16 kernel of mlp_fwd_x3.hip where it applies, else the f32-MFMA kernel of
// mlp_fwd.hip with all images LDS-resident when they fit), 1 = f32 MFMA streaming, 2 = f32 MFMA resident, 3 = split-bf16
std::atomic<int> psf_g_mlp_variant{0};
// Fused producer MLP backward: 0 = auto (= 3), 1 = all-f32-MFMA kernel, 2 = split-bf16 kernel with steps 4 and 5 left on
// the f32 instruction, 3 = split-bf16 kernel on dual-use LDS planes (all five GEMMs on the bf16 matrix pipe)
std::atomic<int> psf_g_wide_fuse{1};  // wide producer MLPs: second layers of narrow-output MLPs in the forward GEMM's epilogue
namespace {

struct Knob {
  const char* key;
  std::atomic<int>* var;
  int lo, hi;
};
Knob g_knobs[] = {
    {"fwd_variant", &g_fwd_variant, 0, 2}, {"bwd_variant", &g_bwd_variant, 0, 1}, {"xcd_remap", &g_xcd_remap, 0, 1},
    {"fwd_split", &g_fwd_split, 0, 2},
    {"fwd_wide", &g_fwd_wide, 0, 4},
    {"dw_variant", &g_dw_variant, 0, 2},
    {"dv_threads", &g_dv_threads, 0, 1},
    {"bwd_fused", &g_bwd_fused, 0, 2},
    {"bwd_fused_wg_limit", &g_bwd_fused_wg_limit, 0, 5},
    {"dw_tgs", &g_dw_tgs, 0, 5},
    {"fwd_wg_limit", &g_fwd_wg_limit, 0, 4},
    {"chain_zigzag", &g_chain_zigzag, 0, 1},
    {"mixer_wg_limit", &g_mixer_wg_limit, 0, 4},
    {"mixer_lds", &g_mixer_lds, 0, 1},
    {"mixer_ablate", &g_mixer_ablate, 0, 15},
    {"bwd_ablate", &g_bwd_ablate, 0, 1023},
    {"bwd_fronts", &g_bwd_fronts, 0, 8},
    {"fwd_rows", &g_fwd_rows, 0, 4},
    {"chain_fused", &g_chain_fused, 0, 2},
    {"chain_cc", &g_chain_cc, 0, 2},
    {"chain_bwd_fused", &g_chain_bwd_fused, 0, 1},
    {"mlp_variant", &psf_g_mlp_variant, 0, 3},
    {"wide_fuse", &psf_g_wide_fuse, 0, 1},
};

// One consistent view of the knobs per entry-point call: every extern "C" function takes ONE snapshot and hands it down, so
// a psf_set_tuning from another thread changes the next call, never the middle of one; `walk_backwards` (zigzag of a chain's
// odd steps) travels in it too instead of in thread-local state.
struct Tuning {
  int fwd_variant, bwd_variant, xcd_remap, fwd_split, dv_threads, bwd_fused, dw_variant,
      dw_tgs, chain_fused, chain_cc, fwd_wide, fwd_wg_limit, bwd_fused_wg_limit, chain_zigzag, mixer_wg_limit, mixer_ablate, mixer_lds, bwd_ablate, bwd_fronts, fwd_rows;
  bool walk_backwards;
};

Tuning snapshot() {
  Tuning t;
  t.fwd_variant = g_fwd_variant.load(), t.bwd_variant = g_bwd_variant.load(), t.xcd_remap = g_xcd_remap.load();
  t.fwd_split = g_fwd_split.load();
  t.dv_threads = g_dv_threads.load(), t.bwd_fused = g_bwd_fused.load();
  t.dw_variant = g_dw_variant.load(), t.dw_tgs = g_dw_tgs.load(), t.chain_fused = g_chain_fused.load();
  t.chain_cc = g_chain_cc.load(), t.fwd_wide = g_fwd_wide.load(), t.fwd_wg_limit = g_fwd_wg_limit.load();
  t.bwd_fused_wg_limit = g_bwd_fused_wg_limit.load(), t.chain_zigzag = g_chain_zigzag.load();
  t.mixer_wg_limit = g_mixer_wg_limit.load(), t.mixer_ablate = g_mixer_ablate.load(), t.mixer_lds = g_mixer_lds.load();
  t.bwd_ablate = g_bwd_ablate.load();
  t.bwd_fronts = g_bwd_fronts.load();
  t.fwd_rows = g_fwd_rows.load();
  t.walk_backwards = false;
  return t;
}

int ceil_log2(int64_t x) {
  int s = 0;
  while (((int64_t)1 << s) < x) ++s;
  return s;
}

// Reduce the caller's offsets (or the chord pattern) into [0, N).
void make_offsets(int64_t N, int32_t L, const int64_t* offsets, Offsets* out) {
  for (int k = 0; k < L; ++k) {
    int64_t o;
    if (offsets != nullptr) {
      o = offsets[k] % N;
      if (o < 0) o += N;
    } else if (k == 0) {
      o = 0;
    } else if (k - 1 < 62) {
      o = ((int64_t)1 << (k - 1)) % N;
    } else {  // 2^(k-1) does not fit in int64: (2^62 mod N) * 2 mod N
      int64_t t = ((int64_t)1 << 62) % N;
      for (int i = 62; i < k - 1; ++i) t = (t * 2) % N;
      o = t;
    }
    out->v[k] = (int32_t)o;
  }
  for (int k = L; k < PSF_MAX_LINKS; ++k) out->v[k] = 0;
}

int check_dims(int64_t B, int64_t N, int32_t L, int64_t C, int64_t v_batch_stride) {
  if (B < 0 || N < 1 || L < 1 || C < 1)
    return fail(PSF_E_SHAPE, "need B >= 0, N >= 1, L >= 1, C >= 1 (got B=%lld N=%lld L=%d C=%lld)", (long long)B,
                (long long)N, (int)L, (long long)C);
  if (L > PSF_MAX_LINKS) return fail(PSF_E_SHAPE, "L=%d exceeds PSF_MAX_LINKS=%d", (int)L, PSF_MAX_LINKS);
  if (N > (int64_t)1 << 30) return fail(PSF_E_SHAPE, "N=%lld exceeds 2^30", (long long)N);
  if (C > (int64_t)1 << 30 || N * C > (int64_t)1 << 40) return fail(PSF_E_SHAPE, "N*C too large");
  if (v_batch_stride != 0 && v_batch_stride != N * C)
    return fail(PSF_E_SHAPE, "v_batch_stride must be 0 (broadcast) or N*C=%lld, got %lld", (long long)(N * C),
                (long long)v_batch_stride);
  return PSF_OK;
}

bool aligned_to(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

// Geometry of a launch over row tiles [tile0, tile0 + tiles) of every batch element.
int make_geom(const Tuning& tn, int64_t B, int64_t N, int32_t L, int64_t C, int vec, int tg_shift, int TR, bool split_channels,
              int64_t v_bstride, int tile0, int tiles, Geom* gm) {
  gm->N = (int32_t)N;
  gm->L = L;
  gm->C = (int32_t)C;
  gm->CG = (int32_t)((C + vec - 1) / vec);
  gm->tg_shift = tg_shift;
  gm->TR = TR;
  gm->tiles_n = tiles;
  gm->tile0 = tile0;
  const int TG = 1 << tg_shift;
  gm->chunks_c = split_channels ? (gm->CG + TG - 1) / TG : 1;
  gm->per_b = gm->tiles_n * gm->chunks_c;
  const int64_t nb = B * (int64_t)gm->per_b;
  if (nb > (int64_t)0x7fffffff)
    return fail(PSF_E_SHAPE, "launch of %lld workgroups exceeds the grid limit", (long long)nb);
  gm->nblocks = (uint32_t)nb;
  gm->per_b_inv = udiv_inv_of((uint32_t)gm->per_b);
  gm->chunks_inv = udiv_inv_of((uint32_t)gm->chunks_c);
  gm->xq = gm->nblocks / kXcds;
  gm->xr = gm->nblocks % kXcds;
  gm->remap = tn.xcd_remap ? (tn.walk_backwards && tn.chain_zigzag ? 2 : 1) : 0;
  gm->aligned = 0;  // window_launches sets it from the pick
  gm->ileave = 0;
  gm->v_bstride = v_bstride;
  return PSF_OK;
}

int generic_geom(const Tuning& tn, int64_t B, int64_t N, int32_t L, int64_t C, int vec, bool split_channels, int64_t v_bstride,
                 Geom* gm) {
  const int64_t CG = (C + vec - 1) / vec;
  const int tgs = ceil_log2(CG) > 6 ? 6 : ceil_log2(CG);
  const int TR = kBlock >> tgs;
  return make_geom(tn, B, N, L, C, vec, tgs, TR, split_channels, v_bstride, 0, (int)((N + TR - 1) / TR), gm);
}

// ------------------------------------------------------------------------------------------------------
// window-kernel selection
// ------------------------------------------------------------------------------------------------------
struct WinPick {
  int tgs, rows, nt, TR, KN;
  int tiles_full;  // row tiles per sequence with all TR rows < N
  bool ragged;     // N % TR != 0: one more, partial, tile per sequence
  bool all_edge;   // every tile must take the EDGE kernel (channel groups not a multiple of TG, or W not chunk-clean)
  bool aligned = false;  // Geom::aligned
};

#define PSF_TGS_SWITCH(FN, ARGS)                 \
  switch (pk.tgs) {                              \
    case 0: return FN<0, 256> ARGS;              \
    case 1: return FN<1, 256> ARGS;              \
    case 2: return FN<2, 256> ARGS;              \
    case 3: return FN<3, 256> ARGS;              \
    case 4: return FN<4, 256> ARGS;              \
    case 5: return FN<5, 256> ARGS;              \
    case 6: return FN<6, 256> ARGS;              \
    default: return hipErrorInvalidValue;        \
  }

hipError_t launch_win(const WinPick& pk, int L, const FwdWinArgs& a) {
  if (pk.nt == kWideThreads) return pk.tgs == kWideTgs ? launch_fwd_win<kWideTgs, kWideThreads>(pk.rows, L, a) : hipErrorInvalidValue;
  PSF_TGS_SWITCH(launch_fwd_win, (pk.rows, L, a))
}

hipError_t launch_dv(const WinPick& pk, int L, const BwdWinArgs& a) {
  if (pk.nt == kWideThreads) return pk.tgs == kWideTgs ? launch_dv_win<kWideTgs, kWideThreads>(pk.rows, L, a) : hipErrorInvalidValue;
  if (pk.nt == kDvMidThreads) {
    switch (pk.tgs) {
      case 0: return launch_dv_win<0, kDvMidThreads>(pk.rows, L, a);
      case 1: return launch_dv_win<1, kDvMidThreads>(pk.rows, L, a);
      default: return hipErrorInvalidValue;
    }
  }
  PSF_TGS_SWITCH(launch_dv_win, (pk.rows, L, a))
}
#undef PSF_TGS_SWITCH

hipError_t launch_fused_step(int tgs, int L, const BwdWinArgs& a) {
  switch (tgs) {
    case 0: return launch_bwd_fused<0>(L, a);
    case 1: return launch_bwd_fused<1>(L, a);
    case 2: return launch_bwd_fused<2>(L, a);
    case 3: return launch_bwd_fused<3>(L, a);
    case 4: return launch_bwd_fused<4>(L, a);
    case 5: return launch_bwd_fused<5>(L, a);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_fused_edge_step(int tgs, int L, const BwdWinArgs& a) {
  switch (tgs) {
    case 0: return launch_bwd_fused_edge<0>(L, a);
    case 1: return launch_bwd_fused_edge<1>(L, a);
    case 2: return launch_bwd_fused_edge<2>(L, a);
    case 3: return launch_bwd_fused_edge<3>(L, a);
    case 4: return launch_bwd_fused_edge<4>(L, a);
    case 5: return launch_bwd_fused_edge<5>(L, a);
    default: return hipErrorInvalidValue;
  }
}

// Row widths the fused backward step takes: exactly 4 << tgs channels with a whole row inside one workgroup. Rounds 3-5: up
// to 32 channels. Round 6 (profiles/r06p_bwd_fused_wide*.log, us per step, two kernels / fused, operands rotating, dZ chained):
// 64 channels at every length — N = 1024: 11.3 / 8.9; 2000: 20.2 / 15.3; 2048: 21.6 / 17.4; 2049 (edge instance): 23.6 / 18.8;
// 4096: 42.8 / 38.3; 4097: 23.8 / 18.7; 8192: 43.7 / 42.9; 16384: 49.3 / 47.2 — and 128 channels up to N = 4096 — ListOps'
// N = 2000: 39.3 / 35.4; 2001: 40.1 / 36.9; 1024, 4096, 4097 equal — but not beyond (N = 16384: 96.3 / 102.8: tiles of 8 rows).
bool fused_step_width(int64_t C, int64_t N) {
  return C == 4 || C == 8 || C == 16 || C == 32 || C == 64 || (C == 128 && N <= 4096);
}

// The EDGE instance of the fused step (bwd_fused.h: chord_bwd_fused_edge_k) takes what pick_fused_step turns away for its
// geometry: any N >= two tiles (N = 2^k + 1 with a CLS token), any far offsets, W / dW at any alignment. Same rows (fused_step_width), chord near offsets, 16-byte aligned row operands. Knob bwd_fused = 2 keeps its meaning (the aligned instance or
// nothing); 1 (default) lets this one in.
bool pick_fused_edge_step(const Tuning& tn, const void* dZ, const void* V, const void* dV, int64_t N, int32_t L, int64_t C,
                          int64_t v_bstride, const Offsets& offs, WinPick* pk) {
  if (tn.bwd_fused != 1 || L < kWinLmin || L > kWinLmax || !fused_step_width(C, N)) return false;
  const int tgs = C == 4 ? 0 : C == 8 ? 1 : C == 16 ? 2 : C == 32 ? 3 : C == 64 ? 4 : 5;
  const int nt = 256;
  const int TR = nt >> tgs;
  if (N < 2 * (int64_t)TR) return false;
  if (!aligned_to(dZ, 16) || !aligned_to(V, 16) || !aligned_to(dV, 16)) return false;
  if (v_bstride != 0 && v_bstride != N * C) return false;
  int KN = 2;
  for (int t = TR; t > 1; t >>= 1) ++KN;
  if (KN > L) KN = L;
  for (int k = 0; k < KN; ++k)
    if (offs.v[k] != chord_off(k)) return false;
  pk->tgs = tgs, pk->rows = 1, pk->nt = nt, pk->TR = TR, pk->KN = KN;
  pk->tiles_full = (int)(N / TR), pk->ragged = (N % TR) != 0, pk->all_edge = true;
  return true;
}

// The fused dV + dW step (bwd_fused.h) applies to full tiles of rows of exactly 4 << tgs channels (fused_step_width), N a multiple of
// the tile (256 >> tgs rows) and at least two tiles, chord near offsets, everything 16-byte aligned and chunk-clean.
bool pick_fused_step(const Tuning& tn, const void* dZ, const void* W, const void* V, const void* dW, const void* dV, int64_t B, int64_t N,
                     int32_t L, int64_t C, int64_t v_bstride, const Offsets& offs, WinPick* pk) {
  const int knob = tn.bwd_fused;
  if (!knob || L < kWinLmin || L > kWinLmax || !fused_step_width(C, N)) return false;
  const int tgs = C == 4 ? 0 : C == 8 ? 1 : C == 16 ? 2 : C == 32 ? 3 : C == 64 ? 4 : 5;
  const int nt = 256;
  const int TR = nt >> tgs;
  if (N % TR != 0 || N < 2 * (int64_t)TR) return false;
  if (!aligned_to(dZ, 16) || !aligned_to(W, 16) || !aligned_to(V, 16) || !aligned_to(dW, 16) || !aligned_to(dV, 16)) return false;
  if ((B * N * (int64_t)L) % 4 != 0 || (v_bstride != 0 && v_bstride != N * C)) return false;
  int KN = 2;
  for (int t = TR; t > 1; t >>= 1) ++KN;
  if (KN > L) KN = L;
  for (int k = 0; k < KN; ++k)
    if (offs.v[k] != chord_off(k)) return false;
  for (int k = KN; k < L; ++k)
    if (offs.v[k] % TR != 0) return false;  // far row blocks are TR-aligned (scalar block addresses in the kernel)
  if (N * C * 4 >= ((int64_t)1 << 31)) return false;
  pk->tgs = tgs, pk->rows = 1, pk->nt = nt, pk->TR = TR, pk->KN = KN;
  pk->tiles_full = (int)(N / TR), pk->ragged = false, pk->all_edge = false;
  return true;
}

// A ragged last tile per sequence (N % TR != 0) runs on the EDGE instance. In a second launch of its own it costs a
// kernel boundary, ~2.7 us whatever the shape (r02d: IMDb N = 4097, C = 32: 15.3 us split vs 12.7 in one predicated
// launch; N = 2000, C = 16: 7.7 vs 5.0; dW at N = 2000, C = 128: 21.9 vs 16.1); predicating EVERY tile costs 0-5 % of the
// launch (r01c: cfg2 27.6 -> 28.9 us). So: one predicated launch unless the launch is long enough for 5 % to exceed the
// boundary, i.e. beyond ~300 MB of algorithmic bytes.
bool ragged_in_one_launch(const Tuning& tn, bool ragged, int64_t B, int64_t N, int32_t L, int64_t C) {
  return ragged && tn.fwd_split == 1 && 4 * B * N * (L + 3 * C) <= (int64_t)300 * 1000 * 1000;
}

hipError_t launch_dwc(const WinPick& pk, int L, const BwdWinArgs& a) {
  switch (pk.tgs) {
    case 3: return launch_dw_chunk<3>(L, a);
    case 4: return launch_dw_chunk<4>(L, a);
    default: return hipErrorInvalidValue;
  }
}

// Chunk-looping dW (bwd_dw_chunk.h), rows of >= 32 channels whose channel groups split into chunks of 8 (or 16) lanes:
// 256 threads x 1 row, so tiles of 32 (16) rows. Fills `pick` when the kernel applies.
bool pick_dw_chunk(const Tuning& tn, const void* dW, int64_t B, int64_t N, int32_t L, int64_t C, const Offsets& offs, bool vec_ok,
                   WinPick* pick) {
  if (!vec_ok || L < kWinLmin || L > kWinLmax) return false;
  const int64_t CG = C / 4;
  if (CG % 8 != 0 || CG / 8 > 4096) return false;
  int tgs = 3;
  const int knob = tn.dw_tgs;
  // 16 lanes per row chunk (16-row tiles) when that spares the launch its ragged last tile (ListOps: N = 2000 = 125 * 16)
  if (knob == 5 || (knob == 0 && CG % 16 == 0 && N % 32 != 0 && N % 16 == 0)) tgs = CG % 16 == 0 ? 4 : 3;
  const int TR = win_tile_rows(tgs, 1, 256);
  if (N < 2 * (int64_t)TR) return false;
  int KN = 2;
  for (int t = TR; t > 1; t >>= 1) ++KN;
  if (KN > L) KN = L;
  for (int k = 0; k < KN; ++k)
    if (offs.v[k] != chord_off(k)) return false;  // near offsets are compile-time constants in this kernel
  if (N * C >= ((int64_t)1 << 31)) return false;                 // 32-bit element offsets inside a batch element
  pick->tgs = tgs;
  pick->rows = 1;
  pick->nt = 256;
  pick->TR = TR;
  pick->KN = KN;
  pick->tiles_full = (int)(N / TR);
  pick->ragged = (N % TR) != 0;
  pick->all_edge = ragged_in_one_launch(tn, pick->ragged, B, N, L, C) || !aligned_to(dW, 16) || ((N * (int64_t)L) % 4) != 0 || !tn.fwd_split;
  return true;
}

hipError_t launch_dw(const WinPick& pk, int L, const BwdWinArgs& a) {
  switch (pk.tgs) {
    case 0: return launch_dw_win<0>(pk.rows, L, a);
    case 1: return launch_dw_win<1>(pk.rows, L, a);
    case 2: return launch_dw_win<2>(pk.rows, L, a);
    case 3: return launch_dw_win<3>(pk.rows, L, a);
    case 4: return launch_dw_win<4>(pk.rows, L, a);
    case 5: return launch_dw_win<5>(pk.rows, L, a);
    case 6: return launch_dw_win<6>(pk.rows, L, a);
    default: return hipErrorInvalidValue;
  }
}

// Decide whether a window kernel applies (f32, vectorisable, chord-like near links); fills pick on success.
// `W` is the flat [B,N,L] array the kernel copies in 16-byte chunks (W itself, or dW for the dW kernel).
// `chunk_channels`: the kernel may split a row's channels over several workgroups (forward, dV) — then wide rows
// (C >= 64) use the wide-row configuration: 32-channel chunks, 1024 threads, 256-row tiles.
bool pick_window(const Tuning& tn, const void* W, int64_t B, int64_t N, int32_t L, int64_t C, const Offsets& offs, bool vec_ok,
                 WinPick* pick, int rows_pref, bool chunk_channels, int nt_pref = 0, bool forward = false) {
  if (!vec_ok || L < kWinLmin || L > kWinLmax) return false;
  const int64_t CG = C / 4;
  int tgs = ceil_log2(CG) > kWinTgsMax ? kWinTgsMax : ceil_log2(CG);
  int nt = 256;
  int rows = rows_pref;  // 2 (forward, dV) or 1 (dW): the compiled rows per thread (fwd_window_launch.h)
  const int wide = tn.fwd_wide;
  // Forward, rows of 64..256 channels, sequences up to 4096: 32-channel chunks on 1024-thread workgroups (256-row tiles: two
  // far links at L = 12 instead of five to seven). With the scalar block addresses of round 4 they beat the whole-row tiles that
  // rounds 1-3 measured faster: N = 2048, B = 32: C = 64 10.7 -> 9.0 us per step, C = 96 16.7 -> 14.3, C = 128 19.5 -> 18.3,
  // C = 192 29.9 -> 25.7, C = 256 36.1 -> 33.9; N = 4096: +2..5 %; C = 512: equal; N = 16384, C = 64, B = 8: 23.7 -> 25.7 (slower)
  // (profiles/r04ak_fwd_wide_rule_sweep.log). The backward kernels keep their configuration.
  // Round 5, with operands rotating beyond the Infinity Cache as a training step has them (profiles/r05m_fwd_wide_mid.log, us per
  // step, chunks / whole rows): N = 2048, C = 64: 10.5 / 11.8; N = 2048, C = 128: 20.8 / 22.0; N = 4096, C = 64: 21.7 / 22.6; C = 256:
  // 21.2 / 21.9 — but ListOps' N = 2000, C = 128: 23.9 / 21.1: 2000 is no multiple of the 256-row chunk tile (every tile then takes the
  // per-lane request form) and a multiple of the whole-row tile. So: chunks only where their tiles divide N or the whole-row tiles do not.
  const int64_t tr_whole = win_tile_rows(tgs, rows, 256), tr_chunk = win_tile_rows(kWideTgs, rows, kWideThreads);
  const bool auto_wide = forward && wide == 0 && CG >= 16 && CG <= 64 && N <= 4096 && (N % tr_chunk == 0 || N % tr_whole != 0);
  if (chunk_channels && (wide == 1 || auto_wide) && CG >= 16 && N >= 2 * (int64_t)win_tile_rows(kWideTgs, rows, kWideThreads)) {
    tgs = kWideTgs;  // 32-channel chunks on 1024-thread workgroups
    nt = kWideThreads;
  } else if (chunk_channels && wide == 2 && CG >= 16) {
    tgs = kWideTgs;  // 32-channel chunks on 256-thread workgroups
  } else if (nt_pref == kDvMidThreads && tgs <= kDvMidTgsMax && N >= 2 * (int64_t)win_tile_rows(tgs, 1, kDvMidThreads)) {
    nt = kDvMidThreads;  // dV: 512 threads x 1 row
    rows = 1;
  }
  // Forward, rows of 16..64 channels: four rows per thread (fwd_window_launch.h: win_rows4_compiled) where the four-row tile
  // divides N, i.e. where its launches take the aligned request form; the per-lane form of other lengths (2^k + 1: LRA's
  // CLS-token column) is faster on the smaller tile (N = 4097 x 32: 11.3 / 12.1 us, N = 1025: 6.9 / 7.2, two rows / four:
  // profiles/r06k_fwd_rows_product.log). From N = 4096 on: below that the two forms are within 3 % of each other and the sign
  // depends on how the step is driven (Pathfinder's shape, N = 1024 x 32: 6.32 / 6.13 us per step inside a chain, but 5.82 /
  // 6.39 us per launch for the same step launched alone again and again under rocprofv3: r06m_fwd_rows_resident.log,
  // r06z_bwd_summary.md of both collections).
  if (forward && win_rows4_compiled(tgs, nt) && tn.fwd_rows != 2 && N >= 2 * (int64_t)win_tile_rows(tgs, 4, nt)) {
    if (tn.fwd_rows == 4 || (N >= 4096 && N % win_tile_rows(tgs, 4, nt) == 0)) rows = 4;
  }
  const int TR = win_tile_rows(tgs, rows, nt);
  if (N < 2 * (int64_t)TR) return false;  // the window may wrap at most once
  int KN = 2;                             // offsets 0, 1, 2, ..., 2^(KN-2) <= TR
  for (int t = TR; t > 1; t >>= 1) ++KN;
  if (KN > L) KN = L;
  for (int k = 0; k < KN; ++k)
    if (offs.v[k] != chord_off(k)) return false;  // near offsets are compile-time constants in the window kernels
  pick->tgs = tgs;
  pick->rows = rows;
  pick->nt = nt;
  pick->TR = TR;
  pick->KN = KN;
  pick->tiles_full = (int)(N / TR);
  pick->ragged = (N % TR) != 0;
  pick->aligned = (N % TR) == 0 && N * C * 4 < ((int64_t)1 << 31);
  for (int k = KN; k < L; ++k)
    if (offs.v[k] % TR != 0) pick->aligned = false;
  const int TG = 1 << tgs;
  pick->all_edge = (CG % TG) != 0 || !aligned_to(W, 16) || ((B * N * (int64_t)L) % 4) != 0 || !tn.fwd_split ||
                   ragged_in_one_launch(tn, pick->ragged, B, N, L, C);
  return true;
}

// The dV window kernel's configuration for a shape.
// default rows per thread (r01 sweep, us at cfg2): dV R=2 31.3 vs R=1 32.7; 512 threads x 1 row per thread instead of
// 256 x 2 is the same tile at C <= 8 (r02 lab 28.65 vs 29.05 us at cfg2)
bool pick_dv(const Tuning& tn, const void* W, int64_t B, int64_t N, int32_t L, int64_t C, const Offsets& offs, bool vec_ok, WinPick* pk) {
  const int dvt = tn.dv_threads;
  const int nt_dv = (dvt == 0 && C <= 8) ? kDvMidThreads : 0;
  return pick_window(tn, W, B, N, L, C, offs, vec_ok, pk, 2, true, nt_dv);
}

// ------------------------------------------------------------------------------------------------------
// typed entry points
// ------------------------------------------------------------------------------------------------------
// Issue the one to two launches of a window kernel: full tiles on the predicate-free instance, the ragged last
// tile of every sequence (if any) on the EDGE instance; everything on the EDGE instance when `all_edge`.
// `launch` reads *gm and *edge, which are filled in before each call.
template <typename F>
int window_launches(const Tuning& tn, const WinPick& pk, bool all_edge, int64_t B, int64_t N, int32_t L, int64_t C,
                    int64_t v_bstride, bool split_channels, Geom* gm, bool* edge, F launch, const char* what) {
  const int tiles_all = pk.tiles_full + (pk.ragged ? 1 : 0);
  struct Part {
    int tile0, tiles;
    bool edge;
  };
  Part parts[2];
  int np = 0;
  if (all_edge) {
    parts[np++] = {0, tiles_all, true};
  } else {
    if (pk.tiles_full > 0) parts[np++] = {0, pk.tiles_full, false};
    if (pk.ragged) parts[np++] = {pk.tiles_full, 1, true};
  }
  for (int i = 0; i < np; ++i) {
    if (int rc = make_geom(tn, B, N, L, C, 4, pk.tgs, pk.TR, split_channels, v_bstride, parts[i].tile0, parts[i].tiles, gm))
      return rc;
    gm->aligned = pk.aligned && !parts[i].edge;
    *edge = parts[i].edge;
    hipError_t e = launch();
    if (e != hipSuccess) return fail_hip(e, what);
  }
  return PSF_OK;
}

int fwd_window_f32(const Tuning& tn, const WinPick& pk, const float* W, const float* V, const float* res, float* out, int64_t B,
                   int64_t N, int32_t L, int64_t C, int64_t v_batch_stride, const Offsets& offs, hipStream_t s) {
  FwdWinArgs a;
  a.W = W;
  a.V = V;
  a.res = res;
  a.out = out;
  a.offs = offs;
  a.w_total = B * N * (int64_t)L;
  a.stream = s;
  // Workgroups per CU. Measured (profiles/r01e_fwd_wg_per_cu.log, us per launch, 4 / 3 per CU): cfg2 (C = 8, 4096
  // tiles) 27.6 / 27.0; the same at B = 40 (2560 tiles) 18.9 / 18.8; C = 32, B = 16: 24.5 / 24.8; N = 4096, C = 16:
  // 8.9 / 9.2; 2 per CU: 30.0 at cfg2. Round 3, chains that keep every step's output (training; N = 16384, C = 8, no limit /
  // three per CU, profiles/r03al_fwd_wg_limit_sweep.log): B = 16 (1024 tiles) 9.4 / 9.7; B = 24 13.5 / 13.2; B = 32 16.5 /
  // 16.1; B = 40 19.9 / 19.0; B = 48 22.8 / 21.8. So: three for narrow rows on launches of >= 1536 tiles, no limit otherwise.
  const int knob = tn.fwd_wg_limit;
  const int64_t tiles_total = B * (int64_t)(pk.tiles_full + (pk.ragged ? 1 : 0));
  // Rows of 32 channels on launches of >= 8192 tiles (round 4, profiles/r04ai_fwd_mid_sweep.log, N = 16384, B = 64): 102.9 / 97.4;
  // at B = 16 (4096 tiles) 22.6 / 22.7, N = 4096, B = 32: 11.2 / 11.6 — so three there too, from 8192 tiles on.
  // Rows of 16 channels (same sweep script, us per step, what fits / three): 8192 tiles (N = 16384, B = 64) 45.9 / 44.3; 2048
  // tiles (B = 16) 13.1 / 12.8; 1024 tiles (N = 4096, B = 32) 7.1 / 7.8: three from 2048 tiles on.
  const bool three = pk.nt == 256 && ((pk.tgs <= 1 && tiles_total >= 1536) || (pk.tgs == 2 && tiles_total >= 2048) ||
                                      (pk.tgs == 3 && tiles_total >= 8192));
  a.wg_per_cu = knob == 0 ? (three ? 3 : 0) : (knob == 1 ? 0 : knob);
  return window_launches(tn, pk, pk.all_edge, B, N, L, C, v_batch_stride, true, &a.gm, &a.edge,
                         [&] { return launch_win(pk, L, a); }, "chord_fwd_win launch");
}

template <typename T>
int fwd_impl(const Tuning& tn, const T* W, const T* V, const T* res, T* out, int64_t B, int64_t N, int32_t L, int64_t C,
             int64_t v_batch_stride, const int64_t* offsets, void* stream) {
  if (int rc = check_dims(B, N, L, C, v_batch_stride)) return rc;
  if (B == 0) return PSF_OK;
  if (!W || !V || !out) return fail(PSF_E_NULL, "W, V and out must be non-NULL");
  if (out == V) return fail(PSF_E_ALIAS, "out must not alias V (rows are gathered from other rows)");
  if (!aligned_to(W, sizeof(T)) || !aligned_to(V, sizeof(T)) || !aligned_to(out, sizeof(T)) ||
      (res && !aligned_to(res, sizeof(T))))
    return fail(PSF_E_ALIGN, "pointers must be aligned to the element size");
  Offsets offs;
  make_offsets(N, L, offsets, &offs);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);

  constexpr int VECW = 16 / (int)sizeof(T);
  const bool vec_ok =
      (C % VECW == 0) && aligned_to(V, 16) && aligned_to(out, 16) && (!res || aligned_to(res, 16));

  const int variant = tn.fwd_variant;
  if constexpr (sizeof(T) == 4) {  // the window kernels are compiled for f32 only (f64 exists for gradcheck)
    WinPick pk;
    if (variant != 1 && pick_window(tn, W, B, N, L, C, offs, vec_ok, &pk, 2, true, 0, true))
      return fwd_window_f32(tn, pk, W, V, res, out, B, N, L, C, v_batch_stride, offs, s);
  }
  if (variant == 2)
    return fail(PSF_E_TUNING, "fwd_variant=2 forced but the window kernel does not apply to N=%lld L=%d C=%lld",
                (long long)N, (int)L, (long long)C);

  Geom gm;
  if (int rc = generic_geom(tn, B, N, L, C, vec_ok ? VECW : 1, true, v_batch_stride, &gm)) return rc;
  if (vec_ok)
    hipLaunchKernelGGL((chord_fwd_generic_k<T, VECW>), dim3(gm.nblocks), dim3(kBlock), 0, s, W, V, res, out, gm, offs);
  else
    hipLaunchKernelGGL((chord_fwd_generic_k<T, 1>), dim3(gm.nblocks), dim3(kBlock), 0, s, W, V, res, out, gm, offs);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail_hip(e, "chord_fwd_generic launch");
  return PSF_OK;
}

template <typename T>
int bwd_impl(const Tuning& tn, const T* dZ, const T* W, const T* V, T* dW, T* dV, int64_t B, int64_t N, int32_t L, int64_t C,
             int64_t v_batch_stride, const int64_t* offsets, void* stream) {
  if (int rc = check_dims(B, N, L, C, v_batch_stride)) return rc;
  if (B == 0) return PSF_OK;
  if (!dZ) return fail(PSF_E_NULL, "dZ must be non-NULL");
  if (dV && !W) return fail(PSF_E_NULL, "dV requested but W is NULL");
  if (dW && !V) return fail(PSF_E_NULL, "dW requested but V is NULL");
  if (dV && dV == dZ) return fail(PSF_E_ALIAS, "dV must not alias dZ");
  if (!aligned_to(dZ, sizeof(T)) || (W && !aligned_to(W, sizeof(T))) || (V && !aligned_to(V, sizeof(T))) ||
      (dW && !aligned_to(dW, sizeof(T))) || (dV && !aligned_to(dV, sizeof(T))))
    return fail(PSF_E_ALIGN, "pointers must be aligned to the element size");
  Offsets offs;
  make_offsets(N, L, offsets, &offs);
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  constexpr int VECW = 16 / (int)sizeof(T);

  if constexpr (sizeof(T) == 4) {  // LDS-window kernels (f32). Whatever they handle is cleared below.
    if (tn.bwd_variant != 1) {
      const int64_t w_total = B * N * (int64_t)L;
      const int TGmax = 1 << kWinTgsMax;
      WinPick pk;
      // default rows per thread (r01 sweep, us at cfg2): dV R=2 31.3 vs R=1 32.7; dW R=1 22.9 vs R=2 28.7
      const int rows_dw = 1;
      // dW before dV: dV's output is the next (earlier) step's dZ, read first thing by that step's kernels; writing
      // it last leaves it cache-hot (dV 27.4 -> 26.9 us, dW 20.5 -> 20.4 us in the Order training step)
      const int dwv = tn.dw_variant;
      if (dW && dV && pick_fused_step(tn, dZ, W, V, dW, dV, B, N, L, C, v_batch_stride, offs, &pk)) {
        BwdWinArgs a{dZ, W, dV, Geom{}, offs, w_total, false, s};
        a.V2 = V;
        a.out2 = dW;
        // Workgroups per CU, re-measured on the round-4 kernel with rotating operands (profiles/r04am_bwd_fused_wg_sweep.log, us per
        // step, what fits / three): 5120 tiles (N = 16384, C = 8, B = 40) 42.3 / 40.9; 8192 tiles (C = 32, B = 16) 51.0 / 49.6; 4096
        // tiles (N = 4096, C = 16, B = 64) 26.6 / 25.4; 2048 tiles 12.6 / 13.5 and 14.7 / 15.0: three from 4096 tiles on.
        // Rows of 64 / 128 channels (round 6, tiles of 16 / 8 rows; profiles/r06r_bwd_rows_wide.log, what fits / three / four):
        // N = 2048 x 64, B = 32: 17.4 / 17.3 / 15.9; N = 4096 x 64, B = 16: 17.4 / 17.3 / 16.3; N = 2000 x 128: 35.5 / 35.6 / 34.8;
        // N = 16384 x 64: 47.0 / 47.1 / 47.5 — four from 4096 tiles on.
        a.wg_per_cu = tn.bwd_fused_wg_limit ? tn.bwd_fused_wg_limit
                                            : (B * (int64_t)pk.tiles_full >= 4096 ? (pk.tgs >= 4 ? 4 : 3) : 0);
        a.ablate = tn.bwd_ablate;
        if (int rc = make_geom(tn, B, N, L, C, 4, pk.tgs, pk.TR, false, v_batch_stride, 0, pk.tiles_full, &a.gm)) return rc;
        // Two interleaved fronts per batch element (Geom::ileave, bwd_fused.h): tile t of the XCD's walk is row block
        // (t mod 2) tiles / 2 + t / 2, so the rows N / 2 apart that the longest link joins are in flight together. Round 6,
        // operands rotating as in the chain's backward (profiles/r06c_bwd_ileave2.log, us per step, one front / two): Order
        // shape (N = 16384, C = 8, B = 40) 41.1 / 39.2, N = 4096 x 32 channels 22.1 / 21.6, genome (N = 16384 x 32) 46.5 / 47.0
        // (noise); four and eight fronts equal two. In the training steps (r06c_step_ileave.log): Order 2.100 -> 2.074 ms,
        // genome 1.764 -> 1.734, IMDb (edge kernel: not applicable) unchanged. Auto: two fronts from N = 8192 on. (The forward
        // window kernel gains nothing from it at any shape — cfg2 25.8 / 26.0 us, genome 22.8 / 23.2 — and keeps one front:
        // profiles/r06c_fwd_fronts.log.)
        {
          const int fronts = tn.bwd_fronts ? tn.bwd_fronts : (N >= 8192 ? 2 : 1);
          int sh = 0;
          while ((2 << sh) <= fronts) ++sh;
          if (sh > 0 && pk.tiles_full % (1 << sh) == 0) a.gm.ileave = sh;
        }
        hipError_t e = launch_fused_step(pk.tgs, L, a);
        if (e != hipSuccess) return fail_hip(e, "chord_bwd_fused");
        dW = nullptr;
        dV = nullptr;
      }
      if (dW && dV && pick_fused_edge_step(tn, dZ, V, dV, N, L, C, v_batch_stride, offs, &pk)) {
        BwdWinArgs a{dZ, W, dV, Geom{}, offs, w_total, true, s};
        a.V2 = V;
        a.out2 = dW;
        if (int rc = make_geom(tn, B, N, L, C, 4, pk.tgs, pk.TR, false, v_batch_stride, 0, pk.tiles_full + (pk.ragged ? 1 : 0), &a.gm))
          return rc;
        hipError_t e = launch_fused_edge_step(pk.tgs, L, a);
        if (e != hipSuccess) return fail_hip(e, "chord_bwd_fused_edge");
        dW = nullptr;
        dV = nullptr;
      }
      if (dW && dwv != 1 &&
          pick_dw_chunk(tn, dW, B, N, L, C, offs, (C % 4 == 0) && aligned_to(dZ, 16) && aligned_to(V, 16), &pk)) {
        BwdWinArgs a{dZ, V, dW, Geom{}, offs, w_total, false, s};
        int rc = window_launches(tn, pk, pk.all_edge, B, N, L, C, v_batch_stride, false, &a.gm, &a.edge,
                                 [&] { return launch_dwc(pk, L, a); }, "chord_dw_chunk");
        if (rc) return rc;
        dW = nullptr;
      } else if (dW && dwv == 2) {
        return fail(PSF_E_TUNING, "dw_variant=2 forced but the chunk-looping dW kernel does not apply to N=%lld L=%d C=%lld",
                    (long long)N, (int)L, (long long)C);
      }
      if (dW && C / 4 <= TGmax &&
          pick_window(tn, dW, B, N, L, C, offs, (C % 4 == 0) && aligned_to(dZ, 16) && aligned_to(V, 16), &pk,
                      rows_dw, false)) {
        // the dW tile store is chunk-clean only if every sequence starts on a 16-byte boundary
        const bool all_edge = pk.all_edge || ((N * (int64_t)L) % 4) != 0;
        BwdWinArgs a{dZ, V, dW, Geom{}, offs, w_total, false, s};
        int rc = window_launches(tn, pk, all_edge, B, N, L, C, v_batch_stride, false, &a.gm, &a.edge,
                                 [&] { return launch_dw(pk, L, a); }, "chord_dw_win");
        if (rc) return rc;
        dW = nullptr;
      }
      if (dV && pick_dv(tn, W, B, N, L, C, offs, (C % 4 == 0) && aligned_to(dZ, 16) && aligned_to(dV, 16), &pk)) {
        BwdWinArgs a{dZ, W, dV, Geom{}, offs, w_total, false, s};
        int rc = window_launches(tn, pk, pk.all_edge, B, N, L, C, N * C, true, &a.gm, &a.edge,
                                 [&] { return launch_dv(pk, L, a); }, "chord_dv_win");
        if (rc) return rc;
        dV = nullptr;
      }
    }
  }

  if (dV) {
    const bool vec_ok = (C % VECW == 0) && aligned_to(dZ, 16) && aligned_to(dV, 16);
    Geom gm;
    if (int rc = generic_geom(tn, B, N, L, C, vec_ok ? VECW : 1, true, N * C, &gm)) return rc;
    if (vec_ok)
      hipLaunchKernelGGL((chord_dv_generic_k<T, VECW>), dim3(gm.nblocks), dim3(kBlock), 0, s, dZ, W, dV, gm, offs);
    else
      hipLaunchKernelGGL((chord_dv_generic_k<T, 1>), dim3(gm.nblocks), dim3(kBlock), 0, s, dZ, W, dV, gm, offs);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, "chord_dv_generic launch");
  }
  if (dW) {
    const bool vec_ok = (C % VECW == 0) && aligned_to(dZ, 16) && aligned_to(V, 16);
    Geom gm;
    if (int rc = generic_geom(tn, B, N, L, C, vec_ok ? VECW : 1, false, v_batch_stride, &gm)) return rc;
    if (vec_ok)
      hipLaunchKernelGGL((chord_dw_generic_k<T, VECW>), dim3(gm.nblocks), dim3(kBlock), 0, s, dZ, V, dW, gm, offs);
    else
      hipLaunchKernelGGL((chord_dw_generic_k<T, 1>), dim3(gm.nblocks), dim3(kBlock), 0, s, dZ, V, dW, gm, offs);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, "chord_dw_generic launch");
  }
  return PSF_OK;
}

template <typename T>
int chain_impl(Tuning tn, const T* const* W_steps, const T* V0, T* const* out_steps, int32_t M, int32_t use_residual,
               int64_t B, int64_t N, int32_t L, int64_t C, int64_t v0_batch_stride, const int64_t* offsets,
               void* stream) {
  if (M < 0) return fail(PSF_E_SHAPE, "M must be >= 0");
  if (M == 0) return PSF_OK;
  if (!W_steps || !out_steps || !V0) return fail(PSF_E_NULL, "W_steps, out_steps and V0 must be non-NULL");
  if (use_residual && v0_batch_stride == 0 && B != 1)
    return fail(PSF_E_SHAPE, "a broadcast V0 cannot be the residual");
  for (int m = 0; m < M; ++m) {
    if (!W_steps[m] || !out_steps[m]) return fail(PSF_E_NULL, "step %d: NULL pointer", m);
    if (use_residual && out_steps[m] == V0) return fail(PSF_E_ALIAS, "step %d: out aliases the residual V0", m);
    if (out_steps[m] == (m == 0 ? V0 : out_steps[m - 1]))
      return fail(PSF_E_ALIAS, "step %d: out aliases the step's input", m);
  }

  if constexpr (sizeof(T) == 4) {
    // Short sequences: the whole chain in ONE launch with the sequence's X slice resident in LDS.
    ChainLdsPlan plan;
    // Every workgroup of a sequence streams the sequence's whole W: with `chunks` workgroups per sequence W crosses
    // L2 -> CU `chunks` times and every output row is stored in `chunks` pieces. Past ~8 the per-step kernels win
    // (profiles/r03n_chain_train_sweep.log, us per chain one launch / per step: ListOps N = 2000, C = 128, 32 chunks:
    // 361 / 223; N = 2048, C = 64, 16 chunks: 184 / 119; Pathfinder C = 32, 4 chunks: 72 / 78). chain_fused = 2 forces it.
    // Round 4 (the kernel built without SLP packing, profiles/r04al_chain_fused*.log, us per step, per-step / one launch):
    // when only the last result is kept (inference: two alternating buffers) the one launch wins wherever it fits, wide rows
    // included — N = 2000, C = 128: 18.6 / 15.1; N = 1024, C = 1024: 20.1 / 10.4; N = 2048, C = 64: 9.0 / 8.4 — and when every
    // step is kept (training) it loses from 65536 elements per sequence on — N = 2048, C = 32: 6.9 / 10.3; C = 64: 10.2 / 17.2 —
    // and wins below — N = 1024, C = 32: 7.1 / 5.6; N = 2048, C = 8: 6.9 / 4.3.
    // Round 6: the workgroups of a sequence now share an XCD (fwd_chain_lds.h: W crosses the fabric once per sequence) — one
    // launch, blockIdx order / XCD-aware, us per step-equivalent (profiles/r06u_chain_lds_xcd.log): N = 2000 x 128: 17.3 / 13.4;
    // N = 2048 x 64: 9.5 / 6.8; Pathfinder 1024 x 32: 3.4 / 2.6; attention map 1024 x 1024: 9.8 / 8.9. With every step kept
    // (training) it now wins up to 65 536 elements per sequence and, for N <= 1024, up to 131 072 (per step / one launch,
    // r06u_chain_keep_sweep.log): 2048 x 32: 6.9 / 5.8; 1024 x 64: 6.7 / 4.0; 1024 x 128: 9.2 / 7.8; but 2048 x 64: 9.8 / 11.0;
    // 2000 x 64: 10.5 / 10.8; 2000 x 128: 20.9 / 23.2.
    // Later in round 6: 1057 <= N <= 2048 on launches of >= 256 workgroups run chord_chain_rows_k (two channel groups per
    // workgroup: half the W streams). us per step, per-step launches / one group / two groups (profiles/r06v_chain_lds8_ab.log; lds8 = this kernel's first name):
    // last kept: 2000 x 128: 19.3 / 13.5 / 8.0; 2048 x 64: 8.9 / 6.8 / 4.3; every step kept: 2000 x 128: 20.9 / 23.6 / 17.2;
    // 2048 x 64: 9.7 / 11.0 / 8.9; 2000 x 64: 10.6 / 10.8 / 8.2; 2000 x 256: 37.7 / 39.5 / 32.6 - so with that instance the one
    // launch also takes training chains (up to the 524 288 elements per sequence measured); ListOps training step 2.507 -> 2.476 ms.
    // 2113 <= N <= 4160 (the LRA text task, N = 4097 x 32, B = 32: 256 workgroups) run the same kernel with one channel group and
    // five rows per thread when only the last result is kept (profiles/r06v_chain_long_ab2.log, per-step / one launch):
    // 13.0 / 7.2 us per step; 4096 x 32: 11.8 / 6.5; 3000 x 32: 9.6 / 5.4; with every step kept the per-step kernels stay
    // (13.5 / 14.5), and below 256 workgroups too (B = 16: 7.4 / 6.6 last kept but 8.3 / 11.3 kept; C = 8: 6.4 / 6.6).
    const int cf = tn.chain_fused;
    int kept = 0;  // step results that reach memory
    for (int m = 0; m < M; ++m) {
      bool later = false;
      for (int q = m + 1; q < M; ++q) later = later || out_steps[q] == out_steps[m];
      kept += later ? 0 : 1;
    }
    const bool few_kept = kept <= 2;
    bool ok = cf && M >= 2 && M <= kChainMaxSteps && B >= 1 && plan_chain_lds(N, C, L, M, &plan, tn.chain_cc, B) &&
              (cf == 2 || few_kept || N * C <= 65536 || (N <= 1024 && N * C <= 131072) || (plan.big == 1 && N * C <= 524288)) &&
              aligned_to(V0, 16) && B * (int64_t)plan.chunks <= 0x7fffffff;
    for (int m = 0; ok && m < M; ++m) ok = aligned_to(W_steps[m], 4) && aligned_to(out_steps[m], 16);
    if (ok) {
      if (int rc = check_dims(B, N, L, C, v0_batch_stride)) return rc;
      ChainArgs a;
      a.store_mask = 0;
      for (int m = 0; m < M; ++m) {
        a.W[m] = W_steps[m];
        a.out[m] = out_steps[m];
        bool later = false;  // a buffer that a later step overwrites (inference ping-pong) need not be stored
        for (int q = m + 1; q < M; ++q) later = later || out_steps[q] == out_steps[m];
        if (!later) a.store_mask |= (uint64_t)1 << m;
      }
      for (int m = M; m < kChainMaxSteps; ++m) {
        a.W[m] = nullptr;
        a.out[m] = nullptr;
      }
      a.V0 = V0;
      a.v0_bstride = v0_batch_stride;
      a.M = M;
      a.N = (int32_t)N;
      a.C = (int32_t)C;
      a.CG = (int32_t)(C / 4);
      a.chunks = plan.chunks;
      a.xcd_remap = tn.xcd_remap && plan.chunks > 1 ? 1 : 0;  // (one workgroup per sequence shares nothing with its neighbours)
      Offsets offs;
      make_offsets(N, L, offsets, &offs);
      hipError_t e = launch_chain_lds(plan, L, use_residual != 0, a, offs, (int)B, reinterpret_cast<hipStream_t>(stream));
      if (e != hipSuccess) return fail_hip(e, "chord_chain_lds launch");
      return PSF_OK;
    }
  }

  for (int m = 0; m < M; ++m) {
    const T* in = m == 0 ? V0 : out_steps[m - 1];
    const int64_t stride = m == 0 ? v0_batch_stride : N * C;
    // zigzag: every XCD walks its tile range forwards on even steps and backwards on odd ones, so a launch begins
    // with the tiles whose inputs the previous launch wrote LAST (still in that XCD's L2), not first
    tn.walk_backwards = (m & 1) != 0;
    int rc = fwd_impl<T>(tn, W_steps[m], in, use_residual ? V0 : nullptr, out_steps[m], B, N, L, C, stride, offsets, stream);
    if (rc) return rc;
  }
  return PSF_OK;
}


// ------------------------------------------------------------------------------------------------------
// the mixer with W computed inside the step (fwd_mlp_step.h)
// ------------------------------------------------------------------------------------------------------
struct MixerPlan {
  bool step_ok;            // the per-step kernels (fwd_mlp_step.h) cover the shape
  bool lds_ok;             // the single-launch LDS-resident kernel (mixer_lds.h) covers it
  int tgs, TR, KN, units;  // units = packed images over all M + 1 MLPs
  MixerLdsPlan lds;
};

// Whether the fused paths cover the shape; fills *mp. Mirrors the limits stated in include/psf_chord.h.
bool plan_mixer(int64_t N, int32_t E, int32_t M, const int32_t* h, int64_t C, int32_t L, MixerPlan* mp) {
  if (!h || M < 1 || M > 31 || E < 4 || E > 32 || (E & 3) || C < 4 || C > 32 || (C & 3) || L < kMlpStepLmin || L > kMlpStepLmax ||
      N < 1 || N > (int64_t)1 << 30)
    return false;
  int units = 0, nu_max = 0;
  for (int k = 0; k <= M; ++k) {
    if (h[k] < 1 || h[k] > 128) return false;
    const int nu = (h[k] + 31) / 32;
    units += nu;
    nu_max = nu > nu_max ? nu : nu_max;
  }
  if (units > 128) return false;
  mp->units = units;
  mp->lds_ok = plan_mixer_lds(N, C, L, M, nu_max, &mp->lds);
  mp->step_ok = false;
  const int tgs = ceil_log2(C / 4);
  if (tgs <= kMlpStepTgsMax) {
    const int TR = mlp_step_tile_rows(tgs);
    if (N >= 2 * (int64_t)TR) {  // the window may wrap at most once
      int KN = 2;
      for (int t = TR; t > 1; t >>= 1) ++KN;
      if (KN > L) KN = L;
      Offsets offs;
      make_offsets(N, L, nullptr, &offs);
      bool chord = true;
      for (int k = 0; k < KN; ++k) chord = chord && offs.v[k] == chord_off(k);  // near offsets are compile-time constants there
      if (chord) mp->step_ok = true, mp->tgs = tgs, mp->TR = TR, mp->KN = KN;
    }
  }
  return mp->step_ok || mp->lds_ok;
}

hipError_t launch_g(int tgs, const FwdMlpArgs& a) {
  switch (tgs) {
    case 0: return launch_mixer_g<0>(a);
    case 1: return launch_mixer_g<1>(a);
    case 2: return launch_mixer_g<2>(a);
    case 3: return launch_mixer_g<3>(a);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_mlp_step(int tgs, int L, const FwdMlpArgs& a) {
  switch (tgs) {
    case 0: return launch_fwd_mlp<0>(L, a);
    case 1: return launch_fwd_mlp<1>(L, a);
    case 2: return launch_fwd_mlp<2>(L, a);
    case 3: return launch_fwd_mlp<3>(L, a);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace

// ------------------------------------------------------------------------------------------------------
// extern "C"
// ------------------------------------------------------------------------------------------------------
// Other translation units of the library report through the same thread-local string (not part of the ABI).
extern "C" int psf_internal_fail(int code, const char* message) { return fail(code, "%s", message); }

extern "C" {

int psf_version(void) { return PSF_ABI_VERSION; }

const char* psf_last_error(void) { return g_err; }

const char* psf_build_info(void) {
  return "libpsf_chord: gfx950 (CDNA4, wave64) | hipcc " __VERSION__
         " | fwd: generic<f32,f64> + LDS-window<f32, L=4..20, LDS-DMA staging> + LDS-resident chain<f32, N<=2112; whole rows per thread: 8 channels N<=2048, 4 channels N<=4160>"
         " | bwd: generic dV/dW<f32,f64> + LDS-window dV/dW<f32> + fused dV+dW step<f32, C<=64; C=128 up to N=4096>"
         " | producers: fused MLP fwd (split-bf16 MFMA at f32 accuracy, f32 MFMA) + fused MLP bwd (split-bf16 MFMA on dual-use LDS planes, f32 MFMA),"
         " tall-skinny weight gradients (f32 MFMA), token embedding + positional add"
         ", wide producer MLPs (E <= 1024: stacked first layers as split-bf16 GEMMs from bf16 term planes, LDS-DMA ring)"
         " | mixer: W_m computed inside the chain step (per-step kernels; one LDS-resident launch for short sequences)"
         " | arithmetic of the chord path: uncontracted mul+add, links ascending"
#ifdef PSF_CSRC_HASH
         " | csrc=" PSF_CSRC_HASH  // build.csrc_hash() of the sources this library was built from (_lib.load compares)
#endif
      ;
}

int psf_device_info(char* buf, int32_t len) {
  if (!buf || len < 1) return fail(PSF_E_NULL, "buf is NULL or empty");
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return fail_hip(e, "hipGetDevice");
  char pci[64] = "?";
  (void)hipDeviceGetPCIBusId(pci, (int)sizeof(pci), dev);
  int xcds = 0, cus = 0;
  (void)hipDeviceGetAttribute(&xcds, hipDeviceAttributeNumberOfXccs, dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  hipDeviceProp_t prop;
  const char* name = hipGetDeviceProperties(&prop, dev) == hipSuccess ? prop.name : "?";
  snprintf(buf, (size_t)len, "pci=%s xcds=%d cus=%d name=%s", pci, xcds, cus, name);
  return PSF_OK;
}

int psf_chord_offsets(int64_t N, int32_t L, int64_t* offsets_out) {
  if (!offsets_out) return fail(PSF_E_NULL, "offsets_out is NULL");
  if (N < 1 || L < 1 || L > PSF_MAX_LINKS)
    return fail(PSF_E_SHAPE, "need N >= 1 and 1 <= L <= %d", PSF_MAX_LINKS);
  if (N > (int64_t)1 << 30) return fail(PSF_E_SHAPE, "N exceeds 2^30");
  Offsets o;
  make_offsets(N, L, nullptr, &o);
  for (int k = 0; k < L; ++k) offsets_out[k] = o.v[k];
  return PSF_OK;
}

int psf_chord_indices(int64_t N, int32_t L, int64_t* rows_out, int64_t* cols_out) {
  if (!rows_out || !cols_out) return fail(PSF_E_NULL, "rows_out / cols_out is NULL");
  if (N < 1 || L < 1 || L > PSF_MAX_LINKS)
    return fail(PSF_E_SHAPE, "need N >= 1 and 1 <= L <= %d", PSF_MAX_LINKS);
  if (N > (int64_t)1 << 30) return fail(PSF_E_SHAPE, "N exceeds 2^30");
  Offsets o;
  make_offsets(N, L, nullptr, &o);
  for (int64_t i = 0; i < N; ++i)
    for (int k = 0; k < L; ++k) {
      rows_out[i * L + k] = i;
      int64_t c = i + o.v[k];
      cols_out[i * L + k] = c >= N ? c - N : c;
    }
  return PSF_OK;
}

int psf_chord_spmm_fwd_f32(const float* W, const float* V, const float* res, float* out, int64_t B, int64_t N,
                           int32_t L, int64_t C, int64_t v_batch_stride, const int64_t* offsets, void* stream) {
  return fwd_impl<float>(snapshot(), W, V, res, out, B, N, L, C, v_batch_stride, offsets, stream);
}
int psf_chord_spmm_fwd_f64(const double* W, const double* V, const double* res, double* out, int64_t B,
                           int64_t N, int32_t L, int64_t C, int64_t v_batch_stride, const int64_t* offsets,
                           void* stream) {
  return fwd_impl<double>(snapshot(), W, V, res, out, B, N, L, C, v_batch_stride, offsets, stream);
}

int psf_chord_spmm_bwd_f32(const float* dZ, const float* W, const float* V, float* dW, float* dV, int64_t B,
                           int64_t N, int32_t L, int64_t C, int64_t v_batch_stride, const int64_t* offsets,
                           void* stream) {
  return bwd_impl<float>(snapshot(), dZ, W, V, dW, dV, B, N, L, C, v_batch_stride, offsets, stream);
}
int psf_chord_spmm_bwd_f64(const double* dZ, const double* W, const double* V, double* dW, double* dV,
                           int64_t B, int64_t N, int32_t L, int64_t C, int64_t v_batch_stride,
                           const int64_t* offsets, void* stream) {
  return bwd_impl<double>(snapshot(), dZ, W, V, dW, dV, B, N, L, C, v_batch_stride, offsets, stream);
}

int psf_chord_chain_fwd_f32(const float* const* W_steps, const float* V0, float* const* out_steps, int32_t M,
                            int32_t use_residual, int64_t B, int64_t N, int32_t L, int64_t C,
                            int64_t v0_batch_stride, const int64_t* offsets, void* stream) {
  return chain_impl<float>(snapshot(), W_steps, V0, out_steps, M, use_residual, B, N, L, C, v0_batch_stride, offsets, stream);
}
int psf_chord_chain_fwd_f64(const double* const* W_steps, const double* V0, double* const* out_steps, int32_t M,
                            int32_t use_residual, int64_t B, int64_t N, int32_t L, int64_t C,
                            int64_t v0_batch_stride, const int64_t* offsets, void* stream) {
  return chain_impl<double>(snapshot(), W_steps, V0, out_steps, M, use_residual, B, N, L, C, v0_batch_stride, offsets, stream);
}

int psf_chord_chain_bwd_supported(int64_t N, int32_t L, int64_t C, int32_t M) {
  return g_chain_bwd_fused.load() && chain_bwd_lds_fits(N, C, L, M) ? 1 : 0;
}

int psf_chord_chain_bwd_f32(const float* dOut, const float* const* W_steps, const float* V0, const float* const* X_steps,
                            float* const* dW_steps, float* dV0, float* const* dX_steps, int32_t M, int32_t use_residual,
                            int64_t B, int64_t N, int32_t L, int64_t C, const int64_t* offsets, void* stream) {
  if (M < 1) return fail(PSF_E_SHAPE, "M must be >= 1");
  if (!dOut || !W_steps || !V0 || !X_steps || !dW_steps || !dV0) return fail(PSF_E_NULL, "a required pointer is NULL");
  if (int rc = check_dims(B, N, L, C, N * C)) return rc;
  if (!g_chain_bwd_fused.load()) return PSF_E_UNSUPPORTED;  // (knob off: the caller runs the steps itself)
  const bool one_launch = chain_bwd_lds_fits(N, C, L, M);
  // the per-step path inside the library needs the M gradient buffers and, with the residual, psf_sum_tensors_f32's limits
  if (!one_launch && (!dX_steps || (use_residual && (M + 1 > 32 || (B * N * C) % 4 != 0)))) return PSF_E_UNSUPPORTED;
  if (B == 0) return PSF_OK;
  for (int m = 0; m < M; ++m) {
    const float* x = m == 0 ? V0 : X_steps[m];
    if (!W_steps[m] || !x || !dW_steps[m]) return fail(PSF_E_NULL, "step %d: NULL pointer", m);
    if (dW_steps[m] == W_steps[m]) return fail(PSF_E_ALIAS, "step %d: dW aliases W", m);
    if (!one_launch && !dX_steps[m]) return fail(PSF_E_NULL, "step %d: dX_steps[m] is NULL", m);
  }
  if (one_launch) {
    if (B > 0x7fffffff) return fail(PSF_E_SHAPE, "B too large");
    ChainBwdArgs a;
    for (int m = 0; m < M; ++m) {
      const float* x = m == 0 ? V0 : X_steps[m];
      if (!aligned_to(W_steps[m], 4) || !aligned_to(dW_steps[m], 4) || !aligned_to(x, 16))
        return fail(PSF_E_ALIGN, "step %d: W / dW must be 4-byte, X 16-byte aligned", m);
      a.W[m] = W_steps[m], a.X[m] = x, a.dW[m] = dW_steps[m];
    }
    for (int m = M; m < kChainMaxSteps; ++m) a.W[m] = nullptr, a.X[m] = nullptr, a.dW[m] = nullptr;
    if (!aligned_to(dOut, 16) || !aligned_to(dV0, 16)) return fail(PSF_E_ALIGN, "dOut and dV0 must be 16-byte aligned");
    a.dOut = dOut, a.dV0 = dV0, a.M = M, a.N = (int32_t)N, a.C = (int32_t)C;
    Offsets offs;
    make_offsets(N, L, offsets, &offs);
    hipError_t e = launch_chain_bwd_lds(L, (int)(C / 4), use_residual != 0, a, offs, (int)B, reinterpret_cast<hipStream_t>(stream));
    if (e != hipSuccess) return fail_hip(e, "chord_chain_bwd_lds launch");
    return PSF_OK;
  }
  // M per-step launches (the kernels psf_chord_spmm_bwd_f32 runs), the gradient handed from buffer to buffer, and ONE pass
  // over the residual terms at the end: what the caller's loop did, without M trips through its language's FFI
  const Tuning tn = snapshot();
  const float* g = dOut;
  const float* terms[kChainMaxSteps + 1];
  int nterms = 0;
  for (int m = M - 1; m >= 0; --m) {
    if (use_residual) terms[nterms++] = g;
    float* dx = (m == 0 && !use_residual) ? dV0 : dX_steps[m];
    if (int rc = bwd_impl<float>(tn, g, W_steps[m], m == 0 ? V0 : X_steps[m], dW_steps[m], dx, B, N, L, C, N * C, offsets, stream))
      return rc;
    g = dx;
  }
  if (use_residual) {
    terms[nterms++] = g;  // ((g_M + g_{M-1}) + ... + g_1) + g_0
    return psf_sum_tensors_f32(terms, nterms, B * N * C, dV0, stream);
  }
  return PSF_OK;
}

int64_t psf_mixer_fwd_workspace(int64_t N, int32_t E, int32_t M, const int32_t* h, int64_t C, int32_t L) {
  MixerPlan mp;
  if (!plan_mixer(N, E, M, h, C, L, &mp)) return -1;
  return (int64_t)mp.units * kX3ImageBytes;
}

int32_t psf_mixer_fwd_plan(int64_t N, int32_t E, int32_t M, const int32_t* h, int64_t C, int32_t L) {
  MixerPlan mp;
  if (!plan_mixer(N, E, M, h, C, L, &mp)) return 0;
  return (mp.lds_ok && g_mixer_lds.load()) ? 2 : (mp.step_ok ? 1 : 0);
}

int psf_mixer_fwd_in_f32(const psf_mixer_input* in, int64_t B, int64_t N, int32_t E, int32_t M, const float* const* A,
                         const float* const* a, const float* const* Bw, const float* const* b, const int32_t* h, int64_t C,
                         int32_t L, int32_t use_residual, float* V0, float* const* out_steps, void* workspace,
                         int64_t workspace_bytes, void* stream) {
  if (!in || !in->src || !A || !a || !Bw || !b || !h || !V0 || !out_steps || !workspace)
    return fail(PSF_E_NULL, "psf_mixer_fwd: NULL argument");
  MixerIn mi;
  mi.src = in->src, mi.weight = in->weight, mi.bias = in->bias, mi.pos = in->pos, mi.kind = in->kind, mi.K = in->K;
  if (in->kind == PSF_MIXER_IN_DATA) {
    mi.weight = mi.bias = mi.pos = nullptr, mi.K = 0;
    if (!aligned_to(in->src, 16)) return fail(PSF_E_ALIGN, "psf_mixer_fwd: X must be 16-byte aligned");
  } else if (in->kind == PSF_MIXER_IN_AFFINE) {
    if (in->K < 1 || in->K > 3) return fail(PSF_E_SHAPE, "psf_mixer_fwd: the affine input takes 1..3 values per position (K=%d)", (int)in->K);
    if (!in->weight) return fail(PSF_E_NULL, "psf_mixer_fwd: affine input without a weight");
    if (!aligned_to(in->src, 4) || !aligned_to(in->weight, 4) || (in->bias && !aligned_to(in->bias, 4)))
      return fail(PSF_E_ALIGN, "psf_mixer_fwd: affine input pointers must be 4-byte aligned");
  } else if (in->kind == PSF_MIXER_IN_TOKENS) {
    if (in->K < 1) return fail(PSF_E_SHAPE, "psf_mixer_fwd: empty vocabulary");
    if (!in->weight) return fail(PSF_E_NULL, "psf_mixer_fwd: token input without a table");
    if (!aligned_to(in->src, 8) || !aligned_to(in->weight, 16))
      return fail(PSF_E_ALIGN, "psf_mixer_fwd: tokens must be 8-byte, the table 16-byte aligned");
    mi.bias = nullptr;
  } else {
    return fail(PSF_E_SHAPE, "psf_mixer_fwd: unknown input kind %d", (int)in->kind);
  }
  if (mi.pos && !aligned_to(mi.pos, 16)) return fail(PSF_E_ALIGN, "psf_mixer_fwd: pos must be 16-byte aligned");
  MixerPlan mp;
  if (!plan_mixer(N, E, M, h, C, L, &mp))
    return fail(PSF_E_SHAPE, "psf_mixer_fwd: shape outside the fused path (N=%lld E=%d M=%d C=%lld L=%d; see psf_mixer_fwd_workspace)",
                (long long)N, (int)E, (int)M, (long long)C, (int)L);
  if (int rc = check_dims(B, N, L, C, N * C)) return rc;
  if (in->kind != PSF_MIXER_IN_DATA && !(mp.lds_ok && g_mixer_lds.load() && B <= 0x7fffffff))  // (before anything is launched)
    return fail(PSF_E_SHAPE, "psf_mixer_fwd: an input recipe (kind %d) is evaluated by the single-launch kernel only (short sequences, "
                "psf_mixer_fwd_plan() == 2); for N=%lld write the rows with psf_affine_rows_f32 / psf_embed_tokens_f32 and pass them",
                (int)in->kind, (long long)N);
  if (B == 0) return PSF_OK;
  if (workspace_bytes < (int64_t)mp.units * kX3ImageBytes || !aligned_to(workspace, 16))
    return fail(PSF_E_SHAPE, "psf_mixer_fwd: workspace too small (psf_mixer_fwd_workspace) or not 16-byte aligned");
  if (!aligned_to(V0, 16)) return fail(PSF_E_ALIGN, "psf_mixer_fwd: V0 must be 16-byte aligned");
  if (B * N > (int64_t)1 << 40) return fail(PSF_E_SHAPE, "psf_mixer_fwd: B*N too large");
  for (int k = 0; k <= M; ++k)
    if (!A[k] || !a[k] || !Bw[k] || !b[k]) return fail(PSF_E_NULL, "psf_mixer_fwd: NULL layer pointer (MLP %d)", k);
  for (int m = 0; m < M; ++m) {
    if (!out_steps[m]) return fail(PSF_E_NULL, "psf_mixer_fwd: step %d: NULL output", m);
    if (!aligned_to(out_steps[m], 16)) return fail(PSF_E_ALIGN, "psf_mixer_fwd: step %d: output not 16-byte aligned", m);
    if (out_steps[m] == V0) return fail(PSF_E_ALIAS, "psf_mixer_fwd: step %d: out aliases V0", m);
    if (m > 0 && out_steps[m] == out_steps[m - 1]) return fail(PSF_E_ALIAS, "psf_mixer_fwd: step %d: out aliases the step's input", m);
  }
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  Tuning tn = snapshot();

  // (1) all M + 1 weight sets -> unit images (one launch)
  int32_t O[32], first_unit[33];
  O[0] = (int32_t)C;
  for (int k = 1; k <= M; ++k) O[k] = L;
  hipError_t e = psf_x3_pack_launch(E, M + 1, A, a, Bw, b, h, O, workspace, first_unit, s);
  if (e != hipSuccess) return fail_hip(e, "psf_mixer_fwd: pack");
  Offsets offs;
  make_offsets(N, L, nullptr, &offs);
  if (mp.lds_ok && tn.mixer_lds && B <= 0x7fffffff) {  // short sequences: the whole mixer in ONE launch, V resident in LDS
    MixerLdsArgs la;
    la.in = mi;
    la.images = reinterpret_cast<const unsigned char*>(workspace);
    for (int k = 0; k <= M + 1; ++k) la.first_unit[k] = first_unit[k];
    la.V0 = V0;
    la.store_mask = 0;
    for (int m = 0; m < kMixerLdsMaxSteps; ++m) la.out[m] = m < M ? out_steps[m] : nullptr;
    for (int m = 0; m < M; ++m) {  // a buffer that a later step overwrites (two-buffer inference) is not stored at all
      bool later = false;
      for (int q = m + 1; q < M; ++q) later = later || out_steps[q] == out_steps[m];
      if (!later) la.store_mask |= 1u << m;
    }
    la.M = M, la.N = (int32_t)N, la.C = (int32_t)C, la.E = E, la.L = L, la.CG = (int32_t)(C / 4), la.WS = mp.lds.WS;
    la.TT = (int32_t)(N / 32), la.nu_max = mp.lds.nu_max;
    e = launch_mixer_lds(mp.lds, use_residual != 0, la, offs, (int)B, s);
    if (e != hipSuccess) return fail_hip(e, "chord_mixer_lds launch");
    return PSF_OK;
  }
  if (!mp.step_ok)
    return fail(PSF_E_TUNING, "psf_mixer_fwd: mixer_lds=0 but only the LDS-resident kernel covers N=%lld C=%lld", (long long)N, (long long)C);
  // the tile geometry of every launch below
  WinPick pk;
  pk.tgs = mp.tgs, pk.rows = mlp_step_rows(mp.tgs), pk.nt = 256, pk.TR = mp.TR, pk.KN = mp.KN;
  pk.tiles_full = (int)(N / mp.TR);
  pk.ragged = (N % mp.TR) != 0;
  const int TG = 1 << mp.tgs;
  // The step kernel's full-tile instance takes every row block as TR-aligned (scalar block addresses, fwd_mlp_step.h): N and
  // every far offset multiples of TR, rows of exactly 4 TG channels, a batch element under 2^31 bytes; anything else runs the
  // predicated instance on every tile. (ragged_in_one_launch's E: this step moves the data row, not the W row.)
  bool blocks_aligned = (N % mp.TR) == 0 && C == 4 * (int64_t)TG && N * C * 4 < ((int64_t)1 << 31) && N * (int64_t)E * 4 < ((int64_t)1 << 31);
  for (int k = mp.KN; k < L; ++k) blocks_aligned = blocks_aligned && (offs.v[k] % mp.TR) == 0;
  pk.all_edge = !blocks_aligned || !tn.fwd_split || ragged_in_one_launch(tn, pk.ragged, B, N, E, C);
  {  // (2) V0 = g(data): the matrix phase alone, on the same tiles
    FwdMlpArgs fa;
    fa.in = mi;
    fa.V = fa.res = nullptr;
    fa.out = V0;
    fa.images = reinterpret_cast<const unsigned char*>(workspace);
    fa.nu = first_unit[1] - first_unit[0];
    fa.E = E;
    fa.offs = offs;
    fa.wg_per_cu = fa.ablate = 0;
    fa.stream = s;
    const bool edge_all = pk.ragged && pk.all_edge;  // the g kernel needs its predicate only for rows >= N
    const int rc = window_launches(tn, pk, edge_all, B, N, L, C, N * C, false, &fa.gm, &fa.edge,
                                   [&] { return launch_g(mp.tgs, fa); }, "chord_mixer_g launch");
    if (rc) return rc;
  }
  for (int m = 0; m < M; ++m) {  // (3) the M steps
    FwdMlpArgs fa;
    fa.in = mi;
    fa.V = m == 0 ? V0 : out_steps[m - 1];
    fa.res = use_residual ? V0 : nullptr;
    fa.out = out_steps[m];
    fa.images = reinterpret_cast<const unsigned char*>(workspace) + (size_t)first_unit[m + 1] * kX3ImageBytes;
    fa.nu = first_unit[m + 2] - first_unit[m + 1];
    fa.E = E;
    fa.offs = offs;
    fa.wg_per_cu = tn.mixer_wg_limit;
    fa.ablate = tn.mixer_ablate;
    fa.stream = s;
    tn.walk_backwards = (m & 1) != 0;  // zigzag, as chain_impl
    const int rc = window_launches(tn, pk, pk.all_edge, B, N, L, C, N * C, false, &fa.gm, &fa.edge,
                                   [&] { return launch_mlp_step(mp.tgs, L, fa); }, "chord_fwd_mlp launch");
    if (rc) return rc;
  }
  return PSF_OK;
}

int psf_mixer_fwd_f32(const float* X, int64_t B, int64_t N, int32_t E, int32_t M, const float* const* A,
                      const float* const* a, const float* const* Bw, const float* const* b, const int32_t* h, int64_t C,
                      int32_t L, int32_t use_residual, float* V0, float* const* out_steps, void* workspace,
                      int64_t workspace_bytes, void* stream) {
  psf_mixer_input in;
  in.kind = PSF_MIXER_IN_DATA, in.K = 0, in.src = X, in.weight = in.bias = in.pos = nullptr;
  return psf_mixer_fwd_in_f32(&in, B, N, E, M, A, a, Bw, b, h, C, L, use_residual, V0, out_steps, workspace, workspace_bytes, stream);
}

int psf_set_tuning(const char* key, int32_t value) {
  if (!key) return fail(PSF_E_NULL, "key is NULL");
  for (auto& k : g_knobs)
    if (strcmp(k.key, key) == 0) {
      if (value < k.lo || value > k.hi)
        return fail(PSF_E_TUNING, "tuning %s: value %d outside [%d, %d]", key, (int)value, k.lo, k.hi);
      k.var->store(value);
      return PSF_OK;
    }
  return fail(PSF_E_TUNING, "unknown tuning key '%s'", key);
}

int psf_get_tuning(const char* key) {
  if (!key) return fail(PSF_E_NULL, "key is NULL");
  for (auto& k : g_knobs)
    if (strcmp(k.key, key) == 0) return k.var->load();
  return fail(PSF_E_TUNING, "unknown tuning key '%s'", key);
}

int psf_describe_fwd(int64_t B, int64_t N, int32_t L, int64_t C, int32_t elem_bytes, char* buf, int32_t cap) {
  if (!buf || cap < 1) return fail(PSF_E_NULL, "buf is NULL");
  if (int rc = check_dims(B, N, L, C, N * C)) return rc;
  if (elem_bytes != 4 && elem_bytes != 8) return fail(PSF_E_SHAPE, "elem_bytes must be 4 or 8");
  Offsets offs;
  make_offsets(N, L, nullptr, &offs);
  const int vecw = 16 / elem_bytes;
  const bool vec_ok = C % vecw == 0;
  WinPick pk;
  const Tuning tn = snapshot();
  const int variant = tn.fwd_variant;
  if (variant != 1 && elem_bytes == 4 && pick_window(tn, nullptr, B, N, L, C, offs, vec_ok, &pk, 2, true, 0, true)) {
    snprintf(buf, cap, "chord_fwd_win_k<f32,L=%d,TG=%d,R=%d,NT=%d> TR=%d near=%d far=%d tiles=%s", (int)L,
             1 << pk.tgs, pk.rows, pk.nt, pk.TR, pk.KN, (int)L - pk.KN,
             pk.all_edge ? "edge" : (pk.ragged ? "full+ragged" : (pk.aligned ? "full, aligned (scalar block addresses)" : "full")));
  } else {
    snprintf(buf, cap, "chord_fwd_generic_k<%s,VEC=%d>", elem_bytes == 4 ? "f32" : "f64", vec_ok ? vecw : 1);
  }
  return PSF_OK;
}

int psf_describe_chain_fwd(int64_t B, int64_t N, int32_t L, int64_t C, int32_t M, char* buf, int32_t cap) {
  if (!buf || cap < 1) return fail(PSF_E_NULL, "buf is NULL");
  if (int rc = check_dims(B, N, L, C, N * C)) return rc;
  ChainLdsPlan plan;
  const Tuning tn = snapshot();
  const int cf = tn.chain_fused;
  // (as an inference chain is run: only the last result kept)
  if (cf && M >= 2 && M <= kChainMaxSteps && B >= 1 && plan_chain_lds(N, C, L, M, &plan, tn.chain_cc, B)) {
    if (plan.big)
      snprintf(buf, cap, "chord_chain_rows_k<f32,L=%d,G=%d,R=%d> one launch for all %d steps, %d threads x %d rows x %d channels, %d workgroup(s) per sequence",
               (int)L, plan.cc, plan.rows, (int)M, plan.threads, plan.rows, 4 * plan.cc, plan.chunks);
    else
      snprintf(buf, cap, "chord_chain_lds_k<f32,L=%d,CC=%d,R=%d> one launch for all %d steps, %d threads, %d workgroup(s) per sequence",
               (int)L, plan.cc, plan.rows, (int)M, plan.threads, plan.chunks);
    return PSF_OK;
  }
  return psf_describe_fwd(B, N, L, C, 4, buf, cap);
}

}  // extern "C"
