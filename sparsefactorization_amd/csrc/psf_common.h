// psf_common.h — shared device/host definitions for the gfx950 chord-spmm kernels.
//
// Everything here is written for CDNA4 (wave64, 256 CUs in 8 XCDs, 160 KB LDS/CU); there is no other target.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/psf_chord.h"

namespace psf {

constexpr int kBlock = 256;  // 4 waves of 64 lanes; every kernel here uses 256-thread workgroups
constexpr int kXcds = 8;     // workgroups are dealt round-robin over the 8 XCDs (blocks b and b+8 share an L2)

// Link offsets, already reduced into [0, N). Passed by value: lives in the kernarg segment, so a
// statically indexed offs.v[k] is one scalar load and costs no VGPR.
struct Offsets {
  int32_t v[PSF_MAX_LINKS];
};

// Work decomposition of one step. A tile is TR consecutive rows x TG channel groups of one batch element;
// a channel group is VEC consecutive channels (one 16-byte access where C allows it).
struct Geom {
  int32_t N, L, C;
  int32_t CG;        // channel groups per row = ceil(C / VEC)
  int32_t tg_shift;  // TG = 1 << tg_shift threads share a row (TG <= 64 so a row never straddles a wave)
  int32_t TR;        // rows per tile
  int32_t tiles_n;   // row tiles per batch element covered by THIS launch
  int32_t tile0;     // index of the first of them (a launch may cover only the full or only the ragged tiles)
  int32_t chunks_c;  // ceil(CG / TG)
  int32_t per_b;     // tiles_n * chunks_c : workgroups per batch element
  uint32_t nblocks;  // B * per_b
  uint32_t per_b_inv, chunks_inv;  // floor(2^32 / per_b), floor(2^32 / chunks_c) (2^32 - 1 for a divisor of 1): udiv_inv
  uint32_t xq, xr;   // nblocks / 8, nblocks % 8 for the bijective XCD remap
  int32_t remap;     // 1: logical block order is contiguous per XCD group
  int32_t ileave;    // s > 0: the tiles of a batch element are walked as 2^s interleaved fronts (position t of the walk is row
                     // block (t mod 2^s) tiles_n / 2^s + t / 2^s): rows N / 2^s apart are in flight together. Needs 2^s | tiles_n.
                     // Knob "bwd_fronts" (fused backward step; the forward kernels gain nothing from it: r06c_fwd_fronts.log)
  int32_t aligned;   // 1: N is a multiple of TR, so is every far offset, and a batch element's rows span < 2^31 bytes: every
                     //    row block a full tile touches is TR-aligned and never wraps inside (scalar block addresses)
  int64_t v_bstride; // elements between batch elements of the gathered operand (0 = broadcast)
};

// Logical workgroup index for this hardware workgroup.
// Hardware deals blockIdx.x round-robin over the XCDs; we want all tiles of a batch element on one XCD so
// that the L re-reads of each V row hit that XCD's 4 MB L2 instead of going out to Infinity Cache / HBM.
// The map is a bijection for any nblocks (speed only, never correctness).
__device__ __forceinline__ uint32_t logical_block(const Geom& gm) {
  const uint32_t bid = blockIdx.x;
  if (!gm.remap) return bid;
  const uint32_t xcd = bid % kXcds, idx = bid / kXcds;
  const uint32_t base = xcd < gm.xr ? xcd * (gm.xq + 1) : gm.xr * (gm.xq + 1) + (xcd - gm.xr) * gm.xq;
  if (gm.remap == 2) {  // the XCD walks its range backwards (odd steps of a chain: start where the last launch ended)
    const uint32_t cnt = gm.xq + (xcd < gm.xr ? 1u : 0u);
    return base + (cnt - 1u - idx);
  }
  return base + idx;
}

// n / d for wave-uniform operands with inv = floor(2^32 / d) from the host (2^32 - 1 when d = 1): the high product is the
// quotient or one less, one compare fixes it — six scalar instructions where the compiler's division by a run-time value
// takes ~25 and a round trip through the vector unit.
__host__ __device__ inline uint32_t udiv_inv_of(uint32_t d) { return d <= 1u ? 0xffffffffu : (uint32_t)(((uint64_t)1 << 32) / d); }
__device__ __forceinline__ uint32_t udiv_inv(uint32_t n, uint32_t d, uint32_t inv) {
  uint32_t q = __umulhi(n, inv);
  if (n - q * d >= d) ++q;
  return q;
}

__device__ __forceinline__ void decode_block(const Geom& gm, int& b, int& tile, int& chunk) {
  const uint32_t lb = logical_block(gm);
  b = (int)udiv_inv(lb, (uint32_t)gm.per_b, gm.per_b_inv);
  const uint32_t rem = lb - (uint32_t)b * (uint32_t)gm.per_b;
  const uint32_t t = udiv_inv(rem, (uint32_t)gm.chunks_c, gm.chunks_inv);
  chunk = (int)(rem - t * (uint32_t)gm.chunks_c);
  // interleaved fronts (Geom::ileave): position t of the walk is row block (t mod 2^s) tiles_n / 2^s + t / 2^s
  const uint32_t tt = gm.ileave ? (t & ((1u << gm.ileave) - 1u)) * ((uint32_t)gm.tiles_n >> gm.ileave) + (t >> gm.ileave) : t;
  tile = (int)tt + gm.tile0;
}

// ---- exact (uncontracted) arithmetic: a rounded product followed by a rounded sum, like the CPU path ----
__device__ __forceinline__ float mul_rn(float a, float b) { return __fmul_rn(a, b); }
__device__ __forceinline__ float add_rn(float a, float b) { return __fadd_rn(a, b); }
__device__ __forceinline__ double mul_rn(double a, double b) { return __dmul_rn(a, b); }
__device__ __forceinline__ double add_rn(double a, double b) { return __dadd_rn(a, b); }

// A wave-uniform global address pinned to scalar registers, opaque to the optimiser. Used as `sbase(block address) + lane
// byte offset`: the sum then stays "scalar base + 32-bit vector offset" and selects that addressing form of the memory
// instructions; left visible, the common `base0 + lane offset` of many loads is hoisted into a 64-bit vector address and
// every load pays a 64-bit vector add for its block offset instead (r04: 180 -> ~60 vector instructions in the request phase
// of the fused backward step). For a value already in scalar registers the two readfirstlanes are copies. The result is a
// global-address-space pointer (the integer round trip would otherwise leave a flat one).
#define PSF_GLOBAL __attribute__((address_space(1)))
template <typename P>
__device__ __forceinline__ PSF_GLOBAL P* sbase(P* p) {
  const uint64_t v = reinterpret_cast<uint64_t>(p);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)v), hi = __builtin_amdgcn_readfirstlane((uint32_t)(v >> 32));
  return reinterpret_cast<PSF_GLOBAL P*>(((uint64_t)hi << 32) | lo);
}

// ---- VEC-wide register vectors ----
template <typename T, int VEC>
struct Vec {
  T e[VEC];
};
template <>
struct __attribute__((aligned(16))) Vec<float, 4> {
  float e[4];
};
template <>
struct __attribute__((aligned(16))) Vec<double, 2> {
  double e[2];
};

template <typename T, int VEC>
__device__ __forceinline__ Vec<T, VEC> ld(const T* p) {
  return *reinterpret_cast<const Vec<T, VEC>*>(p);
}
template <typename T, int VEC>
__device__ __forceinline__ void st(T* p, const Vec<T, VEC>& v) {
  *reinterpret_cast<Vec<T, VEC>*>(p) = v;
}

// A copy of a per-lane byte offset that the optimiser treats as a new value (one v_mov at most). The scalar-base form needs
// to SEE a 32-bit offset being widened in the block of the memory instruction; the widening of a value used in several blocks
// is done once, in the first, and the later blocks get a 64-bit register they can only add to the base with a 64-bit vector
// add. So: one laundered copy per later block that uses the offset.
__device__ __forceinline__ uint32_t lane_off(uint32_t v) {
  asm volatile("" : "+v"(v));
  return v;
}

// ---- LDS results in kernels that also issue MFMAs: the rule (profiles/r04b_mixer_lds_wait.md, isa_lint.py R3) ----
// (The MECHANISM below is unproven — an inference from the failure's signature. Round 6 found the bare pattern in every
// chord kernel, bit-exact beside a co-resident MFMA kernel, and in this rule's own emitted code: what the rule removes is the
// failing build's remaining difference, OTHER LDS READS OF THE ROW STILL IN FLIGHT when a packed pair is consumed.)
// A packed-f32 instruction (v_pk_*_f32: 64-bit register-pair operands) must not be the first consumer of a ds_read result
// right behind the counted `s_waitcnt lgkmcnt(n)` that released it. That is the only thing the ISA of the one sporadically
// wrong build of this library (chord_fwd_mlp_k, round 4) has which its two clean builds have not; the counts themselves were
// right (in-order LDS returns, no scalar load in flight, no hand-written instruction anywhere near). So the kernels that
// mix MFMA phases with LDS-fed f32 arithmetic read ALL LDS operands of a row first, wait for lgkmcnt(0), and make every
// consumer DATA-DEPENDENT on that wait: lds_wait_all() is the wait, behind_wait(r) re-defines r behind it (volatile asm
// statements keep their order; an empty one costs no instruction). A bare `asm volatile("s_waitcnt ...")` with a memory
// clobber does not do that: hipcc is free to linearise register-only arithmetic in front of it, and did (the first
// "fixed" build had its wait behind most of the row's multiplies — r04b, "What the ISA says").
__device__ __forceinline__ void lds_wait_all() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void behind_wait(float& r) { asm volatile("" : "+v"(r)); }
__device__ __forceinline__ void behind_wait(float __attribute__((ext_vector_type(4))) & r) { asm volatile("" : "+v"(r)); }
template <int VEC>
__device__ __forceinline__ void behind_wait(Vec<float, VEC>& v) {
#pragma unroll
  for (int i = 0; i < VEC; ++i) asm volatile("" : "+v"(v.e[i]));
}

// the same through a global-address-space byte pointer (sbase(...) + lane offset); the access itself is made on a built-in
// vector type (a class type cannot be copied out of an address-space-qualified lvalue)
template <typename T, int VEC>
__device__ __forceinline__ Vec<T, VEC> ldg(const PSF_GLOBAL char* p) {
  using Raw = T __attribute__((ext_vector_type(VEC)));
  const Raw r = *reinterpret_cast<const PSF_GLOBAL Raw*>(p);
  Vec<T, VEC> v;
#pragma unroll
  for (int i = 0; i < VEC; ++i) v.e[i] = r[i];
  return v;
}
template <typename T, int VEC>
__device__ __forceinline__ void stg(PSF_GLOBAL char* p, const Vec<T, VEC>& v) {
  using Raw = T __attribute__((ext_vector_type(VEC)));
  Raw r;
#pragma unroll
  for (int i = 0; i < VEC; ++i) r[i] = v.e[i];
  *reinterpret_cast<PSF_GLOBAL Raw*>(p) = r;
}

template <typename T, int VEC>
__device__ __forceinline__ void axpy_rn(Vec<T, VEC>& acc, T w, const Vec<T, VEC>& x) {
#pragma unroll
  for (int i = 0; i < VEC; ++i) acc.e[i] = add_rn(acc.e[i], mul_rn(w, x.e[i]));
}

// The chord pattern's link offsets (get_chord_indices_assym, SyntheticExperiments/psf.py:7-32): 0, 1, 2, 4, ...
// The LDS-window kernels serve the NEAR links (offset <= tile length < N, so never wrapped) with these as compile-time
// constants — every window read is then a per-row base address plus an immediate; the host routes any other near
// offsets to the generic kernels. Far offsets stay runtime values (they wrap mod N).
constexpr int chord_off(int k) { return k == 0 ? 0 : 1 << (k - 1); }

constexpr int ilog2_floor(int x) { return x <= 1 ? 0 : 1 + ilog2_floor(x >> 1); }
constexpr int imin(int a, int b) { return a < b ? a : b; }
__device__ __forceinline__ int imin_rt(int a, int b) { return a < b ? a : b; }
constexpr int imax(int a, int b) { return a > b ? a : b; }

}  // namespace psf
