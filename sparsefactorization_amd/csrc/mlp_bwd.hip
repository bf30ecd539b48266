// mlp_bwd.hip — fused backward of the producer MLPs (training path of mlp_fwd.hip).
//
//   Y_k = GELU(X · A_k^T + a_k) · B_k^T + b_k,  k < K      (MLPBlock, SyntheticExperiments/psf.py:35-60; PSFNet
//   applies g and fs[0..M) to the same `data`, psf.py:165,175)
//
// Given dY_k, one launch produces dX = sum_k dHpre_k · A_k and per-wave partial sums of dA_k, da_k, dB_k, db_k; a
// two-stage fixed-order reduction (no float atomics: bit-reproducible) finishes the weight gradients. Nothing of
// size [T, h] ever exists in memory: the hidden layer is RECOMPUTED from X (16 MFMAs per 32-token tile), which
// is cheaper than the 2 x [T, K*h] floats autograd would save and re-read (2.35 GB at Order N=16384, B=40).
// Through PyTorch this backward is 4K GEMMs + K GELU-backward kernels + K-1 accumulations of dX: 4.9 ms of the
// 8.3 ms training step (profiles/r01_train_step_profile_after.log).
//
// A hidden layer wider than 32 is processed as independent 32-row "units" (u = (k, ht)): Hpre, dHpost, dHpre, dA
// and dB rows of a unit depend on no other unit; only dX (summed over all units) and db (taken from the ht = 0
// unit) cross units.
//
// Per wave: TPW tiles of 32 tokens; X's operand for the contraction over e (xr) is held in registers, the tiles
// themselves stay in LDS for the contraction over tokens (xT); the dX^T accumulators persist over all units. Per unit and tile, on
// v_mfma_f32_32x32x2_f32 (D[i][j] += A[i][k] B[k][j]; A-operand lane l = (i = l&31, k = l>>5), B-operand lane
// l = (k = l>>5, j = l&31), result register r of lane l = D[(r&3) + 8(r>>2) + 4(l>>5)][l&31]):
//   1. Hpre^T  [j x tok] = A_u · X^T + a_u                              16 MFMAs   (A from the LDS image, xr)
//   2. dHpost^T[j x tok] = B_u^T · dY^T                                 O/2 MFMAs  (B^T from the image, dY tile via LDS)
//   3. VALU: Phi, phi from one rcp + one exp;  Hpost = x Phi;  G = dHpost (Phi + x phi)   (G = dHpre^T)
//   4. dB_u^T  [j x o]  += Hpost^T · dY    (contraction over tokens)    16 MFMAs   (Hpost^T re-laid through LDS)
//   5. dA_u    [j x e]  += G · X           (contraction over tokens)    16 MFMAs   (G re-laid through LDS, xT)
//   6. dX^T    [e x tok]+= A_u^T · G       (contraction over j)         16 MFMAs   (G's accumulator registers ARE
//        the B operand: register r pairs rows {row(r,0), row(r,1)}; the A operand takes A_u[that row][e] from LDS)
//   da_u, db_u: sums of the step-5 / step-4 operand registers (their k index is the token).
// After the wave's TPW tiles the unit's 34 accumulator registers are summed over the workgroup's four waves
// through LDS and flushed to the workgroup's slot of the partial buffer.
//
// Limits: E <= 32 (multiple of 4), h <= 128, O <= 32, K <= 32; anything else stays on autograd.
//
// The kernel: mlp_bwd_x3p_k — all five GEMMs on the bf16 pipe, every operand split ONCE into bf16 planes in LDS that serve
// both orientations (row reads and ds_read_b64_tr_b16, mlp_planes.h): 0.70 ms at Order N = 16384, B = 40. See the comment above
// it. (Rounds 1-2 built two predecessors — all five GEMMs on v_mfma_f32_32x32x2_f32, which occupies the vector ALU's datapath:
// 1.17 ms; steps 1, 2, 6 on the bf16 pipe with the token contractions left on the f32 instruction: 1.03 ms — and round 3 a
// variant with the SIMD's two waves in producer / consumer roles, 8 % slower: profiles/r02h_*.log, r03ai_mlp_bwd_roles.log. They
// stayed selectable by knob until round 5 removed them: 1 270 of this file's 1 890 lines.)
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "../../include/psf_chord.h"
#include "mlp_planes.h"
#include "mlp_x3_common.h"

extern "C" int psf_internal_fail(int code, const char* message);

namespace {

using psf_x3::bf16x8;
using psf_x3::cd_row;
using psf_x3::f32x16;
using psf_x3::Frag3;
using psf_x3::mfma6;
using psf_x3::split_pack8;
using psf_x3::split_pack8_pk;

constexpr int kMaxMlps = 32;
constexpr int kMaxUnits = 128;
constexpr int kPart = 1024 + 1024 + 64 + 64;  // per (group, unit): dA [j][e] | dB^T [j][o] | da [half][j] | db [half][o]
constexpr int kSlices = 64;                   // stage-1 reduction slices
static_assert(kPart % 4 == 0, "the stage-1 reduction reads the partial sums as float4");

// split-bf16 unit image (bytes), every part already in MFMA operand order:
//   A terms   3 x [32 j][80 B: 32 e bf16 + pad]            step 1, A operand (8 consecutive e of row j per ds_read_b128)
//   sa        32 f32
//   A^T terms 3 x [2 s][2 half][32 e][8 bf16]               step 6, A operand: A[ht + rho(s, half, i)][e]
//   B^T terms 3 x [2 s][2 half][32 j][8 bf16]               step 2, A operand: B[o = 16 s + 8 half + i][ht + j]
// rho(s, half, i) = (i & 3) + 16 s + 8 (i >> 2) + 4 half: the hidden row that accumulator register 8 s + i of a lane of
// that half holds, so a lane's registers 8s..8s+7 ARE its B-operand fragment of k-step s (accumulator-as-operand).
constexpr int kXARow = 80;
constexpr int kXATerm = 32 * kXARow;                // 2560
constexpr int kXOffSa = 3 * kXATerm;                // 7680
constexpr int kXOffAT = kXOffSa + 128;              // 7808
constexpr int kXFragTerm = 2 * 2 * 32 * 16;         // 2048
constexpr int kXOffBT = kXOffAT + 3 * kXFragTerm;   // 13952
constexpr int kXImgBytes = kXOffBT + 3 * kXFragTerm;  // 20096
constexpr int kImgFloatsMax = kXImgBytes / 4;  // workspace slot per unit (sized for the first split-bf16 image: the planes image is smaller)

struct BwdMlp {
  const float* A;   // [h, E]
  const float* a;   // [h]
  const float* B;   // [O, h]
  const float* dY;  // [T, O]
  float* dA;
  float* da;
  float* dB;
  float* db;
  int32_t h, O;
};

struct BwdArgs {
  BwdMlp m[kMaxMlps];
  // unit -> MLP | hidden block << 8. Dwords, not bytes: a byte table indexed by the (uniform) unit counter becomes a
  // global_load_ubyte + s_waitcnt vmcnt(0) — a full memory round trip that also drains every prefetch in flight — where
  // a dword table is one s_load_dword.
  uint32_t unit[kMaxUnits];
  const float* X;
  float* dX;        // [T, E] or nullptr
  float* images;    // U images of kImg floats
  float* partials;  // [G][U][kPart], one slot per workgroup
  float* stage1;    // [slices][U * kPart] (room for kSlices)
  int32_t slices;   // stage-1 slices in use (<= kSlices)
  int64_t T;
  int64_t G;        // workgroups = partial slots
  int32_t E, K, U;
};

// y = GELU(x) = x Phi(x) and dy/dx = Phi(x) + x phi(x). Phi by Abramowitz &
// Stegun 26.2.17 (|error| <= 7.5e-8): t = 1/(1 + 0.2316419 |x|), q = phi(x) (b1 t + ... + b5 t^5), Phi = x >= 0 ?
// 1 - q : q, with 1/sqrt(2 pi) folded into the b's; E = exp(-x^2/2) = phi(x) sqrt(2 pi) serves both results.
// On single values: packed f32 VALU (v_pk_fma_f32 ...) halves the instruction count but is slow beside MFMAs
// (MI355X_MICROARCH.md constants table: one v_pk_fma_f32 costs +22 cycles against two v_fma_f32 next to a bf16 MFMA): the
// planes kernel 0.873 -> 0.844 ms with the scalar form (profiles/r02ah_nopk.log; profiles/coexec_lab.hip measures why).
__device__ __forceinline__ void gelu_and_grad1(float x, float& y, float& dydx) {
  const float t = __builtin_amdgcn_rcpf(fmaf(fabsf(x), 0.2316419f, 1.0f));
  float p = fmaf(0.53070271f, t, -0.72657602f);
  p = fmaf(p, t, 0.71070687f);
  p = fmaf(p, t, -0.14224837f);
  p = fmaf(p, t, 0.12741479f);
  p = p * t;
  const float E = __builtin_amdgcn_exp2f((x * x) * -0.72134752044448170368f);
  const float dlt = copysignf(0.5f - p * E, x);  // 0.5 - q >= 0
  const float Phi = 0.5f + dlt;
  y = x * Phi;
  dydx = fmaf(x * 0.39894228040143267794f, E, Phi);
}
// ------------------------------------------------------------------------------------------------------------------
// split-bf16 variant on dual-use LDS planes (mlp_planes.h): ALL five GEMMs on the bf16 pipe, every operand split ONCE
// ------------------------------------------------------------------------------------------------------------------
// What mlp_bwd_x3_k pays for the two contractions over tokens (steps 4, 5) is a second split of dY, G and X in the
// transposed orientation — or, as built, the f32 instruction that shares the VALU's datapath. Here every activation is
// split once, stored as bf16 planes, and the orientation an MFMA needs is chosen by the READ: ds_read_b128 along a row,
// ds_read_b64_tr_b16 down a column (the hardware transposes 4 x 16 blocks on the way to the registers).
//   X tile      split at block start into three planes [tok][e] per tile: row read = step-1 B operand, transposed read =
//               step-5 B operand. No f32 copy of X in LDS, no X fragments held in registers.
//   dY tile     each lane loads the 8 (16) outputs of ITS token straight from global memory = its step-2 B fragment; the
//               same split terms go through a [tok][o] scratch plane and come back transposed as the step-4 B operand.
//   Hpost^T, G  accumulator layout -> split16 -> one ds_write_b64 per four registers into a [tok][j] scratch plane ->
//               transposed read = step-4 / step-5 A operand. G's packed terms ARE the step-6 B fragments.
//   A_u         one [j][e] plane per term: row read = step-1 A operand, transposed read in accumulator order = step-6 A
//               operand (the A^T copy of the x3 image is gone: 12.4 KB per unit image instead of 20 KB).
// A wave has TWO 2 KB scratch planes next to its X planes; the terms go through them two at a time and are read back
// transposed into registers (a plane is rewritten as soon as its reads are issued: LDS operations of one wave execute in
// order), so two tiles per wave x eight waves x two image buffers fit 160 KB.
// da_u = G 1: three more MFMAs per k-step on G's transposed terms against a fragment of ones (every column of the result
// is da_u; the matrix pipe has the slack, the VALU does not). db: per-lane sums of the dY registers, reduced over the 32
// token lanes once per unit (DPP), in the ht = 0 unit only.
constexpr int kPOffSa = 3 * psf_x3::kPlaneBytes;       // 6144: sa in accumulator-register order [half][16]
constexpr int kPOffBT = kPOffSa + 128;                 // 6272: B^T terms as in the x3 image
constexpr int kPImgBytes = kPOffBT + 3 * kXFragTerm;   // 12416
constexpr int kPImgVecs = kPImgBytes / 16;             // 776
constexpr int kPScrBytes = 2 * psf_x3::kPlaneBytes + 512;  // two scratch planes; as f32: combine tile [32][32] | da [2][32] | db [2][32]
static_assert(kPImgBytes <= kXImgBytes, "the workspace slot per unit is sized for the x3 image");

__device__ __forceinline__ Frag3 load_frag3(const unsigned char* p, int term_stride) {
  Frag3 f;
  f.t1 = *reinterpret_cast<const bf16x8*>(p);
  f.t2 = *reinterpret_cast<const bf16x8*>(p + term_stride);
  f.t3 = *reinterpret_cast<const bf16x8*>(p + 2 * term_stride);
  return f;
}

__global__ void __launch_bounds__(256) mlp_bwd_x3p_pack_k(const BwdArgs a) {
  using psf_x3::bf16_bits;
  using psf_x3::plane_off;
  using psf_x3::split3;
  const int u = blockIdx.x;
  const BwdMlp d = a.m[a.unit[u] & 0xff];
  const int ht = 32 * (int)(a.unit[u] >> 8), E = a.E;
  unsigned char* img = reinterpret_cast<unsigned char*>(a.images) + (size_t)u * kXImgBytes;
  uint16_t* img16 = reinterpret_cast<uint16_t*>(img);
  float* img32 = reinterpret_cast<float*>(img);
  for (int i = threadIdx.x; i < 32 * 32; i += 256) {  // A planes [j][e], swizzled
    const int j = i >> 5, e = i & 31;
    const float v = (e < E && ht + j < d.h) ? d.A[(ht + j) * E + e] : 0.f;
    uint32_t t1, t2, t3;
    split3(v, t1, t2, t3);
    const int at = (plane_off(j, e >> 3) >> 1) + (e & 7);
    img16[at] = bf16_bits(t1);
    img16[psf_x3::kPlaneBytes / 2 + at] = bf16_bits(t2);
    img16[psf_x3::kPlaneBytes + at] = bf16_bits(t3);
  }
  for (int q = threadIdx.x; q < 32; q += 256) {  // sa[half][r] = a[ht + cd_row(r, half)]
    const int j = ht + cd_row(q & 15, q >> 4);
    img32[kPOffSa / 4 + q] = j < d.h ? d.a[j] : 0.f;
  }
  for (int q = threadIdx.x; q < 2 * 2 * 32 * 8; q += 256) {  // B^T: [s][half][j = col][i] = B[o = 16 s + 8 half + i][ht + j]
    const int i = q & 7, col = (q >> 3) & 31, hf = (q >> 8) & 1, s = q >> 9;
    const int o = 16 * s + 8 * hf + i;
    const float vb = (o < d.O && ht + col < d.h) ? d.B[o * d.h + ht + col] : 0.f;
    uint32_t t1, t2, t3;
    split3(vb, t1, t2, t3);
    img16[(kPOffBT + 0 * kXFragTerm) / 2 + q] = bf16_bits(t1);
    img16[(kPOffBT + 1 * kXFragTerm) / 2 + q] = bf16_bits(t2);
    img16[(kPOffBT + 2 * kXFragTerm) / 2 + q] = bf16_bits(t3);
  }
}

template <int CTRL>
__device__ __forceinline__ float dpp_add(float v) {
  return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
// sum over the 32 lanes of the lane's half of the wave; every lane ends up with the total
__device__ __forceinline__ float half_sum(float v) {
  v = dpp_add<0xB1>(v);   // quad_perm [1,0,3,2]
  v = dpp_add<0x4E>(v);   // quad_perm [2,3,0,1]
  v = dpp_add<0x141>(v);  // row_half_mirror
  v = dpp_add<0x140>(v);  // row_mirror
  return v + __shfl_xor(v, 16, 64);
}

#ifdef PSF_X3P_TRACE  // profiles/x3plab.hip: shader-clock timestamps of one unit of one workgroup, per wave
__device__ unsigned long long psf_x3p_trace[8][32];
#define PSF_TRACE(slot)                                                        \
  do {                                                                         \
    if (trace_on) {                                                            \
      const unsigned long long t_ = clock64();                                 \
      if (lane == 0) psf_x3p_trace[wv][(slot)] = t_;                           \
    }                                                                          \
  } while (0)
#else
#define PSF_TRACE(slot) do { } while (0)
#endif

// 512 threads, one workgroup per CU (two waves per SIMD, <= 256 registers), two image buffers.
// NDY: dY values per lane and tile = outputs of one token that one half of the wave covers: 8 (O <= 16) or 16.
template <int TPW, int NDY>
__global__ void __launch_bounds__(512, 1)
mlp_bwd_x3p_k(const BwdArgs a) {
  using namespace psf_x3;
  constexpr int NW = 8;  // (twelve waves of one tile each, three per SIMD at 168 registers: 0.94 vs 0.86 ms, profiles/r02ae)
  constexpr int NS2 = NDY / 8;  // k-steps of the contraction over outputs (step 2)
  constexpr int kWaveBytes = TPW * 3 * kPlaneBytes + kPScrBytes;
  // static LDS: 2 images + 8 waves x (TPW x 3 X planes + scratch) = 160,000 bytes at TPW = 2
  __shared__ __attribute__((aligned(16))) unsigned char img_lds[2 * kPImgBytes];
  __shared__ __attribute__((aligned(16))) unsigned char wave_lds[NW * kWaveBytes];
  const int tid = threadIdx.x, lane = tid & 63, wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int c = lane & 31, half = lane >> 5;
  unsigned char* XP = wave_lds + wv * kWaveBytes;  // the wave's X planes [tile][term]
  unsigned char* HP = XP + TPW * 3 * kPlaneBytes;                  // scratch plane: Hpost^T / G, one term at a time
  unsigned char* YP = HP + kPlaneBytes;                            // scratch plane: dY, one term at a time
  float* SCR = reinterpret_cast<float*>(HP);                       // the same bytes for the cross-wave combine
  const PlaneLane L = plane_lane(lane);
  const int E = a.E, U = a.U;
  const int64_t tiles = (a.T + 31) / 32;
  const int64_t tiles_per_block = NW * TPW;
  const unsigned char* images = reinterpret_cast<const unsigned char*>(a.images);

  // Image u by LDS-DMA: 16 bytes per lane, global -> LDS without passing through registers. Issued as inline assembly,
  // not through __builtin_amdgcn_global_load_lds: hipcc makes every later read of LDS that may alias a DMA in flight wait
  // for vmcnt(0) (its alias rule for LDS-DMA is all-or-nothing), and that wait also drains the dY prefetch issued a few
  // instructions earlier — a full memory round trip per unit at the first image read, measured (profiles/r02ac_x3plab.log).
  // Hidden from the compiler, the DMA costs one explicit s_waitcnt vmcnt(0) in front of the barrier that publishes the image.
  auto stage = [&](int u) {
    const unsigned char* src = images + (size_t)u * kXImgBytes;
    unsigned char* dst = img_lds + (u & 1) * kPImgBytes;
    for (int v0 = 0; v0 < kPImgVecs; v0 += 64 * NW) {
      const int v = v0 + tid;
      const uint32_t lds_at = __builtin_amdgcn_readfirstlane(
          (uint32_t)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)(dst + 16 * (v0 + (tid & ~63)))));
      uint32_t m0_saved;  // M0 = LDS address of the wave's 1 KB destination; the lane's 16 bytes go to M0 + 16 lane
      if (v < kPImgVecs)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(m0_saved) : "s"(lds_at), "v"(src + 16 * v) : "memory");
    }
  };

  for (int64_t blk = blockIdx.x; blk * tiles_per_block < tiles; blk += gridDim.x) {
    f32x16 dxa[TPW];
    int64_t t0[TPW];
#pragma unroll
    for (int tp = 0; tp < TPW; ++tp) {
      t0[tp] = (blk * tiles_per_block + wv * TPW + tp) * 32;
      const int64_t tok = t0[tp] + c;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const int e0 = 16 * s + 8 * half;
        float4 lo = make_float4(0.f, 0.f, 0.f, 0.f), hi = lo;
        if (tok < a.T && e0 < E) lo = *reinterpret_cast<const float4*>(a.X + tok * E + e0);
        if (tok < a.T && e0 + 4 < E) hi = *reinterpret_cast<const float4*>(a.X + tok * E + e0 + 4);
        const float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        const Frag3 f = split_pack8(v);
        unsigned char* xp = XP + tp * 3 * kPlaneBytes + L.row[s];
        *reinterpret_cast<bf16x8*>(xp) = f.t1;
        *reinterpret_cast<bf16x8*>(xp + kPlaneBytes) = f.t2;
        *reinterpret_cast<bf16x8*>(xp + 2 * kPlaneBytes) = f.t3;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) dxa[tp][r] = 0.f;
    }
    float* part = a.partials + (blk * (int64_t)U) * kPart;  // one partial slot per workgroup
    float dyn[NDY];
    // The lane's token, outputs 16 s + 8 half + i. Loads only (no use of the values here: anything that touches a loaded
    // register waits for it, and eight dependent round trips are what a select per load costs): rows past T and outputs
    // past O read an in-bounds neighbour instead. Outputs >= O meet zero columns of B^T in step 2 and land in columns of
    // dB^T / db that are never read; rows >= T are zeroed by dy_take in the one partial tile.
    auto dy_fetch = [&](int u2, int64_t t02) {
      const BwdMlp& d2 = a.m[a.unit[u2] & 0xff];
      const int W = d2.O;
      const int64_t tok = t02 + c;
      const float* src = d2.dY + (tok < a.T ? tok : 0) * W;
#pragma unroll
      for (int s = 0; s < NS2; ++s)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          const int o = 16 * s + 8 * half + i;
          dyn[8 * s + i] = src[o < W ? o : 0];
        }
    };
    dy_fetch(0, t0[0]);
    __syncthreads();  // the previous block's last unit is done with both image buffers
    stage(0);

    for (int u = 0; u < U; ++u) {
#ifdef PSF_X3P_TRACE
      const bool trace_on = blockIdx.x == 300 && u == 5;
#endif
      PSF_TRACE(0);
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's part of image u has landed
      __syncthreads();  // image u is complete; unit u-1's combine is finished
      PSF_TRACE(1);
      if (u + 1 < U) stage(u + 1);
      const unsigned char* img = img_lds + (u & 1) * kPImgBytes;
      const bool first_block_of_mlp = (a.unit[u] >> 8) == 0;

      f32x16 dA, dBT, dav;  // dav[r] = sum over tokens of G[cd_row(r, half)][tok], the same in every column
      float dbp[NDY];
#pragma unroll
      for (int r = 0; r < 16; ++r) dA[r] = dBT[r] = dav[r] = 0.f;
#pragma unroll
      for (int i = 0; i < NDY; ++i) dbp[i] = 0.f;
#pragma unroll
      for (int tp = 0; tp < TPW; ++tp) {
        PSF_TRACE(2 + 8 * tp);
        const unsigned char* xp = XP + tp * 3 * kPlaneBytes;
        // 1. Hpre^T = A_u X^T + a_u : both operands by row reads. In front of the first use of the dY registers: these twelve
        // MFMAs need the image and the X planes only, and cover what is left of the dY loads' (and, behind the partial-sum
        // stores of the previous unit, the stores') round trip.
        f32x16 acc1;
        {
          const float4* sa4 = reinterpret_cast<const float4*>(img + kPOffSa + 64 * half);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float4 v = sa4[q];
            acc1[4 * q] = v.x;
            acc1[4 * q + 1] = v.y;
            acc1[4 * q + 2] = v.z;
            acc1[4 * q + 3] = v.w;
          }
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const Frag3 wa{row_frag(img, L, s), row_frag(img + kPlaneBytes, L, s), row_frag(img + 2 * kPlaneBytes, L, s)};
          const Frag3 xb{row_frag(xp, L, s), row_frag(xp + kPlaneBytes, L, s), row_frag(xp + 2 * kPlaneBytes, L, s)};
          acc1 = mfma6(wa, xb, acc1);
        }
        asm volatile("" : "+v"(acc1) : : "memory");
        // 0. this tile's dY values -> split terms (the lane's step-2 B fragments); start the next tile-unit's loads
        if (t0[tp] + 32 > a.T) {  // wave-uniform: the partial tile (and tiles past the end)
#pragma unroll
          for (int i = 0; i < NDY; ++i) dyn[i] = t0[tp] + c < a.T ? dyn[i] : 0.f;
        }
        Frag3 dy3[NS2];
#pragma unroll
        for (int s = 0; s < NS2; ++s) {
          const float v[8] = {dyn[8 * s], dyn[8 * s + 1], dyn[8 * s + 2], dyn[8 * s + 3],
                              dyn[8 * s + 4], dyn[8 * s + 5], dyn[8 * s + 6], dyn[8 * s + 7]};
          dy3[s] = split_pack8(v);
        }
        if (first_block_of_mlp) {
#pragma unroll
          for (int i = 0; i < NDY; ++i) dbp[i] += dyn[i];
        }
        // Pin the order "consume the old values, THEN issue the next loads". hipcc hoists the loads above the split, and the
        // wait in front of the split is then vmcnt(0) (partial-sum stores are still pending and make the counter's order
        // unknown to it), which drains the loads it has just issued: one memory round trip per tile-unit, measured.
#pragma unroll
        for (int s = 0; s < NS2; ++s)
          asm volatile("" : "+v"(dy3[s].t1), "+v"(dy3[s].t2), "+v"(dy3[s].t3) : : "memory");
#pragma unroll
        for (int i = 0; i < NDY; i += 4) asm volatile("" : "+v"(dbp[i]), "+v"(dbp[i + 1]), "+v"(dbp[i + 2]), "+v"(dbp[i + 3]) : : "memory");
        if (tp + 1 < TPW)
          dy_fetch(u, t0[tp + 1 < TPW ? tp + 1 : 0]);
        else if (u + 1 < U)
          dy_fetch(u + 1, t0[0]);
        if (t0[tp] >= a.T) continue;  // wave-uniform
        PSF_TRACE(3 + 8 * tp);
        // the same terms through the scratch planes [tok][o] and back transposed: the step-4 B operand. Issued first: the
        // round trips overlap with the MFMAs of steps 1 and 2.
        Frag3 yb[2];
        {
          const bf16x8 zero = __builtin_bit_cast(bf16x8, make_uint4(0u, 0u, 0u, 0u));  // outputs >= 16 of a narrow dY
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            *reinterpret_cast<bf16x8*>(HP + L.row[s]) = s < NS2 ? dy3[s < NS2 ? s : 0].t1 : zero;
            *reinterpret_cast<bf16x8*>(YP + L.row[s]) = s < NS2 ? dy3[s < NS2 ? s : 0].t2 : zero;
          }
          asm volatile("" ::: "memory");
#pragma unroll
          for (int s = 0; s < 2; ++s) yb[s].t1 = tr_frag(HP, L, s), yb[s].t2 = tr_frag(YP, L, s);
          asm volatile("" ::: "memory");
#pragma unroll
          for (int s = 0; s < 2; ++s) *reinterpret_cast<bf16x8*>(HP + L.row[s]) = s < NS2 ? dy3[s < NS2 ? s : 0].t3 : zero;
          asm volatile("" ::: "memory");
#pragma unroll
          for (int s = 0; s < 2; ++s) yb[s].t3 = tr_frag(HP, L, s);
          asm volatile("" ::: "memory");
        }

        f32x16 acc3;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc3[r] = 0.f;
        // 2. dHpost^T = B_u^T dY^T : B operand = the lane's own dY fragment
#pragma unroll
        for (int s = 0; s < NS2; ++s)
          acc3 = mfma6(load_frag3(img + kPOffBT + ((s * 2 + half) * 32 + c) * 16, kXFragTerm), dy3[s], acc3);
        // 3. GELU and its derivative
        PSF_TRACE(4 + 8 * tp);
        // Priority by phase: a wave in its vector phase (GELU, splits) yields to the SIMD's other wave whenever that one is in
        // a matrix phase, whose instructions need one issue slot in eight; raised again in front of the matrix steps below.
        // With step 1 ahead of the dY wait: 3 % (profiles/r03ah_x3p_experiments.log).
        __builtin_amdgcn_s_setprio(0);
        float y[16], g[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float dd;
          gelu_and_grad1(acc1[r], y[r], dd);
          g[r] = acc3[r] * dd;  // G = dHpre^T
        }
        // 4. dB^T[j][o] += Hpost^T[j][tok] dY[tok][o]        (both operands by transposed reads)
        // 5. dA[j][e] += G[j][tok] X[tok][e],  da[j] += G[j][tok] 1
        // 6. dX^T[e][tok] += A_u^T[e][j] G[j][tok]              (A^T by transposed reads of the image, G from registers)
        // The six terms of Hpost^T and G go through the two scratch planes in three rounds (LDS operations of one wave
        // execute in order, so a plane is rewritten as soon as its reads are issued). G is split while round 1 is in
        // flight; step 6, which needs no transposed activation, covers round 2, and step 4 covers round 3.
        PSF_TRACE(5 + 8 * tp);
        Frag3 ha[2], ga[2];
        const Split16 ys = split16(y);
        store_acc_plane(HP, L, ys, 0);
        store_acc_plane(YP, L, ys, 1);
        asm volatile("" ::: "memory");
        const Split16 gs = split16(g);
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int s = 0; s < 2; ++s) ha[s].t1 = tr_frag(HP, L, s), ha[s].t2 = tr_frag(YP, L, s);
        asm volatile("" ::: "memory");
        store_acc_plane(HP, L, ys, 2);
        store_acc_plane(YP, L, gs, 0);
        asm volatile("" ::: "memory");
        if (a.dX) {
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            const Frag3 at{tr_frag_acc(img, L, s), tr_frag_acc(img + kPlaneBytes, L, s), tr_frag_acc(img + 2 * kPlaneBytes, L, s)};
            const Frag3 gb{acc_frag(gs, 0, s), acc_frag(gs, 1, s), acc_frag(gs, 2, s)};
            dxa[tp] = mfma6(at, gb, dxa[tp]);
          }
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int s = 0; s < 2; ++s) ha[s].t3 = tr_frag(HP, L, s), ga[s].t1 = tr_frag(YP, L, s);
        asm volatile("" ::: "memory");
        store_acc_plane(HP, L, gs, 1);
        store_acc_plane(YP, L, gs, 2);
        asm volatile("" ::: "memory");
        PSF_TRACE(6 + 8 * tp);
#pragma unroll
        for (int s = 0; s < 2; ++s) dBT = mfma6(ha[s], yb[s], dBT);
#pragma unroll
        for (int s = 0; s < 2; ++s) ga[s].t2 = tr_frag(HP, L, s), ga[s].t3 = tr_frag(YP, L, s);
        asm volatile("" ::: "memory");
        {
          const bf16x8 ones = __builtin_bit_cast(bf16x8, make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u));
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            const Frag3 xb{tr_frag(xp, L, s), tr_frag(xp + kPlaneBytes, L, s), tr_frag(xp + 2 * kPlaneBytes, L, s)};
            dA = mfma6(ga[s], xb, dA);
            dav = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[s].t3, ones, dav, 0, 0, 0);
            dav = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[s].t2, ones, dav, 0, 0, 0);
            dav = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ga[s].t1, ones, dav, 0, 0, 0);
          }
        }
      }
      // db: the per-lane sums over the 32 token lanes of each half (da came off the matrix pipe)
      PSF_TRACE(18);
      if (first_block_of_mlp) {
#pragma unroll
        for (int i = 0; i < NDY; ++i) dbp[i] = half_sum(dbp[i]);
      }
      // Combine the eight waves' partial sums through LDS (fixed order w = 0..7) and flush once per workgroup.
      float* pu = part + (int64_t)u * kPart;
      const float* wave0 = reinterpret_cast<const float*>(wave_lds + TPW * 3 * kPlaneBytes);  // wave 0's SCR
      auto sum8 = [&](int off) {  // off: float offset into a wave's SCR, 16-byte aligned
        float4 acc = *reinterpret_cast<const float4*>(wave0 + off);
#pragma unroll
        for (int w = 1; w < NW; ++w) {
          const float4 v = *reinterpret_cast<const float4*>(wave0 + w * (kWaveBytes / 4) + off);
          acc.x += v.x;
          acc.y += v.y;
          acc.z += v.z;
          acc.w += v.w;
        }
        return acc;
      };
#pragma unroll
      for (int r = 0; r < 16; ++r) SCR[cd_row(r, half) * 32 + c] = dA[r];
      SCR[1024 + lane] = 0.f;  // da [2][32]: the whole sum goes to row 0
      SCR[1088 + lane] = 0.f;  // db [2][32]
      if (c == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) SCR[1024 + cd_row(r, half)] = dav[r];
#pragma unroll
        for (int s = 0; s < NS2; ++s)
#pragma unroll
          for (int i = 0; i < 8; ++i) SCR[1088 + 16 * s + 8 * half + i] = dbp[8 * s + i];
      }
      PSF_TRACE(19);
      __syncthreads();
      PSF_TRACE(20);
      if (wv < 4) *reinterpret_cast<float4*>(pu + wv * 256 + 4 * lane) = sum8(wv * 256 + 4 * lane);
      if (wv == 4 && lane < 32) *reinterpret_cast<float4*>(pu + 2048 + 4 * lane) = sum8(1024 + 4 * lane);
      PSF_TRACE(21);
      __syncthreads();
      PSF_TRACE(22);
#pragma unroll
      for (int r = 0; r < 16; ++r) SCR[cd_row(r, half) * 32 + c] = dBT[r];
      __syncthreads();
      PSF_TRACE(23);
      if (wv < 4) *reinterpret_cast<float4*>(pu + 1024 + wv * 256 + 4 * lane) = sum8(wv * 256 + 4 * lane);
      PSF_TRACE(24);
    }

    if (a.dX) {  // the lane holds dX^T[e = 8 g + 4 half + (0..3)][tok = c] in registers 4 g .. 4 g + 3
#pragma unroll
      for (int tp = 0; tp < TPW; ++tp) {
        const int64_t tok = t0[tp] + c;
        if (tok < a.T) {
#pragma unroll
          for (int gq = 0; gq < 4; ++gq) {
            const int e0 = 8 * gq + 4 * half;
            if (e0 < E)
              *reinterpret_cast<float4*>(a.dX + tok * E + e0) =
                  make_float4(dxa[tp][4 * gq], dxa[tp][4 * gq + 1], dxa[tp][4 * gq + 2], dxa[tp][4 * gq + 3]);
          }
        }
      }
    }
  }
}

// stage 1: R1[s][i] = sum over the groups of slice s of P[g][i]   (i < U*kPart; fixed order: four interleaved running sums
// g = g0 + 0, 1, 2, 3 (mod 4), then ((s0 + s1) + s2) + s3). float4 per thread and four loads in flight per running sum: the
// rolled scalar loop it replaces had one dependent load in flight per thread (49 us for 167 MB at Order N=16384, B=40).
__global__ void __launch_bounds__(256) mlp_bwd_reduce1_k(const BwdArgs a) {
  const int64_t n4 = (int64_t)a.U * kPart / 4;  // kPart % 4 == 0
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n4) return;
  const int s = blockIdx.y;
  const int64_t per = (a.G + a.slices - 1) / a.slices;
  const int64_t g0 = s * per, g1 = g0 + per < a.G ? g0 + per : a.G;
  const float4* __restrict__ P = reinterpret_cast<const float4*>(a.partials);
  float4 acc[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) acc[q] = make_float4(0.f, 0.f, 0.f, 0.f);
  int64_t g = g0;
  for (; g + 4 <= g1; g += 4) {
    float4 v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = P[(g + q) * n4 + i];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      acc[q].x += v[q].x;
      acc[q].y += v[q].y;
      acc[q].z += v[q].z;
      acc[q].w += v[q].w;
    }
  }
  for (int q = 0; g < g1; ++g, ++q) {
    const float4 v = P[g * n4 + i];
    acc[q].x += v.x;
    acc[q].y += v.y;
    acc[q].z += v.z;
    acc[q].w += v.w;
  }
  float4 r;
  r.x = ((acc[0].x + acc[1].x) + acc[2].x) + acc[3].x;
  r.y = ((acc[0].y + acc[1].y) + acc[2].y) + acc[3].y;
  r.z = ((acc[0].z + acc[1].z) + acc[2].z) + acc[3].z;
  r.w = ((acc[0].w + acc[1].w) + acc[2].w) + acc[3].w;
  reinterpret_cast<float4*>(a.stage1)[s * n4 + i] = r;
}

// stage 2: sum the slices and scatter into the unpadded gradient tensors
__global__ void __launch_bounds__(256) mlp_bwd_reduce2_k(const BwdArgs a) {
  const int64_t n = (int64_t)a.U * kPart;
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n) return;
  const int u = (int)(idx / kPart), i = (int)(idx - (int64_t)u * kPart);
  const BwdMlp& d = a.m[a.unit[u] & 0xff];
  const int ht = 32 * (int)(a.unit[u] >> 8), E = a.E;
  auto total = [&](int64_t at) {
    float acc = 0.f;
    for (int s = 0; s < a.slices; ++s) acc += a.stage1[s * n + at];
    return acc;
  };
  if (i < 1024) {
    const int j = i >> 5, e = i & 31;
    if (ht + j < d.h && e < E) d.dA[(ht + j) * E + e] = total(idx);
  } else if (i < 2048) {
    const int j = (i - 1024) >> 5, o = i & 31;
    if (ht + j < d.h && o < d.O) d.dB[o * d.h + ht + j] = total(idx);
  } else if (i < 2048 + 32) {
    const int j = i - 2048;
    if (ht + j < d.h) d.da[ht + j] = total(idx) + total(idx + 32);
  } else if (i >= 2112 && i < 2112 + 32) {
    const int o = i - 2112;
    if (ht == 0 && o < d.O) d.db[o] = total(idx) + total(idx + 32);
  }
}

struct Plan {
  int U;
  int tpw8;  // tiles per wave: 2, or 1 for short inputs (keeps >= 2 workgroups per CU in flight) and for outputs wider than 16
  int64_t G8;
  uint8_t unit_k[kMaxUnits], unit_hb[kMaxUnits];
};

bool make_plan(int64_t T, int32_t E, int32_t K, const int32_t* h, const int32_t* O, Plan* p) {
  if (T < 1 || E < 4 || E > 32 || (E & 3) || K < 1 || K > kMaxMlps || !h || !O) return false;
  p->U = 0;
  for (int k = 0; k < K; ++k) {
    if (h[k] < 1 || h[k] > 128 || O[k] < 1 || O[k] > 32) return false;
    for (int hb = 0; hb * 32 < h[k]; ++hb) {
      p->unit_k[p->U] = (uint8_t)k;
      p->unit_hb[p->U] = (uint8_t)hb;
      ++p->U;
    }
  }
  const int64_t tiles = (T + 31) / 32;
  const int tpw = (tiles + 7) / 8 >= 512 ? 2 : 1;
  // one workgroup (= one partial slot) per 8 waves x TPW tiles
  int max_o = 1;
  for (int k = 0; k < K; ++k) max_o = O[k] > max_o ? O[k] : max_o;
  // 512-thread kernels: outputs wider than 16 double the dY registers of a lane and two tiles per wave no longer fit 256
  // registers (70 spilled, 0.417 ms at 4 x O = 32, T = 655 k); one tile per wave fits: 0.397 ms.
  p->tpw8 = max_o > 16 ? 1 : tpw;
  p->G8 = (tiles + 8 * p->tpw8 - 1) / (8 * p->tpw8);
  return true;
}

int64_t workspace_floats(const Plan& p) {
  const int64_t slots = p.G8;
  return (int64_t)p.U * kImgFloatsMax + slots * p.U * kPart + (int64_t)kSlices * p.U * kPart;
}

}  // namespace

extern "C" {

int64_t psf_mlp_bwd_workspace(int64_t T, int32_t E, int32_t K, const int32_t* h, const int32_t* O) {
  Plan p;
  if (!make_plan(T, E, K, h, O, &p)) return -1;
  return workspace_floats(p) * (int64_t)sizeof(float);
}

int psf_mlp_bwd_f32(const float* X, int64_t T, int32_t E, int32_t K, const float* const* A, const float* const* a,
                    const float* const* B, const int32_t* h, const int32_t* O, const float* const* dY, float* dX,
                    float* const* dA, float* const* da, float* const* dB, float* const* db, void* workspace,
                    int64_t workspace_bytes, void* stream) {
  if (!X || !A || !a || !B || !h || !O || !dY || !dA || !da || !dB || !db || !workspace)
    return psf_internal_fail(PSF_E_NULL, "psf_mlp_bwd: NULL argument");
  Plan p;
  if (!make_plan(T, E, K, h, O, &p))
    return psf_internal_fail(PSF_E_SHAPE, "psf_mlp_bwd: need T >= 1, E in {4,8,...,32}, 1 <= K <= 32, 1 <= h <= 128, 1 <= O <= 32");
  if ((reinterpret_cast<uintptr_t>(X) & 15) != 0) return psf_internal_fail(PSF_E_ALIGN, "psf_mlp_bwd: X must be 16-byte aligned");
  if (workspace_bytes < workspace_floats(p) * (int64_t)sizeof(float) || (reinterpret_cast<uintptr_t>(workspace) & 15) != 0)
    return psf_internal_fail(PSF_E_SHAPE, "psf_mlp_bwd: workspace too small (psf_mlp_bwd_workspace) or not 16-byte aligned");
  BwdArgs args;
  for (int k = 0; k < kMaxMlps; ++k) args.m[k] = BwdMlp{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0};
  for (int k = 0; k < K; ++k) {
    if (!A[k] || !a[k] || !B[k] || !dY[k] || !dA[k] || !da[k] || !dB[k] || !db[k])
      return psf_internal_fail(PSF_E_NULL, "psf_mlp_bwd: NULL layer pointer");
    args.m[k] = BwdMlp{A[k], a[k], B[k], dY[k], dA[k], da[k], dB[k], db[k], h[k], O[k]};
  }
  for (int u = 0; u < kMaxUnits; ++u) {
    args.unit[u] = u < p.U ? ((uint32_t)p.unit_k[u] | ((uint32_t)p.unit_hb[u] << 8)) : 0u;
  }
  int max_o = 1;
  for (int k = 0; k < K; ++k) max_o = O[k] > max_o ? O[k] : max_o;
  if (dX && (reinterpret_cast<uintptr_t>(dX) & 15) != 0)
    return psf_internal_fail(PSF_E_ALIGN, "psf_mlp_bwd: dX must be 16-byte aligned");
  float* ws = reinterpret_cast<float*>(workspace);
  args.X = X;
  args.dX = dX;
  args.images = ws;
  args.partials = ws + (int64_t)p.U * kImgFloatsMax;
  args.stage1 = args.partials + p.G8 * p.U * kPart;
  args.T = T;
  args.G = p.G8;
  // stage-1 slices: about sixteen partial slots each, at most kSlices. With 64 slices for the 128 slots of a 32 k-token input
  // (CIFAR-10) stage 1 was 1 536 workgroups adding two numbers each and stage 2 read 64 slices per element: 19 + 7.6 us
  // beside a 65 us kernel. The Order / Adding shapes (1 280 slots) keep 64 slices and their bits.
  args.slices = (int32_t)((p.G8 + 15) / 16 < 1 ? 1 : ((p.G8 + 15) / 16 > kSlices ? kSlices : (p.G8 + 15) / 16));
  args.E = E;
  args.K = K;
  args.U = p.U;
  hipStream_t s = reinterpret_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(mlp_bwd_x3p_pack_k, dim3(p.U), dim3(256), 0, s, args);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return psf_internal_fail((int)e, hipGetErrorString(e));

  // every slot of the partial buffer is written exactly once: one workgroup per waves*TPW tiles
  const int64_t blocks = args.G;
  if (blocks > 0x7fffffff) return psf_internal_fail(PSF_E_SHAPE, "psf_mlp_bwd: T too large");
  auto launch_planes = [&](auto kernel, int waves) {  // static LDS: 2 images + 8 x (tpw x 3 planes + scratch) = 160,000 B at tpw = 2
    hipLaunchKernelGGL(kernel, dim3((unsigned)blocks), dim3(64 * waves), 0, s, args);
  };
  if (max_o <= 16) {
    if (p.tpw8 == 2) launch_planes(mlp_bwd_x3p_k<2, 8>, 8);
    else launch_planes(mlp_bwd_x3p_k<1, 8>, 8);
  } else {
    launch_planes(mlp_bwd_x3p_k<1, 16>, 8);  // make_plan: one tile per wave when an output is wider than 16
  }
  e = hipGetLastError();
  if (e != hipSuccess) return psf_internal_fail((int)e, hipGetErrorString(e));
  const int64_t n = (int64_t)p.U * kPart;
  hipLaunchKernelGGL(mlp_bwd_reduce1_k, dim3((unsigned)((n / 4 + 255) / 256), (unsigned)args.slices), dim3(256), 0, s, args);
  hipLaunchKernelGGL(mlp_bwd_reduce2_k, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, args);
  e = hipGetLastError();
  return e == hipSuccess ? PSF_OK : psf_internal_fail((int)e, hipGetErrorString(e));
}

}  // extern "C"
