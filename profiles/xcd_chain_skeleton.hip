// xcd_chain_skeleton.hip — LAB, not product. A bounded experiment on the headline shape (N = 16384, L = 15, C = 8, B = 64):
// what would a forward chain cost whose X stays on chip, every sequence spread over the LDS of G workgroups of ONE XCD
// that exchange their rows through that XCD's L2 between steps?  The skeleton moves the bytes such a kernel would move —
// W streamed once from HBM, the tile written to an exchange buffer, the tiles of the far links read back with
// TCP-bypassing loads, a barrier among the G workgroups of a sequence per step — and does no chord arithmetic.
// It prices the memory system's side of the design before anyone writes the kernel.
//
//   hipcc --offload-arch=gfx950 -O3 -o profiles/bin/xcd_chain_skeleton profiles/xcd_chain_skeleton.hip
//   profiles/bin/xcd_chain_skeleton
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                          \
  do {                                                                                 \
    hipError_t e_ = (x);                                                               \
    if (e_ != hipSuccess) {                                                            \
      std::fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
      std::exit(2);                                                                    \
    }                                                                                  \
  } while (0)

namespace {

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

constexpr int kN = 16384, kL = 15, kC = 8, kSteps = 14, kXcds = 8, kThreads = 1024;
constexpr int kRowBytes = kC * 4, kWRowBytes = kL * 4;
constexpr uint32_t kSpinLimit = 1u << 22;  // every spin ends: a stuck barrier sets `abort` and the grid drains

struct Args {
  const uint8_t* W;   // [steps][B][N][L] f32
  uint8_t* xch;       // [2][B][N][C] f32
  uint32_t* bar;      // one counter per sequence (zeroed before the launch)
  uint32_t* abort;    // set by a barrier that gave up
  uint32_t* xcc;      // [grid] XCC_ID of every workgroup
  float* sink;        // keeps the loads alive
  int B, G, parts;    // G workgroups per sequence; parts: 1 = everything, 2 = W only, 4 = exchange only
  int scope;          // 0: plain loads, 1: sc1 (agent: bypass the TCP), 2: sc0 sc1 (system)
};

__device__ __forceinline__ u32x4 load_x(const uint8_t* p, int scope) {
  u32x4 v;
  if (scope == 1) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  else if (scope == 2) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  else v = *reinterpret_cast<const u32x4*>(p);
  return v;
}

// Four exchange loads in flight, then one wait (a real kernel would keep as many in flight).
__device__ __forceinline__ void load_x4(const uint8_t* p0, const uint8_t* p1, const uint8_t* p2, const uint8_t* p3, int scope,
                                        u32x4& a, u32x4& b, u32x4& c, u32x4& d) {
  if (scope == 1) {
    asm volatile(
        "global_load_dwordx4 %0, %4, off sc1\n\tglobal_load_dwordx4 %1, %5, off sc1\n\t"
        "global_load_dwordx4 %2, %6, off sc1\n\tglobal_load_dwordx4 %3, %7, off sc1\n\ts_waitcnt vmcnt(0)"
        : "=&v"(a), "=&v"(b), "=&v"(c), "=&v"(d)
        : "v"(p0), "v"(p1), "v"(p2), "v"(p3)
        : "memory");
  } else {
    a = load_x(p0, scope);
    b = load_x(p1, scope);
    c = load_x(p2, scope);
    d = load_x(p3, scope);
  }
}

__global__ __launch_bounds__(kThreads) void skeleton_k(Args a) {
  extern __shared__ uint8_t lds[];
  const int tid = threadIdx.x;
  const int xcd = blockIdx.x % kXcds, j = blockIdx.x / kXcds;  // j: 0 .. 31 inside the XCD
  const int G = a.G, T = kN / G;                                // rows of a sequence per workgroup
  const int slots = 32 / G;                                     // sequences an XCD holds at once
  const int q = j % G, slot = j / G;
  if (tid == 0) {
    uint32_t id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    a.xcc[blockIdx.x] = id;
  }
  float acc = 0.f;
  uint32_t epoch = 0;
  const int passes = (a.B + kXcds * slots - 1) / (kXcds * slots);
  for (int pass = 0; pass < passes; ++pass) {
    const int b = (pass * slots + slot) * kXcds + xcd;
    if (b >= a.B) break;  // (whole groups leave together: b depends on slot and xcd only)
    const size_t seq_x = (size_t)b * kN * kRowBytes;
    const size_t x_elems = (size_t)a.B * kN * kRowBytes;
    uint32_t* bar = a.bar + b;
    epoch = 0;
    for (int m = 0; m < kSteps; ++m) {
      // (a) this workgroup's W tile of step m: T rows x 60 B, streamed once
      if (a.parts & 3) {
        const uint8_t* w = a.W + (((size_t)m * a.B + b) * kN + (size_t)q * T) * kWRowBytes;
        const int n16 = T * kWRowBytes / 16;
        for (int i = tid; i < n16; i += 4 * kThreads) {
          u32x4 v0 = *reinterpret_cast<const u32x4*>(w + (size_t)i * 16);
          u32x4 v1 = i + kThreads < n16 ? *reinterpret_cast<const u32x4*>(w + (size_t)(i + kThreads) * 16) : u32x4{0, 0, 0, 0};
          u32x4 v2 = i + 2 * kThreads < n16 ? *reinterpret_cast<const u32x4*>(w + (size_t)(i + 2 * kThreads) * 16) : u32x4{0, 0, 0, 0};
          u32x4 v3 = i + 3 * kThreads < n16 ? *reinterpret_cast<const u32x4*>(w + (size_t)(i + 3 * kThreads) * 16) : u32x4{0, 0, 0, 0};
          acc += __uint_as_float(v0.x ^ v1.y ^ v2.z ^ v3.w);
        }
      }
      if (a.parts & 5) {
        // (b) the tiles the far links reach: q + 1 (the first far link and every overhang), q + 2, q + 4, ... (mod G)
        const uint8_t* src = a.xch + (size_t)(m & 1) * x_elems + seq_x;
        const int n16 = T * kRowBytes / 16;
        for (int d = 1; d < G; d <<= 1) {
          const uint8_t* t = src + (size_t)((q + d) % G) * T * kRowBytes;
          for (int i = tid; i < n16; i += 4 * kThreads) {
            u32x4 v0, v1, v2, v3;
            load_x4(t + (size_t)i * 16, t + (size_t)((i + kThreads) % n16) * 16, t + (size_t)((i + 2 * kThreads) % n16) * 16,
                    t + (size_t)((i + 3 * kThreads) % n16) * 16, a.scope, v0, v1, v2, v3);
            acc += __uint_as_float(v0.x ^ v1.y ^ v2.z ^ v3.w);
          }
        }
        // (c) the new tile, for the others
        uint8_t* dst = a.xch + (size_t)((m + 1) & 1) * x_elems + seq_x + (size_t)q * T * kRowBytes;
        for (int i = tid; i < n16; i += kThreads) {
          u32x4 v = {__float_as_uint(acc), (uint32_t)i, (uint32_t)m, (uint32_t)q};
          *reinterpret_cast<u32x4*>(dst + (size_t)i * 16) = v;
        }
        // (d) barrier among the G workgroups of this sequence: stores acknowledged by L2, then one counter
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        ++epoch;
        if (tid == 0) {
          __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const uint32_t want = epoch * (uint32_t)G;
          uint32_t spins = 0;
          while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
            if (++spins > kSpinLimit || __hip_atomic_load(a.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
              __hip_atomic_store(a.abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              break;
            }
            __builtin_amdgcn_s_sleep(1);
          }
        }
        __syncthreads();
      }
    }
  }
  if (acc == 1234.5f) a.sink[blockIdx.x * kThreads + tid] = acc + lds[tid];
}


// ---- second form: what a real kernel could overlap. The exchange loads of a step all in flight at once (sc1, straight to
// registers), W through a four-deep LDS ring filled by LDS-DMA (7 680 B = 128 rows per chunk) and read back from LDS, the
// first chunks of the NEXT step's W already in flight across the barrier.  parts: 1 W, 2 exchange loads, 4 tile store,
// 8 barrier.
constexpr int kChunkRows = 128, kChunkBytes = kChunkRows * kWRowBytes, kRing = 4;  // 7 680 B = 480 x 16 B

__device__ __forceinline__ void glds16(uint32_t lds_at, const uint8_t* g) {
  uint32_t m0_saved;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %2, off\n\ts_mov_b32 m0, %0"
               : "=&s"(m0_saved)
               : "s"(lds_at), "v"(g)
               : "memory");
}

template <int G>
__global__ __launch_bounds__(kThreads) void skeleton2_k(Args a) {
  extern __shared__ __attribute__((aligned(1024))) uint8_t lds[];
  constexpr int T = kN / G, kSlots = 32 / G, kChunks = T / kChunkRows;
  constexpr int kFar = G == 4 ? 2 : 3;                   // tiles q+1, q+2 (, q+4)
  constexpr int kXv = T * kRowBytes / 16 / kThreads;     // exchange vectors per thread and tile: 8 (G=4), 4 (G=8)
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int xcd = blockIdx.x % kXcds, j = blockIdx.x / kXcds;
  const int q = j % G, slot = j / G;
  const uint32_t lds_base = (uint32_t)(uintptr_t)lds;
  float acc = 0.f;
  const int passes = (a.B + kXcds * kSlots - 1) / (kXcds * kSlots);
  const size_t x_elems = (size_t)a.B * kN * kRowBytes;
  // chunk c of W tile `w` into ring slot c % kRing: waves 0..7 move 480 vectors (wave 7: 32 lanes)
  auto issue_chunk = [&](const uint8_t* w, int c) {
    if (wv < 8) {
      const int v = wv * 64 + lane;
      if (v < kChunkBytes / 16)
        glds16(lds_base + (uint32_t)((c % kRing) * 8192 + wv * 1024), w + (size_t)c * kChunkBytes + (size_t)v * 16);
    }
  };
  for (int pass = 0; pass < passes; ++pass) {
    const int b = (pass * kSlots + slot) * kXcds + xcd;
    if (b >= a.B) break;
    const size_t seq_x = (size_t)b * kN * kRowBytes;
    uint32_t* bar = a.bar + b;
    uint32_t epoch = 0;
    auto w_tile = [&](int m) { return a.W + (((size_t)m * a.B + b) * kN + (size_t)q * T) * kWRowBytes; };
    if (a.parts & 1)
      for (int c = 0; c < kRing - 1; ++c) issue_chunk(w_tile(0), c);
    for (int m = 0; m < kSteps; ++m) {
      u32x4 xr[kFar * kXv];
      if (a.parts & 2) {
        const uint8_t* src = a.xch + (size_t)(m & 1) * x_elems + seq_x;
#pragma unroll
        for (int f = 0; f < kFar; ++f) {
          const uint8_t* t = src + (size_t)((q + (1 << f)) % G) * T * kRowBytes + (size_t)tid * 16;
#pragma unroll
          for (int i = 0; i < kXv; ++i)
            asm volatile("global_load_dwordx4 %0, %1, off sc1" : "=&v"(xr[f * kXv + i]) : "v"(t + (size_t)i * kThreads * 16) : "memory");
        }
      }
      if (a.parts & 1) {
        const uint8_t* w = w_tile(m);
        const uint8_t* wn = w_tile(m + 1 < kSteps ? m + 1 : m);
        for (int c = 0; c < kChunks; ++c) {
          const int ahead = c + kRing - 1;
          if (ahead < kChunks) issue_chunk(w, ahead);
          else if (m + 1 < kSteps) issue_chunk(wn, ahead - kChunks);  // the next step's first chunks, across the barrier
          else if (wv < 8) asm volatile("s_nop 0");
          // chunk c has landed when at most kRing - 1 younger DMAs are outstanding (the waves that issue count theirs)
          if (m + 1 < kSteps || ahead < kChunks) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
          else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          __syncthreads();
          const u32x4 v = *reinterpret_cast<const u32x4*>(lds + (c % kRing) * 8192 + (tid & 511) * 16 % kChunkBytes);
          acc += __uint_as_float(v.x ^ v.w);
          __syncthreads();
        }
      }
      if (a.parts & 2) {
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(0) : "memory");
#pragma unroll
        for (int i = 0; i < kFar * kXv; ++i) {
          asm volatile("" : "+v"(xr[i]));
          acc += __uint_as_float(xr[i].x ^ xr[i].z);
        }
      }
      if (a.parts & 4) {
        uint8_t* dst = a.xch + (size_t)((m + 1) & 1) * x_elems + seq_x + (size_t)q * T * kRowBytes + (size_t)tid * 16;
#pragma unroll
        for (int i = 0; i < kXv; ++i) {
          u32x4 v = {__float_as_uint(acc), (uint32_t)i, (uint32_t)m, (uint32_t)q};
          *reinterpret_cast<u32x4*>(dst + (size_t)i * kThreads * 16) = v;
        }
      }
      if (a.parts & 8) {
        // stores acknowledged by L2 (the W DMAs in flight for the next step are younger: a counted wait would do in a
        // real kernel; the skeleton waits for everything on the waves that store, which is every wave)
        if (a.parts & 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        ++epoch;
        if (tid == 0) {
          __hip_atomic_fetch_add(bar, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          const uint32_t want = epoch * (uint32_t)G;
          uint32_t spins = 0;
          while (__hip_atomic_load(bar, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) {
            if (++spins > kSpinLimit || __hip_atomic_load(a.abort, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
              __hip_atomic_store(a.abort, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
              break;
            }
          }
        }
        __syncthreads();
      }
    }
  }
  if (acc == 1234.5f) a.sink[blockIdx.x * kThreads + tid] = acc;
}

}  // namespace

int main(int argc, char** argv) {
  const int B = argc > 1 ? std::atoi(argv[1]) : 64;
  CK(hipSetDevice(0));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  std::printf("%s, %d CUs, cooperative launch %d\n", prop.gcnArchName, prop.multiProcessorCount, prop.cooperativeLaunch);
  const int grid = 256;
  if (prop.multiProcessorCount < grid) {
    std::printf("needs %d CUs\n", grid);
    return 0;
  }
  const size_t w_bytes = (size_t)kSteps * B * kN * kWRowBytes, x_bytes = (size_t)2 * B * kN * kRowBytes;
  uint8_t *W, *xch;
  uint32_t *bar, *abort_flag, *xcc;
  float* sink;
  CK(hipMalloc(&W, w_bytes));
  CK(hipMalloc(&xch, x_bytes));
  CK(hipMalloc(&bar, 4 * B));
  CK(hipMalloc(&abort_flag, 4));
  CK(hipMalloc(&xcc, 4 * grid));
  CK(hipMalloc(&sink, 4 * grid * kThreads));
  CK(hipMemset(W, 1, w_bytes));
  CK(hipMemset(xch, 0, x_bytes));
  CK(hipMemset(abort_flag, 0, 4));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));

  const int lds_bytes = 124 * 1024;  // what the real kernel would hold: forces one workgroup per CU
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(skeleton_k), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
  int per_cu = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, skeleton_k, kThreads, lds_bytes));
  std::printf("workgroups per CU at %d KB LDS: %d\n", lds_bytes / 1024, per_cu);

  bool placement_ok = true;
  const char* part_name[] = {"", "W + exchange + barrier", "W only", "", "exchange + barrier only"};
  for (int G : {8, 4}) {
    for (int parts : {1, 2, 4}) {
      for (int scope : {1, 0, 2}) {
        if (parts == 2 && scope != 1) continue;
        Args a{W, xch, bar, abort_flag, xcc, sink, B, G, parts, scope};
        void* params[] = {&a};
        float best = 1e30f;
        for (int rep = 0; rep < 6; ++rep) {
          CK(hipMemsetAsync(bar, 0, 4 * B, st));
          CK(hipEventRecord(e0, st));
          CK(hipLaunchCooperativeKernel(reinterpret_cast<const void*>(skeleton_k), dim3(grid), dim3(kThreads), params, lds_bytes, st));
          CK(hipEventRecord(e1, st));
          CK(hipStreamSynchronize(st));
          float ms;
          CK(hipEventElapsedTime(&ms, e0, e1));
          if (rep > 0 && ms < best) best = ms;
        }
        uint32_t ab = 0;
        CK(hipMemcpy(&ab, abort_flag, 4, hipMemcpyDeviceToHost));
        std::printf("G=%d (T=%5d rows, %d sequences per XCD at once)  %-26s scope %d: %8.1f us per chain = %6.2f us per step%s\n", G,
                    kN / G, 32 / G, part_name[parts], scope, best * 1e3, best * 1e3 / kSteps, ab ? "  ** barrier gave up **" : "");
        if (ab) {
          CK(hipMemset(abort_flag, 0, 4));
        }
      }
    }
    std::vector<uint32_t> ids(grid);
    CK(hipMemcpy(ids.data(), xcc, 4 * grid, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < grid; ++i) bad += (int)(ids[i] & 15u) != (int)(ids[i % kXcds] & 15u);
    std::printf("placement: workgroup i on the XCD of workgroup i mod 8 for %d of %d; XCC_ID of workgroups 0..7:", grid - bad, grid);
    for (int i = 0; i < kXcds; ++i) std::printf(" %u", ids[i] & 15u);
    std::printf("\n");
    placement_ok = placement_ok && bad == 0;
  }

  std::printf("\nsecond form (everything a real kernel could overlap):\n");
  const char* p2[] = {"W", "exchange loads", "tile store", "barrier"};
  for (int G : {4, 8}) {
    const void* k = G == 4 ? reinterpret_cast<const void*>(skeleton2_k<4>) : reinterpret_cast<const void*>(skeleton2_k<8>);
    CK(hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    for (int parts : {15, 1, 2, 4, 8, 6, 14, 7}) {
      Args a{W, xch, bar, abort_flag, xcc, sink, B, G, parts, 1};
      void* params[] = {&a};
      float best = 1e30f;
      for (int rep = 0; rep < 6; ++rep) {
        CK(hipMemsetAsync(bar, 0, 4 * B, st));
        CK(hipEventRecord(e0, st));
        CK(hipLaunchCooperativeKernel(k, dim3(grid), dim3(kThreads), params, lds_bytes, st));
        CK(hipEventRecord(e1, st));
        CK(hipStreamSynchronize(st));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best) best = ms;
      }
      uint32_t ab = 0;
      CK(hipMemcpy(&ab, abort_flag, 4, hipMemcpyDeviceToHost));
      std::printf("G=%d parts", G);
      for (int i = 0; i < 4; ++i)
        if (parts >> i & 1) std::printf(" [%s]", p2[i]);
      std::printf(": %8.1f us per chain = %6.2f us per step%s\n", best * 1e3, best * 1e3 / kSteps, ab ? "  ** barrier gave up **" : "");
      if (ab) CK(hipMemset(abort_flag, 0, 4));
    }
  }
  std::printf("for scale: the shipped chain takes 25.8 us per step (14 launches of chord_fwd_win_k)\n");
  return placement_ok ? 0 : 1;
}
