// mixer_lds_inst.hip — instances and launcher of the single-launch mixer for short sequences (mixer_lds.h).
#include <atomic>

#include "mixer_lds.h"

namespace psf {

bool plan_mixer_lds(int64_t N, int64_t C, int32_t L, int32_t M, int32_t nu_max, MixerLdsPlan* p) {
  if (N < 32 || N % 32 != 0 || (C != 4 && C != 8) || L < 2 || L > 20 || M < 1 || M > kMixerLdsMaxSteps || nu_max < 1 || nu_max > 4)
    return false;
  const int64_t CG = C / 4, slots = N * CG;
  if (slots > 1024) return false;
  const int R = slots <= 256 ? 1 : 2;
  int64_t threads = (slots + R - 1) / R;
  threads = (threads + 63) / 64 * 64;
  if (threads > 512 || threads % CG != 0) return false;
  if (N / 32 > 2 * (threads / 64)) return false;  // at most two token tiles per wave
  p->threads = (int)threads;
  p->rows = R;
  p->WS = mixer_lds_ws(L);
  p->nu_max = nu_max;
  p->lds_bytes = (int)(2 * slots * 16 + N * p->WS * 4 + (int64_t)nu_max * kImgBytes + 512);
  return p->lds_bytes <= 160 * 1024;
}

namespace {

template <bool RES, int KIND>
hipError_t launch_one(const MixerLdsPlan& p, const MixerLdsArgs& a, const Offsets& offs, int B, hipStream_t s) {
  auto kern = chord_mixer_lds_k<RES, KIND>;
  if (p.lds_bytes > 48 * 1024) {
    static std::atomic<int> done{0};
    if (done.load() < p.lds_bytes) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, p.lds_bytes);
      if (e != hipSuccess) return e;
      done.store(p.lds_bytes);
    }
  }
  hipLaunchKernelGGL(kern, dim3((unsigned)B), dim3(p.threads), p.lds_bytes, s, a, offs);
  return hipGetLastError();
}

template <bool RES>
hipError_t launch_kind(const MixerLdsPlan& p, const MixerLdsArgs& a, const Offsets& offs, int B, hipStream_t s) {
  switch (a.in.kind) {
    case 0: return launch_one<RES, 0>(p, a, offs, B, s);
    case 1: return launch_one<RES, 1>(p, a, offs, B, s);
    case 2: return launch_one<RES, 2>(p, a, offs, B, s);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace

hipError_t launch_mixer_lds(const MixerLdsPlan& p, bool res, const MixerLdsArgs& a, const Offsets& offs, int B, hipStream_t s) {
  return res ? launch_kind<true>(p, a, offs, B, s) : launch_kind<false>(p, a, offs, B, s);
}

}  // namespace psf
