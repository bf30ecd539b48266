// mixer_lds_launch.h — host-side interface of the single-launch mixer for short sequences (mixer_lds.h), for psf_chord.hip.
#pragma once

#include "fwd_mlp_step_launch.h"

namespace psf {

constexpr int kMixerLdsMaxSteps = 32;

struct MixerLdsArgs {
  MixerIn in;
  const unsigned char* images;                // unit images of all M + 1 MLPs (g first), mlp_x3_image.h
  int32_t first_unit[kMixerLdsMaxSteps + 2];  // first unit of MLP k; [M + 1] = total
  float* V0;                                  // [B,N,C] receives g(data), or nullptr
  float* out[kMixerLdsMaxSteps];              // step results; written where bit m of store_mask is set
  uint32_t store_mask;
  int32_t M, N, C, E, L, CG, WS, TT, nu_max;
};

struct MixerLdsPlan {
  int threads, rows, lds_bytes, WS, nu_max;
};

// false when the shape is outside the kernel's limits; fills *p otherwise. nu_max = the most hidden units of any MLP.
bool plan_mixer_lds(int64_t N, int64_t C, int32_t L, int32_t M, int32_t nu_max, MixerLdsPlan* p);
hipError_t launch_mixer_lds(const MixerLdsPlan& p, bool res, const MixerLdsArgs& a, const Offsets& offs, int B, hipStream_t s);

}  // namespace psf
