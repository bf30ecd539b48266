// fwd_mlp_step_launch.h — host-side interface of the forward step that computes its own W tile (fwd_mlp_step.h), for the
// dispatcher in psf_chord.hip. The kernels are compiled per channel-group shift in fwd_mlp_step_inst.hip.
#pragma once

#include "psf_common.h"

namespace psf {

constexpr int kMlpStepTgsMax = 3;                                   // rows of <= 32 channels
constexpr int kMlpStepLmin = 4, kMlpStepLmax = 20;                  // compiled link counts
constexpr int mlp_step_rows(int tgs) { return tgs == 0 ? 1 : 2; }  // rows per thread: tiles of 256, 256, 128, 64 rows
constexpr int mlp_step_tile_rows(int tgs) { return (256 >> tgs) * mlp_step_rows(tgs); }

// How a row of `data` [B, N, E] is obtained (include/psf_chord.h: psf_mixer_input).
struct MixerIn {
  const void* src;      // kind 0: float X [B,N,E]; kind 1: float in [B,N,K]; kind 2: int64 tokens [B,N]
  const float* weight;  // kind 1: W_i [E,K] (nn.Linear.weight); kind 2: table [V,E] (nn.Embedding.weight)
  const float* bias;    // kind 1: b_i [E] or nullptr
  const float* pos;     // kinds 1, 2: positional rows [N,E] added per position, or nullptr
  int32_t kind, K;      // K: kind 1: inputs per position (<= 3); kind 2: vocabulary size
};

struct FwdMlpArgs {
  MixerIn in;                   // data [B, N, E] or its recipe
  const float* V;               // step input [B, N, C] (or [N, C] with gm.v_bstride == 0)
  const float* res;             // residual [B, N, C] or nullptr
  float* out;                   // [B, N, C]
  const unsigned char* images;  // nu packed unit images of THIS step's MLP (mlp_x3_image.h)
  int32_t nu, E;
  Geom gm;
  Offsets offs;
  bool edge;
  int wg_per_cu;
  int ablate;  // knob "mixer_ablate": timing experiments only (non-zero gives wrong results), see fwd_mlp_step.h
  hipStream_t stream;
};

// Launch the instance (L, TGS). Returns hipErrorInvalidValue when it is not compiled.
template <int TGS>
hipError_t launch_fwd_mlp(int L, const FwdMlpArgs& a);

// V0 = g(data): the same tiles, the matrix phase only, its O = C outputs stored as rows of `out` (a.V, a.res, a.offs unused)
template <int TGS>
hipError_t launch_mixer_g(const FwdMlpArgs& a);

}  // namespace psf
