// mlp_fwd_x3.h — internal interface of the split-bf16 variant of the fused MLP forward (mlp_fwd_x3.hip), used by
// psf_mlp_fwd_f32 / psf_mlp_fwd_workspace in mlp_fwd.hip. Arguments as psf_mlp_fwd_f32 (include/psf_chord.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// bytes of workspace the variant needs, or -1 if it does not cover these sizes (E > 32, ...)
int64_t psf_x3_mlp_fwd_workspace(int32_t E, int32_t K, const int32_t* h, const int32_t* O);

// arguments already validated by the caller; `workspace` 16-byte aligned and at least psf_x3_mlp_fwd_workspace bytes
hipError_t psf_x3_mlp_fwd_launch(const float* X, int64_t T, int32_t E, int32_t K, const float* const* A,
                                 const float* const* a, const float* const* B, const float* const* b, const int32_t* h,
                                 const int32_t* O, float* const* Y, void* workspace, hipStream_t s, bool packed = false);

// Pack only: the unit images (mlp_x3_image.h, 14080 bytes each) of K MLPs into `workspace`, MLP by MLP; first_unit[k]
// (K + 1 entries, host) receives the index of MLP k's first unit. A later psf_x3_mlp_fwd_launch(..., packed = true) on the
// first K' <= K of the same MLPs and the same workspace skips its own packing.
constexpr int kX3ImageBytes = 14080;
hipError_t psf_x3_pack_launch(int32_t E, int32_t K, const float* const* A, const float* const* a, const float* const* B,
                              const float* const* b, const int32_t* h, const int32_t* O, void* workspace,
                              int32_t* first_unit, hipStream_t s);
