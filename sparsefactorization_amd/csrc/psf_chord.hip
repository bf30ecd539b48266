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
#include "fwd_kernels.h"

using namespace psf;

// ------------------------------------------------------------------------------------------------------
// errors, tuning
// ------------------------------------------------------------------------------------------------------
namespace {

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
std::atomic<int> g_fwd_rows{0};     // 0 = table default, else rows per thread (R) of the window kernel
std::atomic<int> g_fwd_dma{1};      // 1 = stage W tile / V window with LDS-DMA, 0 = through registers

struct Knob {
  const char* key;
  std::atomic<int>* var;
  int lo, hi;
};
Knob g_knobs[] = {
    {"fwd_variant", &g_fwd_variant, 0, 2},
    {"bwd_variant", &g_bwd_variant, 0, 1},
    {"xcd_remap", &g_xcd_remap, 0, 1},
    {"fwd_rows", &g_fwd_rows, 0, 8},
    {"fwd_dma", &g_fwd_dma, 0, 1},
};

int ceil_log2(int64_t x) {
  int s = 0;
  while (((int64_t)1 << s) < x) ++s;
  return s;
}

// Reduce the caller's offsets (or the chord pattern) into [0, N).
int make_offsets(int64_t N, int32_t L, const int64_t* offsets, Offsets* out) {
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
  return PSF_OK;
}

int check_dims(int64_t B, int64_t N, int32_t L, int64_t C, int64_t v_batch_stride) {
  if (B < 0 || N < 1 || L < 1 || C < 1) return fail(PSF_E_SHAPE, "need B >= 0, N >= 1, L >= 1, C >= 1 (got B=%lld N=%lld L=%d C=%lld)", (long long)B, (long long)N, (int)L, (long long)C);
  if (L > PSF_MAX_LINKS) return fail(PSF_E_SHAPE, "L=%d exceeds PSF_MAX_LINKS=%d", (int)L, PSF_MAX_LINKS);
  if (N > (int64_t)1 << 30) return fail(PSF_E_SHAPE, "N=%lld exceeds 2^30", (long long)N);
  if (C > (int64_t)1 << 30 || N * C > (int64_t)1 << 40) return fail(PSF_E_SHAPE, "N*C too large");
  if (v_batch_stride != 0 && v_batch_stride != N * C) return fail(PSF_E_SHAPE, "v_batch_stride must be 0 (broadcast) or N*C=%lld, got %lld", (long long)(N * C), (long long)v_batch_stride);
  return PSF_OK;
}

bool aligned_to(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) & (a - 1)) == 0; }

// Geometry of the generic (row, channel-group) decomposition.
int make_geom(int64_t B, int64_t N, int32_t L, int64_t C, int vec, int tg_shift, int TR, bool split_channels,
              int64_t v_bstride, Geom* gm) {
  gm->N = (int32_t)N;
  gm->L = L;
  gm->C = (int32_t)C;
  gm->CG = (int32_t)((C + vec - 1) / vec);
  gm->tg_shift = tg_shift;
  gm->TR = TR;
  gm->tiles_n = (int32_t)((N + TR - 1) / TR);
  const int TG = 1 << tg_shift;
  gm->chunks_c = split_channels ? (gm->CG + TG - 1) / TG : 1;
  gm->per_b = gm->tiles_n * gm->chunks_c;
  const int64_t nb = B * (int64_t)gm->per_b;
  if (nb > (int64_t)0x7fffffff) return fail(PSF_E_SHAPE, "launch of %lld workgroups exceeds the grid limit", (long long)nb);
  gm->nblocks = (uint32_t)nb;
  gm->xq = gm->nblocks / kXcds;
  gm->xr = gm->nblocks % kXcds;
  gm->remap = g_xcd_remap.load();
  gm->v_bstride = v_bstride;
  return PSF_OK;
}

// ------------------------------------------------------------------------------------------------------
// window-kernel instance table
// ------------------------------------------------------------------------------------------------------
constexpr int kWinLmin = 4, kWinLmax = 20;

// default rows per thread by channel-group shift (C = 4 << TGS): small rows want long tiles for a deep
// near window; wide rows already move >= 512 B per row and want more rows per thread.
// (measured r01, cfg2 C=8: R=2 32.7 us vs R=1 34.3 us vs R=4 41.8 us per launch)
constexpr int default_rows(int tgs) { return tgs == 1 ? 2 : (tgs <= 2 ? 1 : (tgs <= 4 ? 2 : 4)); }

template <typename T, int L, int TGS, int R, bool DMA>
hipError_t launch_win_dma(const T* W, const T* V, const T* res, T* out, const Geom& gm, const Offsets& offs,
                          int64_t w_total, hipStream_t s) {
  using Cfg = FwdWinCfg<T, L, TGS, R>;
  auto kern = chord_fwd_win_k<T, L, TGS, R, DMA>;
  if (Cfg::lds_bytes > 48 * 1024) {
    static std::atomic<int> done{0};
    if (!done.load()) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                         hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::lds_bytes);
      if (e != hipSuccess) return e;
      done.store(1);
    }
  }
  hipLaunchKernelGGL(kern, dim3(gm.nblocks), dim3(kBlock), Cfg::lds_bytes, s, W, V, res, out, gm, offs, w_total);
  return hipGetLastError();
}

template <typename T, int L, int TGS, int R>
hipError_t launch_win(const T* W, const T* V, const T* res, T* out, const Geom& gm, const Offsets& offs,
                      int64_t w_total, hipStream_t s) {
  return g_fwd_dma.load() ? launch_win_dma<T, L, TGS, R, true>(W, V, res, out, gm, offs, w_total, s)
                          : launch_win_dma<T, L, TGS, R, false>(W, V, res, out, gm, offs, w_total, s);
}

struct WinPick {
  int tgs, rows, TR, KN;
};

template <typename T, int TGS, int R>
hipError_t launch_win_L(int L, const T* W, const T* V, const T* res, T* out, const Geom& gm,
                        const Offsets& offs, int64_t w_total, hipStream_t s) {
  switch (L) {
#define PSF_CASE(LL) \
  case LL:           \
    return launch_win<T, LL, TGS, R>(W, V, res, out, gm, offs, w_total, s);
    PSF_CASE(4) PSF_CASE(5) PSF_CASE(6) PSF_CASE(7) PSF_CASE(8) PSF_CASE(9) PSF_CASE(10) PSF_CASE(11)
    PSF_CASE(12) PSF_CASE(13) PSF_CASE(14) PSF_CASE(15) PSF_CASE(16) PSF_CASE(17) PSF_CASE(18)
    PSF_CASE(19) PSF_CASE(20)
#undef PSF_CASE
    default:
      return hipErrorInvalidValue;
  }
}

// which (TGS, R) pairs are compiled
constexpr bool win_compiled(int tgs, int r) {
  if (tgs < 0 || tgs > 6) return false;
  if (r == default_rows(tgs)) return true;
  // tuning alternates
  if (tgs == 1) return r == 1 || r == 2 || r == 4;
  if (tgs == 3) return r == 1 || r == 2 || r == 4;
  if (tgs == 5) return r == 2 || r == 4 || r == 8;
  return false;
}

template <typename T>
hipError_t launch_win_any(int tgs, int rows, int L, const T* W, const T* V, const T* res, T* out,
                          const Geom& gm, const Offsets& offs, int64_t w_total, hipStream_t s) {
#define PSF_WIN(TGS, R) \
  if (tgs == TGS && rows == R) return launch_win_L<T, TGS, R>(L, W, V, res, out, gm, offs, w_total, s);
  PSF_WIN(0, 1)
  PSF_WIN(1, 1) PSF_WIN(1, 2) PSF_WIN(1, 4)
  PSF_WIN(2, 1)
  PSF_WIN(3, 1) PSF_WIN(3, 2) PSF_WIN(3, 4)
  PSF_WIN(4, 2)
  PSF_WIN(5, 2) PSF_WIN(5, 4) PSF_WIN(5, 8)
  PSF_WIN(6, 4)
#undef PSF_WIN
  return hipErrorInvalidValue;
}

// Decide whether the window kernel applies; fills pick on success.
template <typename T>
bool pick_window(int64_t N, int32_t L, int64_t C, const Offsets& offs, bool vec_ok, WinPick* pick) {
  if (sizeof(T) != 4 || !vec_ok) return false;
  if (L < kWinLmin || L > kWinLmax) return false;
  const int64_t CG = C / 4;
  const int tgs = ceil_log2(CG) > 6 ? 6 : ceil_log2(CG);
  int rows = g_fwd_rows.load();
  if (rows == 0 || !win_compiled(tgs, rows)) rows = default_rows(tgs);
  const int TR = (kBlock >> tgs) * rows;
  if (N < 2 * (int64_t)TR) return false;  // window may wrap at most once
  const int KN = imin(L, ilog2_floor(TR) + 2);
  for (int k = 0; k < KN; ++k)
    if (offs.v[k] > TR) return false;  // near links must fall inside the window
  pick->tgs = tgs;
  pick->rows = rows;
  pick->TR = TR;
  pick->KN = KN;
  return true;
}

// ------------------------------------------------------------------------------------------------------
// typed entry points
// ------------------------------------------------------------------------------------------------------
template <typename T>
int fwd_impl(const T* W, const T* V, const T* res, T* out, int64_t B, int64_t N, int32_t L, int64_t C,
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
  const bool vec_ok = (C % VECW == 0) && aligned_to(V, 16) && aligned_to(out, 16) && (!res || aligned_to(res, 16));

  const int variant = g_fwd_variant.load();
  if constexpr (sizeof(T) == 4) {  // the window kernel is compiled for f32 only (f64 exists for gradcheck)
    WinPick pick;
    if (variant != 1 && pick_window<T>(N, L, C, offs, vec_ok, &pick)) {
      Geom gm;
      if (int rc = make_geom(B, N, L, C, VECW, pick.tgs, pick.TR, true, v_batch_stride, &gm)) return rc;
      hipError_t e = launch_win_any<T>(pick.tgs, pick.rows, L, W, V, res, out, gm, offs, B * N * (int64_t)L, s);
      if (e != hipSuccess) return fail_hip(e, "chord_fwd_win launch");
      return PSF_OK;
    }
  }
  if (variant == 2) return fail(PSF_E_TUNING, "fwd_variant=2 forced but the window kernel does not apply to N=%lld L=%d C=%lld", (long long)N, (int)L, (long long)C);

  const int vec = vec_ok ? VECW : 1;
  const int64_t CG = (C + vec - 1) / vec;
  const int tgs = ceil_log2(CG) > 6 ? 6 : ceil_log2(CG);
  Geom gm;
  if (int rc = make_geom(B, N, L, C, vec, tgs, kBlock >> tgs, true, v_batch_stride, &gm)) return rc;
  if (vec_ok)
    hipLaunchKernelGGL((chord_fwd_generic_k<T, VECW>), dim3(gm.nblocks), dim3(kBlock), 0, s, W, V, res, out, gm, offs);
  else
    hipLaunchKernelGGL((chord_fwd_generic_k<T, 1>), dim3(gm.nblocks), dim3(kBlock), 0, s, W, V, res, out, gm, offs);
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail_hip(e, "chord_fwd_generic launch");
  return PSF_OK;
}

template <typename T>
int bwd_impl(const T* dZ, const T* W, const T* V, T* dW, T* dV, int64_t B, int64_t N, int32_t L, int64_t C,
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

  if (dV) {
    const bool vec_ok = (C % VECW == 0) && aligned_to(dZ, 16) && aligned_to(dV, 16);
    const int vec = vec_ok ? VECW : 1;
    const int64_t CG = (C + vec - 1) / vec;
    const int tgs = ceil_log2(CG) > 6 ? 6 : ceil_log2(CG);
    Geom gm;
    if (int rc = make_geom(B, N, L, C, vec, tgs, kBlock >> tgs, true, N * C, &gm)) return rc;
    if (vec_ok)
      hipLaunchKernelGGL((chord_dv_generic_k<T, VECW>), dim3(gm.nblocks), dim3(kBlock), 0, s, dZ, W, dV, gm, offs);
    else
      hipLaunchKernelGGL((chord_dv_generic_k<T, 1>), dim3(gm.nblocks), dim3(kBlock), 0, s, dZ, W, dV, gm, offs);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail_hip(e, "chord_dv_generic launch");
  }
  if (dW) {
    const bool vec_ok = (C % VECW == 0) && aligned_to(dZ, 16) && aligned_to(V, 16);
    const int vec = vec_ok ? VECW : 1;
    const int64_t CG = (C + vec - 1) / vec;
    const int tgs = ceil_log2(CG) > 6 ? 6 : ceil_log2(CG);
    Geom gm;
    if (int rc = make_geom(B, N, L, C, vec, tgs, kBlock >> tgs, false, v_batch_stride, &gm)) return rc;
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
int chain_impl(const T* const* W_steps, const T* V0, T* const* out_steps, int32_t M, int32_t use_residual,
               int64_t B, int64_t N, int32_t L, int64_t C, int64_t v0_batch_stride, const int64_t* offsets,
               void* stream) {
  if (M < 0) return fail(PSF_E_SHAPE, "M must be >= 0");
  if (M == 0) return PSF_OK;
  if (!W_steps || !out_steps || !V0) return fail(PSF_E_NULL, "W_steps, out_steps and V0 must be non-NULL");
  if (use_residual && v0_batch_stride == 0 && B != 1) return fail(PSF_E_SHAPE, "a broadcast V0 cannot be the residual");
  for (int m = 0; m < M; ++m) {
    if (!W_steps[m] || !out_steps[m]) return fail(PSF_E_NULL, "step %d: NULL pointer", m);
    if (use_residual && out_steps[m] == V0) return fail(PSF_E_ALIAS, "step %d: out aliases the residual V0", m);
  }
  for (int m = 0; m < M; ++m) {
    const T* in = m == 0 ? V0 : out_steps[m - 1];
    const int64_t stride = m == 0 ? v0_batch_stride : N * C;
    int rc = fwd_impl<T>(W_steps[m], in, use_residual ? V0 : nullptr, out_steps[m], B, N, L, C, stride, offsets, stream);
    if (rc) return rc;
  }
  return PSF_OK;
}

}  // namespace

// ------------------------------------------------------------------------------------------------------
// extern "C"
// ------------------------------------------------------------------------------------------------------
extern "C" {

int psf_version(void) { return PSF_ABI_VERSION; }

const char* psf_last_error(void) { return g_err; }

const char* psf_build_info(void) {
  return "libpsf_chord: gfx950 (CDNA4, wave64) | hipcc " __VERSION__
         " | fwd: generic<f32,f64> + LDS-window<f32, L=4..20, C=4..> | bwd: generic dV/dW<f32,f64>"
         " | arithmetic: uncontracted mul+add, links ascending";
}

int psf_chord_offsets(int64_t N, int32_t L, int64_t* offsets_out) {
  if (!offsets_out) return fail(PSF_E_NULL, "offsets_out is NULL");
  if (N < 1 || L < 1 || L > PSF_MAX_LINKS) return fail(PSF_E_SHAPE, "need N >= 1 and 1 <= L <= %d", PSF_MAX_LINKS);
  if (N > (int64_t)1 << 30) return fail(PSF_E_SHAPE, "N exceeds 2^30");
  Offsets o;
  make_offsets(N, L, nullptr, &o);
  for (int k = 0; k < L; ++k) offsets_out[k] = o.v[k];
  return PSF_OK;
}

int psf_chord_indices(int64_t N, int32_t L, int64_t* rows_out, int64_t* cols_out) {
  if (!rows_out || !cols_out) return fail(PSF_E_NULL, "rows_out / cols_out is NULL");
  if (N < 1 || L < 1 || L > PSF_MAX_LINKS) return fail(PSF_E_SHAPE, "need N >= 1 and 1 <= L <= %d", PSF_MAX_LINKS);
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
  return fwd_impl<float>(W, V, res, out, B, N, L, C, v_batch_stride, offsets, stream);
}
int psf_chord_spmm_fwd_f64(const double* W, const double* V, const double* res, double* out, int64_t B,
                           int64_t N, int32_t L, int64_t C, int64_t v_batch_stride, const int64_t* offsets,
                           void* stream) {
  return fwd_impl<double>(W, V, res, out, B, N, L, C, v_batch_stride, offsets, stream);
}

int psf_chord_spmm_bwd_f32(const float* dZ, const float* W, const float* V, float* dW, float* dV, int64_t B,
                           int64_t N, int32_t L, int64_t C, int64_t v_batch_stride, const int64_t* offsets,
                           void* stream) {
  return bwd_impl<float>(dZ, W, V, dW, dV, B, N, L, C, v_batch_stride, offsets, stream);
}
int psf_chord_spmm_bwd_f64(const double* dZ, const double* W, const double* V, double* dW, double* dV,
                           int64_t B, int64_t N, int32_t L, int64_t C, int64_t v_batch_stride,
                           const int64_t* offsets, void* stream) {
  return bwd_impl<double>(dZ, W, V, dW, dV, B, N, L, C, v_batch_stride, offsets, stream);
}

int psf_chord_chain_fwd_f32(const float* const* W_steps, const float* V0, float* const* out_steps, int32_t M,
                            int32_t use_residual, int64_t B, int64_t N, int32_t L, int64_t C,
                            int64_t v0_batch_stride, const int64_t* offsets, void* stream) {
  return chain_impl<float>(W_steps, V0, out_steps, M, use_residual, B, N, L, C, v0_batch_stride, offsets, stream);
}
int psf_chord_chain_fwd_f64(const double* const* W_steps, const double* V0, double* const* out_steps, int32_t M,
                            int32_t use_residual, int64_t B, int64_t N, int32_t L, int64_t C,
                            int64_t v0_batch_stride, const int64_t* offsets, void* stream) {
  return chain_impl<double>(W_steps, V0, out_steps, M, use_residual, B, N, L, C, v0_batch_stride, offsets, stream);
}

int psf_set_tuning(const char* key, int32_t value) {
  if (!key) return fail(PSF_E_NULL, "key is NULL");
  for (auto& k : g_knobs)
    if (strcmp(k.key, key) == 0) {
      if (value < k.lo || value > k.hi) return fail(PSF_E_TUNING, "tuning %s: value %d outside [%d, %d]", key, (int)value, k.lo, k.hi);
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
  WinPick pick;
  const int variant = g_fwd_variant.load();
  if (variant != 1 && elem_bytes == 4 && pick_window<float>(N, L, C, offs, vec_ok, &pick)) {
    snprintf(buf, cap, "chord_fwd_win_k<f32,L=%d,TG=%d,R=%d,%s> TR=%d near=%d far=%d", (int)L, 1 << pick.tgs,
             pick.rows, g_fwd_dma.load() ? "dma" : "reg", pick.TR, pick.KN, (int)L - pick.KN);
  } else {
    snprintf(buf, cap, "chord_fwd_generic_k<%s,VEC=%d>", elem_bytes == 4 ? "f32" : "f64", vec_ok ? vecw : 1);
  }
  return PSF_OK;
}

}  // extern "C"
