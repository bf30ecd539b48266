// membench.hip — memory ceilings of the box the kernels run on (the denominators next to the 8 TB/s spec).
//   hipcc -O3 --offload-arch=gfx950 profiles/membench.hip -o gpurun_out/membench && gpurun_out/membench
// Reads / writes / copies float4 streams of several footprints: > 256 MiB (HBM), 64-128 MiB (Infinity Cache
// resident when re-read), <= 32 MiB (aggregate L2). Also the mix of the PSF forward step: read 3 streams of
// which two are cache-resident, write one.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x)                                                                      \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));    \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

template <int U>
__global__ void __launch_bounds__(256) k_read(const float4* __restrict__ a, size_t n, float* sink) {
  size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256 * U;
  float4 acc = {0, 0, 0, 0};
  for (; i + (U - 1) * 256 < n; i += stride) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = a[i + u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      acc.x += v[u].x;
      acc.y += v[u].y;
      acc.z += v[u].z;
      acc.w += v[u].w;
    }
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) *sink = acc.x;
}

template <int U>
__global__ void __launch_bounds__(256) k_write(float4* __restrict__ a, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256 * U;
  const float4 v = {1, 2, 3, 4};
  for (; i + (U - 1) * 256 < n; i += stride) {
#pragma unroll
    for (int u = 0; u < U; ++u) a[i + u * 256] = v;
  }
}

template <int U>
__global__ void __launch_bounds__(256) k_copy(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256 * U;
  for (; i + (U - 1) * 256 < n; i += stride) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = a[i + u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u) b[i + u * 256] = v[u];
  }
}

// PSF-step-like mix: out = f(w[2x], v, r): read 63 MB unique (w) + 33.5 MB (v) + 33.5 MB (r), write 33.5 MB.
__global__ void __launch_bounds__(256) k_mix(const float4* __restrict__ w, const float4* __restrict__ v,
                                             const float4* __restrict__ r, float4* __restrict__ o, size_t n) {
  size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * 256;
  for (; i < n; i += stride) {
    const float4 w0 = w[2 * i], w1 = w[2 * i + 1], vv = v[i], rr = r[i];
    float4 x;
    x.x = w0.x * vv.x + w1.x + rr.x;
    x.y = w0.y * vv.y + w1.y + rr.y;
    x.z = w0.z * vv.z + w1.z + rr.z;
    x.w = w0.w * vv.w + w1.w + rr.w;
    o[i] = x;
  }
}

template <typename F>
double time_us(F f, int iters) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) f();
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3 / iters;
}

int main() {
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  printf("device: %s  CUs=%d  clock=%d MHz  memclock=%d MHz  L2=%d MiB\n", p.name, p.multiProcessorCount,
         p.clockRate / 1000, p.memoryClockRate / 1000, p.l2CacheSize >> 20);
  const size_t MAXB = (size_t)2 << 30;
  float4 *a, *b;
  float* sink;
  CK(hipMalloc(&a, MAXB));
  CK(hipMalloc(&b, MAXB));
  CK(hipMalloc(&sink, 4));
  CK(hipMemset(a, 0, MAXB));
  CK(hipMemset(b, 0, MAXB));
  const int grids[] = {2048, 8192};
  const size_t sizes_mb[] = {16, 32, 64, 128, 192, 512, 2048};
  printf("%-8s %-8s %10s %10s %10s   (GB/s; footprint re-used every iteration)\n", "MiB", "grid", "read", "write", "copy");
  for (size_t mb : sizes_mb) {
    const size_t n = (mb << 20) / 16;
    for (int g : grids) {
      const int iters = mb <= 128 ? 50 : 10;
      double r = time_us([&] { hipLaunchKernelGGL(k_read<4>, dim3(g), dim3(256), 0, 0, a, n, sink); }, iters);
      double w = time_us([&] { hipLaunchKernelGGL(k_write<4>, dim3(g), dim3(256), 0, 0, a, n); }, iters);
      double c = time_us([&] { hipLaunchKernelGGL(k_copy<4>, dim3(g), dim3(256), 0, 0, a, b, n); }, iters);
      printf("%-8zu %-8d %10.0f %10.0f %10.0f\n", mb, g, (double)(mb << 20) / r / 1e3, (double)(mb << 20) / w / 1e3,
             2.0 * (double)(mb << 20) / c / 1e3);
    }
  }
  // PSF-step mix: w streams through 14 distinct 63 MB buffers (882 MB, like W_1..W_14), v/o ping-pong, r fixed
  {
    const size_t nrow = (size_t)64 * 16384 * 8 / 4;  // float4 per V-sized tensor (33.5 MB)
    const size_t vbytes = nrow * 16;
    float4* w = a;                      // 14 x 2*vbytes = 939 MB region of a
    float4* v0 = b;                     // residual
    float4* pp[2] = {b + nrow, b + 2 * nrow};
    for (int g : {2048, 4096, 8192}) {
      int m = 0;
      double t = time_us(
          [&] {
            for (int s = 0; s < 14; ++s) {
              const float4* vin = s == 0 ? v0 : pp[(s - 1) & 1];
              hipLaunchKernelGGL(k_mix, dim3(g), dim3(256), 0, 0, w + (size_t)s * 2 * nrow, vin, v0, pp[s & 1], nrow);
            }
            ++m;
          },
          10);
      const double bytes = 14.0 * 5.0 * vbytes;
      printf("psf-mix  grid=%-5d  %8.1f us/chain  %8.2f us/step  %8.0f GB/s (5 x 33.5 MB per step)\n", g, t, t / 14,
             bytes / t / 1e3);
    }
  }
  return 0;
}
