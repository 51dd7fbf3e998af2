// Box-measured peaks for the roofline report (bench.py `roofline.peak_measured`; BASELINE.md section 3 asks for a stream-copy
// and an FP64 FMA / MFMA peak measured on the box next to the vendor figures).  Not on the product path: three self-contained
// kernels timed with HIP events on a private stream, about 50 ms each.
#include <hip/hip_runtime.h>

#include <cstdint>

#include "spamtree_hip.h"

namespace {

typedef double pd4 __attribute__((ext_vector_type(4)));
typedef double pd2 __attribute__((ext_vector_type(2)));

// grid-stride copy, 16 bytes per lane and request (the widest global access): read + write traffic = 2 x bytes.  Two shapes
// are timed and the better one reported (round 3, one box: plain loads at 4 workgroups per CU 5.86 TB/s, four non-temporal
// requests in flight per lane at 64 workgroups per CU 5.39 TB/s, hipMemcpyDtoD 5.19 TB/s)
__global__ __launch_bounds__(256) void k_probe_copy(const pd2 *__restrict__ src, pd2 *__restrict__ dst, long long n2) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) dst[i] = src[i];
}
__global__ __launch_bounds__(256) void k_probe_copy4(const pd2 *__restrict__ src, pd2 *__restrict__ dst, long long n2) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + 3 * stride < n2; i += 4 * stride) {
    const pd2 v0 = __builtin_nontemporal_load(src + i), v1 = __builtin_nontemporal_load(src + i + stride);
    const pd2 v2 = __builtin_nontemporal_load(src + i + 2 * stride), v3 = __builtin_nontemporal_load(src + i + 3 * stride);
    __builtin_nontemporal_store(v0, dst + i); __builtin_nontemporal_store(v1, dst + i + stride);
    __builtin_nontemporal_store(v2, dst + i + 2 * stride); __builtin_nontemporal_store(v3, dst + i + 3 * stride);
  }
  for (; i < n2; i += stride) dst[i] = src[i];
}

// v_mfma_f64_16x16x4_f64 back to back, four independent accumulators per wave, 8 waves per CU-resident workgroup
__global__ __launch_bounds__(512) void k_probe_mfma(double *out, int iters) {
  pd4 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = (pd4){0.0, 0.0, 0.0, 0.0};
  const double a = 1e-3 * threadIdx.x, b = 1.0 + 1e-4 * threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// v_fma_f64, eight independent chains per lane
__global__ __launch_bounds__(512) void k_probe_fma(double *out, int iters) {
  double x[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) x[i] = 1e-3 * threadIdx.x + i;
  const double m = 1.0000001, c = 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r)
#pragma unroll
      for (int i = 0; i < 8; ++i) x[i] = __builtin_fma(x[i], m, c);
  }
  double s = 0.0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += x[i];
  out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

#define PCHK(call)                          \
  do {                                      \
    if ((call) != hipSuccess) { rc = -1; goto done; } \
  } while (0)

}  // namespace

// out[0] = stream copy GB/s (read + write bytes over the kernel time, `bytes` per buffer, best of `reps` launches)
// out[1] = FP64 MFMA TFLOP/s (v_mfma_f64_16x16x4_f64: 2048 flop per wave-instruction), out[2] = FP64 FMA TFLOP/s (v_fma_f64)
extern "C" int st_probe_peaks(int device, int64_t bytes, int reps, double *out3) {
  int rc = 0;
  double *a = nullptr, *b = nullptr;
  hipStream_t st = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipDeviceProp_t prop;
  float ms = 0.f;
  if (!out3 || bytes < (1 << 20) || reps < 1) return -2;
  out3[0] = out3[1] = out3[2] = 0.0;
  PCHK(hipSetDevice(device));
  PCHK(hipGetDeviceProperties(&prop, device));
  PCHK(hipStreamCreate(&st));
  PCHK(hipEventCreate(&e0));
  PCHK(hipEventCreate(&e1));
  PCHK(hipMalloc(&a, (size_t)bytes));
  PCHK(hipMalloc(&b, (size_t)bytes));
  PCHK(hipMemsetAsync(a, 1, (size_t)bytes, st));
  PCHK(hipMemsetAsync(b, 0, (size_t)bytes, st));
  {
    const int ncu = prop.multiProcessorCount;
    const long long n2 = bytes / 16;
    double best = 0.0;
    for (int r = 0; r < 2 * (reps + 1); ++r) {   // two shapes alternately; the first launch of each untimed
      PCHK(hipEventRecord(e0, st));
      if (r & 1) hipLaunchKernelGGL(k_probe_copy4, dim3(ncu * 64), dim3(256), 0, st, (const pd2 *)a, (pd2 *)b, n2);
      else hipLaunchKernelGGL(k_probe_copy, dim3(ncu * 4), dim3(256), 0, st, (const pd2 *)a, (pd2 *)b, n2);
      PCHK(hipEventRecord(e1, st));
      PCHK(hipEventSynchronize(e1));
      PCHK(hipEventElapsedTime(&ms, e0, e1));
      if (r > 1 && ms > 0.f) { const double g = 2.0 * (double)(n2 * 16) / (ms * 1e-3) / 1e9; if (g > best) best = g; }
    }
    out3[0] = best;
    const int iters = 4000;
    for (int which = 0; which < 2; ++which) {
      double bestf = 0.0;
      for (int r = 0; r < 3; ++r) {
        PCHK(hipEventRecord(e0, st));
        if (which == 0) hipLaunchKernelGGL(k_probe_mfma, dim3(ncu * 2), dim3(512), 0, st, a, iters);
        else hipLaunchKernelGGL(k_probe_fma, dim3(ncu * 2), dim3(512), 0, st, a, iters);
        PCHK(hipEventRecord(e1, st));
        PCHK(hipEventSynchronize(e1));
        PCHK(hipEventElapsedTime(&ms, e0, e1));
        // MFMA: 16 wave-instructions x 2048 flop per iteration and wave; FMA: 64 instructions x 64 lanes x 2 flop
        const double flop = (double)ncu * 2 * 8 * iters * (which == 0 ? 16.0 * 2048.0 : 64.0 * 128.0);
        if (r > 0 && ms > 0.f) { const double t = flop / (ms * 1e-3) / 1e12; if (t > bestf) bestf = t; }
      }
      out3[1 + which] = bestf;
    }
  }
done:
  if (a) (void)hipFree(a);
  if (b) (void)hipFree(b);
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (st) (void)hipStreamDestroy(st);
  return rc;
}
