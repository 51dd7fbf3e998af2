// Diagnostic micro-benchmark (not part of the product): how the FP64 matrix and vector pipes of one gfx950 SIMD behave.
//   1. v_mfma_f64_16x16x4_f64 issue interval: one dependent chain vs 2 / 4 independent chains, 1 / 2 waves per SIMD
//   2. do FP64 MFMA and FP64 VALU (v_fma_f64) of two different waves on a SIMD overlap?
// Build: hipcc --offload-arch=gfx950 -O3 -o f64_pipes f64_pipes.hip ; run: ./f64_pipes
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
#define MF(a, b, c) c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0)

template <int CH>
__global__ __launch_bounds__(512) void k_mfma(double *out, int iters, long long *cyc) {
  d4 acc[CH];
  for (int i = 0; i < CH; ++i) acc[i] = (d4){0, 0, 0, 0};
  double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
  long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8 / CH; ++r)
#pragma unroll
      for (int i = 0; i < CH; ++i) MF(a, b, acc[i]);
  }
  long long t1 = clock64();
  double s = 0;
  for (int i = 0; i < CH; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

// mode bit 0: even waves run MFMA; bit 1: odd waves run v_fma_f64 chains (8 independent); waves not selected exit at once
__global__ __launch_bounds__(512) void k_mix(double *out, int iters, int mode, long long *cyc) {
  const int wid = threadIdx.x >> 6;
  const bool odd = (wid >> 2) & 1;   // waves 0-3 land on SIMDs 0-3, waves 4-7 again on SIMDs 0-3
  double s = 0;
  long long t0 = clock64();
  if (!odd && (mode & 1)) {
    d4 a0 = {0, 0, 0, 0}, a1 = a0;
    double a = threadIdx.x * 1e-3, b = 1.0 + threadIdx.x * 1e-4;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { MF(a, b, a0); MF(a, b, a1); }
    }
    s = a0[0] + a1[1];
  } else if (odd && (mode & 2)) {
    double x[8];
    for (int i = 0; i < 8; ++i) x[i] = threadIdx.x * 1e-3 + i;
    const double m = 1.0000001, c = 1e-9;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = __builtin_fma(x[i], m, c);   // 128 v_fma_f64 per iteration = 512 issue cycles
    }
    for (int i = 0; i < 8; ++i) s += x[i];
  }
  long long t1 = clock64();
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) cyc[wid] = t1 - t0;
}

int main() {
  double *out; long long *cyc, h[8];
  hipMalloc(&out, 1 << 24); hipMalloc(&cyc, 64);
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto run = [&](auto fn, const char *name, int threads, double n_mfma_per_wave) {
    fn(threads); hipDeviceSynchronize();
    hipEventRecord(e0); fn(threads); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-44s threads/WG %3d  %8.3f ms  clock64 ticks/MFMA (wave 0) %7.2f  ns/MFMA/wave %7.2f\n", name, threads, ms,
           (double)h[0] / n_mfma_per_wave, ms * 1e6 / n_mfma_per_wave);
  };
  for (int threads : {256, 512}) {   // 256 = 1 wave per SIMD, 512 = 2 waves per SIMD; 256 workgroups = one per CU
    run([&](int t) { hipLaunchKernelGGL(k_mfma<1>, dim3(256), dim3(t), 0, 0, out, iters, cyc); }, "mfma f64 16x16x4, 1 dependent chain", threads, iters * 8.0);
    run([&](int t) { hipLaunchKernelGGL(k_mfma<2>, dim3(256), dim3(t), 0, 0, out, iters, cyc); }, "mfma f64 16x16x4, 2 chains", threads, iters * 8.0);
    run([&](int t) { hipLaunchKernelGGL(k_mfma<4>, dim3(256), dim3(t), 0, 0, out, iters, cyc); }, "mfma f64 16x16x4, 4 chains", threads, iters * 8.0);
  }
  for (int mode : {1, 2, 3}) {
    hipLaunchKernelGGL(k_mix, dim3(256), dim3(512), 0, 0, out, iters, mode, cyc); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k_mix, dim3(256), dim3(512), 0, 0, out, iters, mode, cyc); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
    printf("mix mode %d (1 = MFMA waves only, 2 = v_fma_f64 waves only, 3 = both on the same SIMDs): %8.3f ms; ticks wave0 (mfma) %lld wave4 (valu) %lld\n",
           mode, ms, h[0], h[4]);
  }
  printf("(8 MFMA and 128 v_fma_f64 per iteration, %d iterations)\n", iters);
  return 0;
}
