// Diagnostic micro-benchmark (not part of the product): the one-wave 25 x 25 inverse-Cholesky of k_factor_quad's reference
// levels, v_readlane elimination (wave_chol_eliminate<27>, copied below) against the blocked 16 + 9 DPP / MFMA scheme
// (csrc/chol_blocked.hpp).  Four waves per workgroup (one per SIMD, as in the kernel), one matrix each.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I spamtree_amd/csrc -o profiles/micro/chol25 profiles/micro/chol25.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CH_LD 33
typedef double d4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ double readlane_f64(double v, int l) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, l);
  hi = __builtin_amdgcn_readlane(hi, l);
  return __hiloint2double(hi, lo);
}
#include "chol_blocked.hpp"

template <int MM>
__device__ __forceinline__ void wave_chol_eliminate_old(const double *Am, double *Bm, int m, int *fail, int lane) {
  double a[MM], b[MM];
  const bool row = lane < m;
#pragma unroll
  for (int j = 0; j < MM; ++j) {
    a[j] = (row && j <= lane) ? Am[min(lane, 31) * CH_LD + j] : (j == lane ? 1.0 : 0.0);
    b[j] = j == lane ? 1.0 : 0.0;
  }
  double dd = 1.0;
  bool bad = false;
#pragma unroll
  for (int k = 0; k < MM; ++k) {
    if (k < m) {
      const double d = readlane_f64(a[k], k);
      bad = bad || !(d > 0.0);
      dd = lane == k ? d : dd;
      double rd = __builtin_amdgcn_rcp(d);
      rd = fma(fma(-d, rd, 1.0), rd, rd);
      rd = fma(fma(-d, rd, 1.0), rd, rd);
      const double f = lane > k ? -a[k] * rd : 0.0;
#pragma unroll
      for (int j = k + 1; j < MM; ++j) a[j] = fma(f, readlane_f64(a[k], j), a[j]);
#pragma unroll
      for (int j = 0; j < MM; ++j)
        if (j <= k) b[j] = fma(f, readlane_f64(b[j], k), b[j]);
    }
  }
  if (bad && lane == 0) *fail = 1;
  const double rs = rsqrt(dd);
  if (row) {
#pragma unroll
    for (int j = 0; j < MM; ++j)
      if (j <= lane) Bm[lane * CH_LD + j] = b[j] * rs;
  }
}

template <int MODE>
__global__ __launch_bounds__(256) void k_chol(const double *Rg, double *Og, int m, int reps, long long *cyc, int *failg) {
  __shared__ double sR[4][32 * CH_LD], sB[4][32 * CH_LD];
  __shared__ int s_fail[4];
  const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t mat = (size_t)blockIdx.x * 4 + wid;
  if (lane == 0) s_fail[wid] = 0;
  long long tot = 0;
  for (int rep = 0; rep < reps; ++rep) {
    for (int e = lane; e < 32 * CH_LD; e += 64) { sR[wid][e] = __builtin_nan(""); sB[wid][e] = 0.0; }
    for (int e = lane; e < m * m; e += 64) { const int r = e / m, c = e % m; if (c <= r) sR[wid][r * CH_LD + c] = Rg[mat * m * m + e]; }
    __syncthreads();
    const long long t0 = clock64();
    if (MODE == 0) wave_chol_eliminate_old<27>(sR[wid], sB[wid], m, &s_fail[wid], lane);
    else wave_chol_eliminate_blocked<11>(sR[wid], sB[wid], m, &s_fail[wid], lane);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    tot += clock64() - t0;
    __syncthreads();
  }
  for (int e = lane; e < m * m; e += 64) { const int r = e / m, c = e % m; Og[mat * m * m + e] = c <= r ? sB[wid][r * CH_LD + c] : 0.0; }
  if (lane == 0) { atomicAdd((unsigned long long *)cyc, (unsigned long long)tot); if (s_fail[wid]) atomicAdd(failg, 1); }
}

int main(int argc, char **argv) {
  const int nwg = 256, nmat = nwg * 4, reps = 20;
  for (int m : {25, 16, 9, 27, 17, 20}) {
    std::vector<double> R((size_t)nmat * m * m), O(R.size()), ref(R.size());
    srand(7 + m);
    for (int t = 0; t < nmat; ++t) {   // SPD: exponential covariance of random points + nugget, like the kernel's Schur blocks
      std::vector<double> x(m), y(m);
      for (int i = 0; i < m; ++i) { x[i] = rand() / (double)RAND_MAX; y[i] = rand() / (double)RAND_MAX; }
      double *A = &R[(size_t)t * m * m];
      for (int i = 0; i < m; ++i) for (int j = 0; j < m; ++j)
        A[i * m + j] = 2.3 * exp(-6.0 * sqrt((x[i] - x[j]) * (x[i] - x[j]) + (y[i] - y[j]) * (y[i] - y[j]))) + (i == j ? 1e-3 : 0.0);
      // reference: L = chol(A), X = L^{-1} in long double
      std::vector<long double> L(m * m, 0.0L), X(m * m, 0.0L);
      for (int j = 0; j < m; ++j) {
        long double s = A[j * m + j];
        for (int k = 0; k < j; ++k) s -= L[j * m + k] * L[j * m + k];
        L[j * m + j] = sqrtl(s);
        for (int i = j + 1; i < m; ++i) {
          long double v = A[i * m + j];
          for (int k = 0; k < j; ++k) v -= L[i * m + k] * L[j * m + k];
          L[i * m + j] = v / L[j * m + j];
        }
      }
      for (int c = 0; c < m; ++c)
        for (int i = c; i < m; ++i) {
          long double v = i == c ? 1.0L : 0.0L;
          for (int k = c; k < i; ++k) v -= L[i * m + k] * X[k * m + c];
          X[i * m + c] = v / L[i * m + i];
        }
      for (int e = 0; e < m * m; ++e) ref[(size_t)t * m * m + e] = (double)X[e];
    }
    double *dR, *dO; long long *dc; int *df;
    hipMalloc(&dR, R.size() * 8); hipMalloc(&dO, R.size() * 8); hipMalloc(&dc, 8); hipMalloc(&df, 4);
    hipMemcpy(dR, R.data(), R.size() * 8, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 2; ++mode) {
      hipMemset(dc, 0, 8); hipMemset(df, 0, 4); hipMemset(dO, 0, R.size() * 8);
      if (mode == 0) hipLaunchKernelGGL(k_chol<0>, dim3(nwg), dim3(256), 0, 0, dR, dO, m, reps, dc, df);
      else hipLaunchKernelGGL(k_chol<1>, dim3(nwg), dim3(256), 0, 0, dR, dO, m, reps, dc, df);
      if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
      long long c; int f;
      hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost); hipMemcpy(&f, df, 4, hipMemcpyDeviceToHost);
      hipMemcpy(O.data(), dO, R.size() * 8, hipMemcpyDeviceToHost);
      double err = 0, mx = 0;
      for (size_t e = 0; e < O.size(); ++e) { err = fmax(err, fabs(O[e] - ref[e])); mx = fmax(mx, fabs(ref[e])); if (O[e] != O[e]) err = 1e300; }
      printf("m %2d  %-28s clock64 ticks per elimination %8.0f   max |err| %.3e (max |Ri| %.3e)  fails %d\n", m,
             mode == 0 ? "readlane (27 pivots)" : "blocked 16 + 9 DPP / MFMA", (double)c / ((double)nmat * reps), err, mx, f);
    }
    hipFree(dR); hipFree(dO); hipFree(dc); hipFree(df);
  }
  return 0;
}
