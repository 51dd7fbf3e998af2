// Device-side helpers and the structures shared by every kernel translation unit of libspamtree_hip.so (split out of
// spamtree_hip.hip in round 3: one TU per kernel family, see spamtree_amd/build.py).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>

#include "spamtree_hip.h"
#include <rccl/rccl.h>

#define NT 256
#define MAXJ ST_MAX_ANCESTORS
#define QMAX ST_MAX_Q
#define HL2PI (-0.91893853320467274178032973640562)

struct CovPar {
  int q;
  int ncb;
  double ai1[QMAX], ai2[QMAX], phi[QMAX];
  double tmv[3];
  double D[QMAX * QMAX];
  // per pair of outcomes, filled on the host (finish_covpar): everything of the Apanasovich-Genton form that does not
  // depend on the distance.  cov = amp exp(-rate h) [+ amp2 exp(-phi[vi] h) where the Dmat entry is exactly zero]
  double rate[QMAX * QMAX], amp[QMAX * QMAX], amp2[QMAX * QMAX];
};

struct Blk {
  long long row0;       // first device row
  long long panel_off;  // doubles, into a slot's panel arena (-1: none)
  long long acc_off;    // doubles, into the message arena
  int m, P, nanc, anc_ptr;
  int isref, nobs, dch_ptr, ndch;
  int acc_len, level, ld, model_id;   // acc_len: on the device = offset of the children's records FOR this block inside their records
  long long chain_off;  // panel the DESCENDANTS read as this block's rows of their chain factor: = panel_off, or (limited_tree)
                        // the block's marginal inverse Cholesky chol(K_uu)^{-1}, m x m (k_marginal_invchol)
};

// ---------------------------------------------------------------------------------------------------------------
// device helpers
// ---------------------------------------------------------------------------------------------------------------
// exp(x) for the covariance kernels: two-step Cody-Waite reduction to |r| <= ln2/2, degree-13 Taylor polynomial in
// Estrin form (truncation error < 5e-18; short dependency chains, no register copies), v_ldexp_f64 for the scaling
// (overflow -> inf, underflow -> denormals / 0 as in libm).  Under half the instructions of the library routine;
// relative error < 3e-16.
__device__ __forceinline__ double cov_exp(double x) {
  x = fmax(x, -1500.0);
  const double t = __builtin_rint(x * 1.44269504088896338700e+00);
  double r = fma(t, -6.93147180369123816490e-01, x);
  r = fma(t, -1.90821492927058770002e-10, r);
  const double r2 = r * r, r4 = r2 * r2, r8 = r4 * r4;
  const double a0 = 1.0 + r;
  const double a1 = fma(1.6666666666666666e-01, r, 0.5);                         // 1/3!, 1/2!
  const double a2 = fma(8.3333333333333332e-03, r, 4.1666666666666664e-02);      // 1/5!, 1/4!
  const double a3 = fma(1.9841269841269841e-04, r, 1.3888888888888889e-03);      // 1/7!, 1/6!
  const double a4 = fma(2.7557319223985893e-06, r, 2.4801587301587302e-05);      // 1/9!, 1/8!
  const double a5 = fma(2.5052108385441720e-08, r, 2.7557319223985888e-07);      // 1/11!, 1/10!
  const double a6 = fma(1.6059043836821613e-10, r, 2.0876756987868100e-09);      // 1/13!, 1/12!
  const double b0 = fma(a1, r2, a0), b1 = fma(a3, r2, a2), b2 = fma(a5, r2, a4);
  const double d0 = fma(b1, r4, b0), d1 = fma(a6, r4, b2);
  const double p = fma(d1, r8, d0);
  return __builtin_ldexp(p, (int)t);   // |t| < 2^31 after the clamp above, or +huge -> saturating conversion -> inf
}

// exp(x) with a 64-entry table of 2^(j/64) (in LDS: EXP2_64 copied by the kernel), for the covariance pass of k_factor_quad,
// which runs at the FP64 pipe's issue rate (44 FP64 instructions per entry, 4 cycles each, stamps of round 3): the reduced
// argument is |r| <= ln2/128, so a degree-5 polynomial is exact to 3.5e-17 and the whole exponential costs 16 FP64
// instructions + one LDS read instead of 24.  x = t ln2/64 + r, t = 64 k + j: exp(x) = 2^k 2^(j/64) e^r.  Relative error
// < 3e-16 (the table entries are correctly rounded); over- / underflow as cov_exp.
__device__ const double EXP2_64[64] = {
  0x1.0000000000000p+0, 0x1.02c9a3e778061p+0, 0x1.059b0d3158574p+0, 0x1.0874518759bc8p+0,
  0x1.0b5586cf9890fp+0, 0x1.0e3ec32d3d1a2p+0, 0x1.11301d0125b51p+0, 0x1.1429aaea92de0p+0,
  0x1.172b83c7d517bp+0, 0x1.1a35beb6fcb75p+0, 0x1.1d4873168b9aap+0, 0x1.2063b88628cd6p+0,
  0x1.2387a6e756238p+0, 0x1.26b4565e27cddp+0, 0x1.29e9df51fdee1p+0, 0x1.2d285a6e4030bp+0,
  0x1.306fe0a31b715p+0, 0x1.33c08b26416ffp+0, 0x1.371a7373aa9cbp+0, 0x1.3a7db34e59ff7p+0,
  0x1.3dea64c123422p+0, 0x1.4160a21f72e2ap+0, 0x1.44e086061892dp+0, 0x1.486a2b5c13cd0p+0,
  0x1.4bfdad5362a27p+0, 0x1.4f9b2769d2ca7p+0, 0x1.5342b569d4f82p+0, 0x1.56f4736b527dap+0,
  0x1.5ab07dd485429p+0, 0x1.5e76f15ad2148p+0, 0x1.6247eb03a5585p+0, 0x1.6623882552225p+0,
  0x1.6a09e667f3bcdp+0, 0x1.6dfb23c651a2fp+0, 0x1.71f75e8ec5f74p+0, 0x1.75feb564267c9p+0,
  0x1.7a11473eb0187p+0, 0x1.7e2f336cf4e62p+0, 0x1.82589994cce13p+0, 0x1.868d99b4492edp+0,
  0x1.8ace5422aa0dbp+0, 0x1.8f1ae99157736p+0, 0x1.93737b0cdc5e5p+0, 0x1.97d829fde4e50p+0,
  0x1.9c49182a3f090p+0, 0x1.a0c667b5de565p+0, 0x1.a5503b23e255dp+0, 0x1.a9e6b5579fdbfp+0,
  0x1.ae89f995ad3adp+0, 0x1.b33a2b84f15fbp+0, 0x1.b7f76f2fb5e47p+0, 0x1.bcc1e904bc1d2p+0,
  0x1.c199bdd85529cp+0, 0x1.c67f12e57d14bp+0, 0x1.cb720dcef9069p+0, 0x1.d072d4a07897cp+0,
  0x1.d5818dcfba487p+0, 0x1.da9e603db3285p+0, 0x1.dfc97337b9b5fp+0, 0x1.e502ee78b3ff6p+0,
  0x1.ea4afa2a490dap+0, 0x1.efa1bee615a27p+0, 0x1.f50765b6e4540p+0, 0x1.fa7c1819e90d8p+0,
};
__device__ __forceinline__ double cov_exp_tab(double x, const double *tab) {
  x = fmax(x, -1500.0);
  const double t = __builtin_rint(x * 9.233248261689366e+01);          // 64 / ln 2
  double r = fma(t, -1.08304246932675596327e-02, x);                   // ln2 / 64: the 32-bit head of cov_exp's split, scaled
  r = fma(t, -2.98158582698529328128e-12, r);
  const int ti = (int)t;
  const double T = tab[ti & 63];
  const double r2 = r * r;
  const double a0 = 1.0 + r;
  const double a1 = fma(1.6666666666666666e-01, r, 0.5);
  const double a2 = fma(8.3333333333333332e-03, r, 4.1666666666666664e-02);
  const double p = fma(fma(a2, r2, a1), r2, a0);
  return __builtin_ldexp(p * T, ti >> 6);
}

// sqrt(a) for squared distances (a >= 0): v_rsq_f64 seed (relative error 5e-8 on gfx950, measured), one coupled Goldschmidt
// step (4e-15) and ONE residual correction, after which the result is the correctly rounded square root for every one of 2^20
// random arguments in [1e-8, 2] (round 3 probe; the library's scheme adds a second correction, which changed nothing).  a is
// clamped to 1e-300 from below, so coincident points give 1e-150 instead of 0 (exp(-phi * 1e-150) == 1 exactly); squared
// distances above ~1e300 are outside the contract.
__device__ __forceinline__ double cov_sqrt(double a) {
  a = fmax(a, 1e-300);
  const double y = __builtin_amdgcn_rsq(a);
  double g = a * y, h = 0.5 * y;
  const double r = fma(-h, g, 0.5);
  g = fma(g, r, g); h = fma(h, r, h);
  const double d = fma(-g, g, a);
  return fma(d, h, g);
}

__device__ __forceinline__ double cov_entry(const CovPar &c, double xi, double yi, int vi, double xj, double yj, int vj) {
  const double dx = xi - xj, dy = yi - yj;
  const double h = cov_sqrt(dx * dx + dy * dy);
  if (c.q == 1) return c.ai1[0] * cov_exp(-c.tmv[0] * h);  // cexpcov: sigmasq = ai1(0), phi = thetamv(0)
  // mvCovAG20107 (covariance_functions.cpp:213-286): C_base(h, 0, v) = exp(-c h / psi) / psi^2 with psi = (a v + 1)^(b/2)
  // (q > 2) or exp(-c h / sqrt(v + 1)) / (v + 1) (q = 2); psi, the amplitudes and the v == 0 case are per outcome pair
  const int ij = vi * c.q + vj;
  double r = c.amp[ij] * cov_exp(-c.rate[ij] * h);
  const double a2 = c.amp2[ij];
  if (a2 != 0.0) r += a2 * cov_exp(-c.phi[vi] * h);
  return r;
}

__device__ __forceinline__ double cov_entry_tab(const CovPar &c, const double *tab, double xi, double yi, int vi, double xj, double yj, int vj) {
  const double dx = xi - xj, dy = yi - yj;
  const double h = cov_sqrt(dx * dx + dy * dy);
  if (c.q == 1) return c.ai1[0] * cov_exp_tab(-c.tmv[0] * h, tab);
  const int ij = vi * c.q + vj;
  double r = c.amp[ij] * cov_exp_tab(-c.rate[ij] * h, tab);
  const double a2 = c.amp2[ij];
  if (a2 != 0.0) r += a2 * cov_exp_tab(-c.phi[vi] * h, tab);
  return r;
}

// host: the distance-independent parts of the multivariate form
static void finish_covpar(CovPar *c) {
  const int q = c->q;
  for (int vi = 0; vi < q; ++vi)
    for (int vj = 0; vj < q; ++vj) {
      const int ij = vi * q + vj;
      const double v = c->D[ij];
      double rate, den;
      if (q > 2) {
        const double ps = std::exp(0.5 * c->tmv[1] * std::log1p(c->tmv[0] * v));
        rate = c->tmv[2] / ps; den = ps * ps;
      } else {
        const double ps = std::sqrt(v + 1.0);
        rate = c->tmv[0] / ps; den = v + 1.0;
      }
      c->rate[ij] = rate;
      if (v == 0.0) { c->amp[ij] = c->ai1[vi] * c->ai1[vi] / den; c->amp2[ij] = c->ai2[vi] * c->ai2[vi]; }
      else { c->amp[ij] = c->ai1[vi] * c->ai1[vj] / den; c->amp2[ij] = 0.0; }
    }
}

// the ancestor whose rows hold chain row k (s_ao ascending, s_ao[0] = 0): independent compares -- a search loop is a chain of
// dependent LDS reads on every block's latency path
__device__ __forceinline__ int anc_of(const int *s_ao, int J, int k) {
  int t = 0;
#pragma unroll
  for (int j = 1; j < 8; ++j) t += (j < J && k >= s_ao[j]) ? 1 : 0;
  for (int j = 8; j < J; ++j) t += (k >= s_ao[j]) ? 1 : 0;
  return t;
}

// workgroup barrier that orders LDS traffic only: unlike __syncthreads() it does not wait for global loads in flight
// (a prefetched sub-panel keeps travelling across it) nor for global stores
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  return v;
}

// block-wide sum of one double per thread; result valid in every thread. red: >= NT/64 doubles of LDS.
__device__ __forceinline__ double block_sum(double v, double *red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) red[wv] = v;
  __syncthreads();
  double s = 0.0;
  for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += red[i];
  return s;
}

// In-place lower Cholesky of the m x m row-major matrix A (lower triangle referenced).  *fail set when a pivot
// is not > 0 (LAPACK dpotrf's test, NaN included).  All threads of the block must call.
__device__ void chol_lower_inplace(double *A, int m, int *fail) {
  for (int k = 0; k < m; ++k) {
    __syncthreads();
    const double d = A[k * m + k];
    if (!(d > 0.0)) {
      if (threadIdx.x == 0) *fail = 1;
    }
    const double piv = sqrt(d);
    __syncthreads();
    for (int i = k + threadIdx.x; i < m; i += blockDim.x) A[i * m + k] = (i == k) ? piv : A[i * m + k] / piv;
    __syncthreads();
    const int r = m - k - 1;
    for (int idx = threadIdx.x; idx < r * r; idx += blockDim.x) {
      const int i = k + 1 + idx / r, j = k + 1 + idx % r;
      if (j <= i) A[i * m + j] -= A[i * m + k] * A[j * m + k];
    }
  }
  __syncthreads();
}

// Ri = L^{-1} (lower, zeros above the diagonal), one thread per column.
__device__ void tri_inverse_lower(const double *L, double *Ri, int m) {
  for (int j = threadIdx.x; j < m; j += blockDim.x) {
    for (int i = 0; i < j; ++i) Ri[i * m + j] = 0.0;
    for (int i = j; i < m; ++i) {
      double s = (i == j) ? 1.0 : 0.0;
      for (int k = j; k < i; ++k) s -= L[i * m + k] * Ri[k * m + j];
      Ri[i * m + j] = s / L[i * m + i];
    }
  }
  __syncthreads();
}

// Ri = chol(A)^{-1} for an m x m matrix in LDS (row-major, stride m, lower triangle referenced; A is destroyed), by the
// whole workgroup with ONE barrier per pivot: the symmetric elimination of [A | I] (A = L~ D L~', row i of I becomes row i of
// L~^{-1}), then Ri = D^{-1/2} L~^{-1} -- the scheme of wave_chol_eliminate for blocks too wide for one wave's registers
// (75-row blocks of the default multivariate tree).  chol_lower_inplace + tri_inverse_lower cost three barriers per pivot
// and then one THREAD per column of the inverse: 23-29 % of a reference level of config #4 (profiles/r02).
// *fail set when a pivot is not > 0.  All threads of the block must call; Ri must not alias A.
__device__ void block_chol_invert(double *A, double *Ri, int m, int *fail) {
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int idx = tid; idx < m * m; idx += nt) { const int i = idx / m, j = idx - i * m; Ri[idx] = (i == j) ? 1.0 : 0.0; }
  __syncthreads();
  const int tj = tid & 31, ti = tid >> 5, nti = nt >> 5;
  for (int k = 0; k < m; ++k) {
    const double d = A[k * m + k];
    if (!(d > 0.0) && tid == 0) *fail = 1;
    const double rd = 1.0 / d;
    for (int i = k + 1 + ti; i < m; i += nti) {
      const double f = -A[i * m + k] * rd;
      for (int j = tj; j <= i; j += 32) {
        if (j <= k) Ri[i * m + j] = fma(f, Ri[k * m + j], Ri[i * m + j]);      // row k of the identity part is final
        else A[i * m + j] = fma(f, A[j * m + k], A[i * m + j]);                // A[k][j] = A[j][k]: column k is not written in this step
      }
    }
    __syncthreads();
  }
  for (int idx = tid; idx < m * m; idx += nt) {
    const int i = idx / m, j = idx - i * m;
    if (j <= i) Ri[idx] *= rsqrt(A[i * m + i]);
  }
  __syncthreads();
}

// Philox4x32-10 (Salmon et al. 2011) -- same stream contract as oracle.StRng
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                              unsigned out[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = 0xD2511F53ull * c0, p1 = 0xCD9E8D57ull * c2;
    const unsigned hi0 = (unsigned)(p0 >> 32), lo0 = (unsigned)p0, hi1 = (unsigned)(p1 >> 32), lo1 = (unsigned)p1;
    const unsigned n0 = hi1 ^ c1 ^ k0, n2 = hi0 ^ c3 ^ k1;
    c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ double u01(unsigned a, unsigned b) {
  return ((double)(((unsigned long long)(a >> 5) << 26) + (unsigned long long)(b >> 6)) + 0.5) * (1.0 / 9007199254740992.0);
}
__device__ __forceinline__ double philox_normal(unsigned long long idx, unsigned iter, unsigned stream, unsigned long long seed) {
  unsigned o[4];
  philox4x32_10((unsigned)idx, (unsigned)(idx >> 32), iter, stream, (unsigned)seed, (unsigned)(seed >> 32), o);
  const double u1 = u01(o[0], o[1]), u2 = u01(o[2], o[3]);
  return sqrt(-2.0 * log(u1)) * cos(6.283185307179586476925286766559 * u2);
}


// ---------------------------------------------------------------------------------------------------------------
// Phase A, fast path: one workgroup (4 waves) per COLUMN GROUP = one reference block, or several sibling
// non-reference blocks (same ancestor chain), M <= 32 columns, chain P <= 256.  All dense contractions run on the
// FP64 matrix cores (v_mfma_f64_16x16x4_f64: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], C[(l>>4)+4r][l&15]):
//   per ancestor panel (last to first), in sub-panels of <= 16 rows staged in LDS:
//     V_sub = Linv_sub * K[0:Kb, :]          16 x 32 tile pair, K split over the two wave pairs
//     T^T[0:Kb, :] += Linv_sub^T * V_sub      accumulators stay in registers (<= 8 tiles of 16x16 per wave);
//                                             the V tile in C layout IS the B operand of this product
//   epilogue: R = K_uu - V'V (MFMA), Cholesky + inverse in LDS, panel_u = [-Ri*T | Ri] (MFMA), log-density terms.
// K/V live in LDS as KV[k][ldKV]; T^T is dumped into the same buffer for the epilogue.
// ---------------------------------------------------------------------------------------------------------------

// 64-bit v_readlane (the lane index must be wave-uniform)
__device__ __forceinline__ double readlane_f64(double v, int l) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_readlane(lo, l);
  hi = __builtin_amdgcn_readlane(hi, l);
  return __hiloint2double(hi, lo);
}

// sum over the 64 lanes, returned wave-uniform, without touching the LDS crossbar: rotate-and-add inside each row of 16
// lanes (DPP row_ror 8, 4, 2, 1), then the four row sums (lanes 0, 16, 32, 48) through v_readlane, added in that order
__device__ __forceinline__ double dpp_ror_add(double x, const int ctrl_sel) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  int lo2, hi2;
  if (ctrl_sel == 8) { lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x128, 0xf, 0xf, false); hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x128, 0xf, 0xf, false); }
  else if (ctrl_sel == 4) { lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x124, 0xf, 0xf, false); hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x124, 0xf, 0xf, false); }
  else if (ctrl_sel == 2) { lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x122, 0xf, 0xf, false); hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x122, 0xf, 0xf, false); }
  else { lo2 = __builtin_amdgcn_update_dpp(0, lo, 0x121, 0xf, 0xf, false); hi2 = __builtin_amdgcn_update_dpp(0, hi, 0x121, 0xf, 0xf, false); }
  return x + __hiloint2double(hi2, lo2);
}
__device__ __forceinline__ double wave_allsum(double x) {
  x = dpp_ror_add(x, 8); x = dpp_ror_add(x, 4); x = dpp_ror_add(x, 2); x = dpp_ror_add(x, 1);
  const double r0 = readlane_f64(x, 0), r1 = readlane_f64(x, 16), r2 = readlane_f64(x, 32), r3 = readlane_f64(x, 48);
  return ((r0 + r1) + r2) + r3;
}

typedef __attribute__((address_space(3))) void q_lds_void;          // LDS-DMA destinations / sources (global_load_lds)
typedef __attribute__((address_space(1))) const void q_glb_void;
#define CH_LD 33

typedef double d4 __attribute__((ext_vector_type(4)));

// Team elimination [A | I] -> [. | L^{-1}] of an m x m SPD matrix (m <= 32): the workgroup is split into teams of TEAM
// threads (128, or the whole workgroup), one matrix each; all teams run the same pivot loop (mmax = largest m, uniform) and share its barrier.
// Every thread keeps EPT elements of the lower triangles of A and of B = I in registers (m (m + 1) <= TEAM * EPT).
// Per pivot k the team publishes, UNSCALED, column k of A strictly below the diagonal, the pivot d_k itself, and row k
// of B; everything outside those ranges reads as zero, so the update is the same two instructions for every element
// at every pivot -- val -= (x1 x2) / d_k with x1 = A[i][k], x2 = A[j][k] or B[k][j] -- with no range tests; rows are
// scaled by 1 / sqrt(d_i) once at the end.
//   Am: LDS, row stride CH_LD, lower triangle valid.   Bm: receives L^{-1} (lower triangle).
//   pub: 224 doubles per team: 2 x 96 published cells ([0,36) column, [36,72) row, [80] pivot, [95] always zero), rsd[32]
typedef __attribute__((address_space(3))) double q_lds_double;
template <int EPT, int TEAM = 128>
__device__ __forceinline__ void team_chol_eliminate(double *Am, double *Bm, int m, int mmax, double *pub, int *fail, int ttid) {
  const int nA = m * (m + 1) / 2, nE = 2 * nA;
  double *rsd = pub + 192;
  for (int i = ttid; i < 192; i += TEAM) pub[i] = 0.0;
  lds_barrier();   // Am was written by other threads; pub is zero
  // per element: 32-bit LDS addresses of its two factors and of its publication cell in buffer 0 (buffer 1 = +96
  // doubles, an immediate offset in the unrolled pivot pair below), the pivot at which it is published, its output slot
  q_lds_double *p1[EPT], *p2[EPT], *pp[EPT];
  int khi[EPT], eoff[EPT];
  double val[EPT];
  q_lds_double *pub3 = (q_lds_double *)pub;
#pragma unroll
  for (int r = 0; r < EPT; ++r) {
    const int e = ttid + TEAM * r;
    p1[r] = pub3 + 95; p2[r] = pub3 + 95; pp[r] = pub3 + 94; khi[r] = -1; eoff[r] = -1; val[r] = 0.0;
    if (e < nE) {
      const int t = e < nA ? 0 : 1;
      const int f = e - t * nA;
      int i = (int)((sqrtf(8.0f * (float)f + 1.0f) - 1.0f) * 0.5f);
      while (i * (i + 1) / 2 > f) --i;
      while ((i + 1) * (i + 2) / 2 <= f) ++i;
      const int j = f - i * (i + 1) / 2;
      p1[r] = pub3 + i;
      p2[r] = pub3 + (t == 0 ? j : 36 + j);
      pp[r] = pub3 + (t == 0 ? (i == j ? 80 : i) : 36 + j);
      khi[r] = t == 0 ? j : i;
      eoff[r] = t == 1 ? i * CH_LD + j : -1;
      val[r] = t == 0 ? Am[i * CH_LD + j] : (i == j ? 1.0 : 0.0);
    }
  }
#define TCH_PIVOT(k_, PAR)                                                                                     \
  {                                                                                                            \
    _Pragma("unroll") for (int r = 0; r < EPT; ++r) if ((k_) == khi[r]) pp[r][(PAR) * 96] = val[r];            \
    if (ttid == 0) { pub3[(PAR) * 96 + (k_)] = 0.0; if ((k_) > 0) pub3[(PAR) * 96 + (k_) - 1] = 0.0; }          \
    lds_barrier();                                                                                             \
    if ((k_) < m) {                                                                                            \
      const double d = pub3[(PAR) * 96 + 80];                                                                  \
      if (ttid == 0) { if (!(d > 0.0)) *fail = 1; rsd[(k_)] = rsqrt(d); }                                      \
      double rd = __builtin_amdgcn_rcp(d);                                                                     \
      rd = fma(fma(-d, rd, 1.0), rd, rd);                                                                      \
      rd = fma(fma(-d, rd, 1.0), rd, rd);                                                                      \
      _Pragma("unroll") for (int r = 0; r < EPT; ++r) {                                                        \
        const double x1 = p1[r][(PAR) * 96], x2 = p2[r][(PAR) * 96];                                           \
        val[r] = fma(-(x1 * x2), rd, val[r]);                                                                  \
      }                                                                                                        \
    }                                                                                                          \
  }
  for (int k = 0; k < mmax; k += 2) {
    TCH_PIVOT(k, 0)
    if (k + 1 < mmax) TCH_PIVOT(k + 1, 1)
  }
#undef TCH_PIVOT
  lds_barrier();   // rsd complete
#pragma unroll
  for (int r = 0; r < EPT; ++r)
    if (eoff[r] >= 0) Bm[eoff[r]] = val[r] * rsd[khi[r]];
  lds_barrier();
}


// The same elimination by ONE wave, without LDS traffic or barriers: lane i keeps row i of A (lower triangle) and of
// B = I in registers; per pivot the pivot, column k of A (lane j's a[k]) and row k of B (lane k's b[j]) travel through
// v_readlane (wave-uniform SGPR operands of the updates).  Fully unrolled (static register indices): MM pivots of MM
// broadcasts + MM fused multiply-adds, about 400 cycles per pivot on an otherwise idle SIMD -- the team version's pivot
// costs a workgroup barrier round trip (about 1.4k cycles with eight waves).  m <= MM <= 32; rows >= m behave as identity.
//   Am: LDS, row stride CH_LD, lower triangle valid.   Bm: receives L^{-1} (lower triangle).  All 64 lanes must call.
// The block-Gibbs draw w = L^{-T} (L^{-1} b + z), S = L L' (spamtree_model.cpp:1054, 1086: Sigi_chol = L^{-1},
// w = Sigi_chol' (Sigi_chol Smu + z)), by ONE wave without ever forming L^{-1}: lane i keeps row i of S in registers;
// the right-hand side rides along the elimination as one more column (forward substitution for free); the backward
// substitution costs one wave sum per row.  No LDS traffic, no barriers inside.  m <= MM <= 32.  All 64 lanes must call.
//   Sm: LDS, row stride CH_LD, lower triangle valid.  bm, zm, wout: LDS vectors (wout may alias bm or zm).
// core: lane i holds row i of S in a[] (entries j <= i; the caller sets a[j] = (j == lane) for rows >= m and 0 above the
// diagonal), c = b_i, zi = z_i; returns w_i (lanes >= m: unspecified)
template <int MM>
__device__ __forceinline__ double wave_chol_solve_core(double (&a)[MM], double c, const double zi, int m, int *fail, int lane) {
  double dd = 1.0;
  bool bad = false;
#pragma unroll
  for (int k = 0; k < MM; ++k) {
    if (k < m) {   // wave-uniform
      const double d = readlane_f64(a[k], k);
      bad = bad || !(d > 0.0);
      dd = lane == k ? d : dd;
      double rd = __builtin_amdgcn_rcp(d);
      rd = fma(fma(-d, rd, 1.0), rd, rd);
      rd = fma(fma(-d, rd, 1.0), rd, rd);
      const double f = lane > k ? -a[k] * rd : 0.0;
#pragma unroll
      for (int j = k + 1; j < MM; ++j) a[j] = fma(f, readlane_f64(a[k], j), a[j]);
      c = fma(f, readlane_f64(c, k), c);
    }
  }
  if (bad && lane == 0) *fail = 1;
  // a[k] of lane i > k is now L_ik L_kk, the diagonal d_i = L_ii^2, c = L_ii y_i
  const double rs = rsqrt(dd);
  const double t = fma(c, rs, zi);      // y_i + z_i
  double w = 0.0;
#pragma unroll
  for (int k = MM - 1; k >= 0; --k) {
    if (k < m) {   // wave-uniform
      const double sk = wave_allsum(lane > k ? a[k] * w : 0.0);          // sum_{i > k} L_ik L_kk w_i
      const double rk = readlane_f64(rs, k);
      const double wk = (readlane_f64(t, k) - rk * sk) * rk;
      w = lane == k ? wk : w;
    }
  }
  return w;
}
template <int MM>
__device__ __forceinline__ void wave_chol_solve(const double *Sm, const double *bm, const double *zm, double *wout, int m, int *fail, int lane) {
  double a[MM];
  const bool row = lane < m;
#pragma unroll
  for (int j = 0; j < MM; ++j) a[j] = (row && j <= lane) ? Sm[min(lane, 31) * CH_LD + j] : (j == lane ? 1.0 : 0.0);
  const double c = row ? bm[min(lane, 31)] : 0.0;   // running right-hand side: ends as L_ii y_i
  const double zi = row ? zm[min(lane, 31)] : 0.0;
  const double w = wave_chol_solve_core<MM>(a, c, zi, m, fail, lane);
  if (row) wout[lane] = w;
}

template <int MM, int J0 = 0, int J1 = MM>
__device__ __forceinline__ void wave_chol_eliminate(const double *Am, double *Bm, int m, int *fail, int lane) {
  // J0, J1: this wave produces columns [J0, J1) of L^{-1}.  Two waves can share one matrix: both run the (cheaper half of
  // the) elimination of A redundantly -- no communication -- and each carries half of B's columns.
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
    if (k < m) {   // wave-uniform
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
      for (int j = J0; j < J1; ++j)
        if (j <= k) b[j] = fma(f, readlane_f64(b[j], k), b[j]);
    }
  }
  if (bad && lane == 0) *fail = 1;
  const double rs = rsqrt(dd);
  if (row) {
#pragma unroll
    for (int j = J0; j < J1; ++j)
      if (j <= lane) Bm[lane * CH_LD + j] = b[j] * rs;
  }
}


template <int J>
__device__ __forceinline__ void fmac_bcast(double &d, const double src, const double f) {
  // d += (lane J of this lane's 16-lane row: src) * f
  asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(src), "v"(f), "n"(J));
}

// Elimination of one diagonal tile (mr <= MM <= 16 rows) held in registers: a[j] = A[i][j] (j <= i, else 0; rows >= mr:
// unit vectors), on return b[jj] = L^{-1}[i][4 jj + g] for lane 16 g + i.  Returns true in every lane when a pivot was not > 0.
template <int MM>
__device__ __forceinline__ bool dpp_tile_eliminate(double (&a)[MM], double (&b)[(MM + 3) / 4], int mr, int lane) {
  constexpr int NB = (MM + 3) / 4;
  const int i = lane & 15, g = lane >> 4;
#pragma unroll
  for (int jj = 0; jj < NB; ++jj) b[jj] = (4 * jj + g == i) ? 1.0 : 0.0;
  double dd = 1.0;
  bool bad = false;
#pragma unroll
  for (int k = 0; k < MM; ++k) {
    if (k < mr) {   // wave-uniform
      const double d = readlane_f64(a[k], k);
      bad = bad || !(d > 0.0);
      dd = i == k ? d : dd;
      double rd = __builtin_amdgcn_rcp(d);
      rd = fma(fma(-d, rd, 1.0), rd, rd);
      rd = fma(fma(-d, rd, 1.0), rd, rd);
      const double f = i > k ? -a[k] * rd : 0.0;
      asm volatile("s_nop 1" ::: "memory");   // VALU write -> DPP read of the same VGPR needs two wait states; inline asm is not covered by the hazard recogniser
      // A[i][j] -= A[i][k] A[j][k] / d   (lane j's a[k], row-broadcast)
#pragma unroll
      for (int j = 0; j < MM; ++j)
        if (j > k) {
          switch (j) {   // j is a compile-time constant after unrolling: one case survives
#define CA_CASE(J_) case J_: fmac_bcast<J_>(a[J_ < MM ? J_ : 0], a[k], f); break;
            CA_CASE(1) CA_CASE(2) CA_CASE(3) CA_CASE(4) CA_CASE(5) CA_CASE(6) CA_CASE(7) CA_CASE(8)
            CA_CASE(9) CA_CASE(10) CA_CASE(11) CA_CASE(12) CA_CASE(13) CA_CASE(14) CA_CASE(15)
#undef CA_CASE
          }
        }
      // B[i][c] -= (A[i][k] / d) B[k][c]   (lane k's b[jj]); row k of B is zero beyond column k
#pragma unroll
      for (int jj = 0; jj < NB; ++jj)
        if (4 * jj <= k) {
          switch (k) {   // k is a compile-time constant after unrolling: one case survives
#define CB_CASE(K_) case K_: fmac_bcast<K_>(b[jj], b[jj], f); break;
            CB_CASE(0) CB_CASE(1) CB_CASE(2) CB_CASE(3) CB_CASE(4) CB_CASE(5) CB_CASE(6) CB_CASE(7)
            CB_CASE(8) CB_CASE(9) CB_CASE(10) CB_CASE(11) CB_CASE(12) CB_CASE(13) CB_CASE(14) CB_CASE(15)
#undef CB_CASE
          }
        }
    }
  }
  const double rs = rsqrt(dd);
#pragma unroll
  for (int jj = 0; jj < NB; ++jj) b[jj] *= rs;
  return bad;
}


// Inverse Cholesky factor of a 16 x 16 (or smaller) diagonal tile that sits inside a larger matrix, by ONE wave (strides as parameters).
//   Am: tile's first element, row stride lda, lower triangle valid.  Bm: receives L^{-1} (lower triangle), row stride ldb.
//   mr <= 16 rows; rows >= mr behave as identity and are not written.  All 64 lanes must call.
// Round 3: the DPP elimination of chol_blocked.hpp (row i in lane 16 g + i of all four lane rows, broadcasts as the DPP control of
// a v_fmac_f64, each lane row a quarter of L^{-1}'s columns) instead of the v_readlane one: a third of the instructions, the same
// arithmetic in the same order.
__device__ __forceinline__ void wave_chol_eliminate_tile(const double *Am, int lda, double *Bm, int ldb, int mr, int *fail, int lane) {
  const int i = lane & 15, g = lane >> 4;
  double a[16], b[4];
#pragma unroll
  for (int j = 0; j < 16; ++j) a[j] = (i < mr && j <= i) ? Am[(size_t)i * lda + j] : (j == i ? 1.0 : 0.0);
  const bool bad = dpp_tile_eliminate<16>(a, b, mr, lane);
  if (bad && lane == 0) *fail = 1;
  if (i < mr) {
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
      if (4 * jj + g <= i) Bm[(size_t)i * ldb + 4 * jj + g] = b[jj];
  }
}

// Ri = chol(A)^{-1} (same contract as block_chol_invert: m x m in LDS, row stride m, lower triangle of A valid, A destroyed,
// *fail set when a pivot is not > 0, all threads must call, Ri must not alias A) as a BLOCKED factorisation on 16 x 16 tiles:
// per block column, the diagonal tile's inverse Cholesky factor X_kk by one wave in registers (wave_chol_eliminate_tile), the
// panel L_ik = A_ik X_kk' and the trailing update A_ij -= L_ik L_jk' on the FP64 matrix cores (tiles dealt over the waves);
// then the inverse by block sub-diagonals, Ri_ij = -X_ii sum_k L_ik Ri_kj.  3 barriers per block column + 1 per sub-diagonal:
// 19 for a 75 x 75 matrix, against one per PIVOT (75) of block_chol_invert, whose 8-wave barrier round trips were 23 % of a
// 75-column reference level of config #4.  Entries above the diagonal of Ri are zero.
// the factorisation half of block_chol_invert_mfma: on return A (row stride lda) holds L_ik in its tiles below the block
// diagonal and X (row stride ldx; may be A itself: X_kk then replaces the lower triangle of A's diagonal tile) holds
// X_kk = L_kk^{-1} in the lower triangles of its diagonal tiles.  All threads of the block must call.
__device__ __forceinline__ void block_chol_factor_mfma(double *A, int lda, double *X, int ldx, int m, int *fail) {
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6), nw = blockDim.x >> 6;
  const int nt = (m + 15) >> 4;
  auto la = [&](int r, int c) -> double { return (r < m && c < m) ? A[(size_t)r * lda + c] : 0.0; };
  for (int kb = 0; kb < nt; ++kb) {
    const int k0 = 16 * kb;
    if (wid == 0) wave_chol_eliminate_tile(A + (size_t)k0 * lda + k0, lda, X + (size_t)k0 * ldx + k0, ldx, min(16, m - k0), fail, lane);
    __syncthreads();
    // panel: L_ik = A_ik X_kk'  (B operand: X_kk'[k][n] = X_kk[n][k], lower triangular)
    for (int ib = kb + 1 + wid; ib < nt; ib += nw) {
      d4 c = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) {
        const int xr = k0 + l15, xc = k0 + 4 * s2 + l4;
        const double xb = (xr < m && xc <= xr) ? X[(size_t)xr * ldx + xc] : 0.0;
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(la(16 * ib + l15, k0 + 4 * s2 + l4), xb, c, 0, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = 16 * ib + 4 * q + l4, cc = k0 + l15;
        if (r < m && cc < m) A[(size_t)r * lda + cc] = c[q];
      }
    }
    __syncthreads();
    // trailing update: A_ij -= L_ik L_jk' for kb < jb <= ib
    {
      const int nr = nt - kb - 1, npair = nr * (nr + 1) / 2;
      for (int e = wid; e < npair; e += nw) {
        int ii = 0;
        while ((ii + 1) * (ii + 2) / 2 <= e) ++ii;
        const int jj = e - ii * (ii + 1) / 2;
        const int ib = kb + 1 + ii, jb = kb + 1 + jj;
        d4 c;
#pragma unroll
        for (int q = 0; q < 4; ++q) c[q] = la(16 * ib + 4 * q + l4, 16 * jb + l15);
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2)
          c = __builtin_amdgcn_mfma_f64_16x16x4f64(-la(16 * ib + l15, k0 + 4 * s2 + l4), la(16 * jb + l15, k0 + 4 * s2 + l4), c, 0, 0, 0);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int r = 16 * ib + 4 * q + l4, cc = 16 * jb + l15;
          if (r < m && cc < m) A[(size_t)r * lda + cc] = c[q];
        }
      }
    }
    __syncthreads();
  }
}

__device__ void block_chol_invert_mfma(double *A, double *Ri, int m, int *fail) {
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6), nw = blockDim.x >> 6;
  const int nt = (m + 15) >> 4;
  for (int idx = tid; idx < m * m; idx += blockDim.x) Ri[idx] = 0.0;
  __syncthreads();
  auto ld = [&](const double *M_, int r, int c) -> double { return (r < m && c < m) ? M_[(size_t)r * m + c] : 0.0; };
  block_chol_factor_mfma(A, m, Ri, m, m, fail);
  // A now holds L_ik below the block diagonal, Ri's diagonal tiles X_kk.  Ri_ij = -X_ii sum_{k = j}^{i-1} L_ik Ri_kj
  for (int d = 1; d < nt; ++d) {
    for (int jb = wid; jb + d < nt; jb += nw) {
      const int ib = jb + d;
      d4 w = (d4){0.0, 0.0, 0.0, 0.0};
      for (int kb = jb; kb < ib; ++kb) {
#pragma unroll
        for (int s2 = 0; s2 < 4; ++s2)
          w = __builtin_amdgcn_mfma_f64_16x16x4f64(ld(A, 16 * ib + l15, 16 * kb + 4 * s2 + l4), ld(Ri, 16 * kb + 4 * s2 + l4, 16 * jb + l15), w, 0, 0, 0);
      }
      d4 r4 = (d4){0.0, 0.0, 0.0, 0.0};   // the accumulator layout of W is the B-operand layout of its four K-steps
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2)
        r4 = __builtin_amdgcn_mfma_f64_16x16x4f64(-ld(Ri, 16 * ib + l15, 16 * ib + 4 * s2 + l4), w[s2], r4, 0, 0, 0);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int r = 16 * ib + 4 * q + l4, cc = 16 * jb + l15;
        if (r < m && cc < m) Ri[(size_t)r * m + cc] = r4[q];
      }
    }
    __syncthreads();
  }
}


// w = L^{-T} (L^{-1} b + z), S = L L' (m x m in LDS, row stride lds_, lower triangle valid, destroyed), for 33..80-row blocks, by
// the whole workgroup: the blocked factorisation above with X_kk stored over S's own diagonal tiles, then block forward and
// backward substitutions -- thread i owns row i; a block step is one 16 x 16 triangular product with X_kk (through `vec`, m
// doubles of LDS) and one rank-16 update of the rows below / above.  About 35 barriers of four waves and five one-wave 16 x 16
// eliminations, against the 75 dependent pivots of wave_chol_solve_lds.  bv (LDS): in b, out w.  All threads must call.
__device__ void block_chol_solve_mfma(double *S, int lds_, double *vec, double *bv, const double *zg, int m, int *fail) {
  const int tid = threadIdx.x;
  const int nt = (m + 15) >> 4;
  block_chol_factor_mfma(S, lds_, S, lds_, m, fail);
  const int i = tid, ib = tid >> 4;           // row i (tid < m)
  const bool rowok = i < m;
  double r = rowok ? bv[i] : 0.0;
  for (int kb = 0; kb < nt; ++kb) {           // forward: y_k = X_kk r_k, then r_i -= L_ik y_k for the rows below
    const int k0 = 16 * kb, k1 = min(m, k0 + 16);
    if (rowok && ib == kb) vec[i] = r;
    __syncthreads();
    double y = 0.0;
    if (rowok && ib == kb) {
      for (int c = k0; c <= i; ++c) y += S[(size_t)i * lds_ + c] * vec[c];
    }
    __syncthreads();
    if (rowok && ib == kb) { vec[i] = y; r = y; }
    __syncthreads();
    if (rowok && ib > kb) {
      for (int c = k0; c < k1; ++c) r -= S[(size_t)i * lds_ + c] * vec[c];
    }
  }
  double t = rowok ? r + zg[i] : 0.0;         // r = y = L^{-1} b
  for (int kb = nt - 1; kb >= 0; --kb) {      // backward: w_k = X_kk' t_k, then t_i -= L_ki' w_k for the rows above
    const int k0 = 16 * kb, k1 = min(m, k0 + 16);
    __syncthreads();
    if (rowok && ib == kb) vec[i] = t;
    __syncthreads();
    double w = 0.0;
    if (rowok && ib == kb) {
      for (int c = i; c < k1; ++c) w += S[(size_t)c * lds_ + i] * vec[c];
    }
    __syncthreads();
    if (rowok && ib == kb) { vec[i] = w; t = w; }
    __syncthreads();
    if (rowok && ib < kb) {
      for (int c = k0; c < k1; ++c) t -= S[(size_t)c * lds_ + i] * vec[c];
    }
  }
  if (rowok) bv[i] = t;
  __syncthreads();
}

#ifdef FM_STAMPS
// diagnostic build only (never shipped): per-section shader-clock totals of k_factor_mfma, thread 0 of every workgroup
static __device__ unsigned long long g_stamps[16];
#define STAMP_DECL unsigned long long st_acc[16] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0}; unsigned long long st_t0 = clock64(), st_t1 = 0;
#define STAMP(slot) do { st_t1 = clock64(); st_acc[slot] += st_t1 - st_t0; st_t0 = st_t1; } while (0)
static __device__ int g_stamp_level = -1;   // >= 0: only workgroups of that tree level report (k_factor_quad)
#define STAMP_FLUSH_IF(c_) do { if (threadIdx.x == 0 && (c_)) { for (int q_ = 0; q_ < 16; ++q_) atomicAdd(&g_stamps[q_], st_acc[q_]); } } while (0)
#define STAMP_FLUSH STAMP_FLUSH_IF(g_stamp_level < 0)
#define STAMP_FLUSH_LEVEL(l_) STAMP_FLUSH_IF(g_stamp_level < 0 || g_stamp_level == (l_))
#ifdef ST_STAMP_SUFFIX
#define ST_STAMP_CAT2(a, b) a##b
#define ST_STAMP_CAT(a, b) ST_STAMP_CAT2(a, b)
extern "C" int ST_STAMP_CAT(st_debug_stamp_level, ST_STAMP_SUFFIX)(int level) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_level), &level, sizeof(int)); return 0; }
extern "C" int ST_STAMP_CAT(st_debug_stamps, ST_STAMP_SUFFIX)(unsigned long long *out, int reset) {
  if (out) (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 16);
  if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof(z)); }
  return 0;
}
#endif   // ST_STAMP_SUFFIX
#else
#define STAMP_DECL
#define STAMP(slot) do {} while (0)
#define STAMP_FLUSH do {} while (0)
#define STAMP_FLUSH_LEVEL(l_) do {} while (0)
#endif


struct Grp {
  long long row0;  // first device row of the group's columns
  int blk0, nblk;  // device blocks blk0 .. blk0+nblk-1 (siblings)
  int M, P;
};

// Group descriptor: everything the per-group kernels used to chase through grps -> blks -> anc_idx -> blks -> dch_idx ->
// blks, flattened on the host into one fixed-stride record of 64-bit words (one global round trip instead of four):
//   [0] row0  [1] acc_off  [2] M | P<<32  [3] J | nblk<<32  [4] isref | level<<32  [5] nch | acc_len<<32  [6] blk0
//   [7] total record length   then per ancestor t: am | ao<<32, first row, panel offset, record offset (4 words)
//   then per block: panel offset, first row, ld (3 words)   then per direct child holding a record: its acc_off
#define GD_MAXW 272
struct GdHead {
  long long row0, acc_off;
  int M, P, nanc, nblk, isref, level, ndch, acc_len, blk0;
};
__device__ __forceinline__ GdHead gd_unpack(const long long *s_gd, int tid, int *s_am, int *s_ao, long long *s_arow, long long *s_apan,
                                            long long *s_aoff, long long *s_bpan, long long *s_brow, int *s_bld, long long *s_coff) {
  GdHead H;
  // every thread reads the same words: keep the header in scalar registers
  auto sll = [](long long v) {
    const int lo = __builtin_amdgcn_readfirstlane((int)(v & 0xffffffffLL)), hi = __builtin_amdgcn_readfirstlane((int)(v >> 32));
    return ((long long)hi << 32) | (unsigned int)lo;
  };
  auto slo = [](long long v) { return __builtin_amdgcn_readfirstlane((int)(v & 0xffffffffLL)); };
  auto shi = [](long long v) { return __builtin_amdgcn_readfirstlane((int)(v >> 32)); };
  H.row0 = sll(s_gd[0]); H.acc_off = sll(s_gd[1]);
  H.M = slo(s_gd[2]); H.P = shi(s_gd[2]);
  H.nanc = slo(s_gd[3]); H.nblk = shi(s_gd[3]);
  H.isref = slo(s_gd[4]); H.level = shi(s_gd[4]);
  H.ndch = slo(s_gd[5]); H.acc_len = shi(s_gd[5]);
  H.blk0 = slo(s_gd[6]);
  if (tid < H.nanc) {
    const long long *a = s_gd + 8 + 4 * tid;
    s_am[tid] = (int)(a[0] & 0xffffffffLL); s_ao[tid] = (int)(a[0] >> 32);
    s_arow[tid] = a[1];
    if (s_apan) s_apan[tid] = a[2];
    if (s_aoff) s_aoff[tid] = a[3];
  }
  if (tid == 0) { s_ao[H.nanc] = H.P; if (s_aoff) s_aoff[H.nanc] = s_gd[7]; }
  if (tid >= 64 && tid < 64 + H.nblk) {
    const long long *b = s_gd + 8 + 4 * H.nanc + 3 * (tid - 64);
    s_bpan[tid - 64] = b[0]; s_brow[tid - 64] = b[1]; s_bld[tid - 64] = (int)b[2];
  }
  if (s_coff && tid >= 128 && tid < 128 + H.ndch) s_coff[tid - 128] = s_gd[8 + 4 * H.nanc + 3 * H.nblk + (tid - 128)];
  return H;
}

