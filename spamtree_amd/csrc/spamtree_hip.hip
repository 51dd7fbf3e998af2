// spamtree_hip.hip -- gfx950 kernels + C-ABI for spamtree's per-Gibbs-sweep DAG-node linear algebra.
//
// What each kernel replaces in the reference (paths relative to /root/reference/src):
//   k_factor<MODE_FACTOR>  spamtree_model.cpp:834-998  get_loglik_comps_w_std  (phase A)  + covariance_functions.cpp:95-111, 213-286
//   k_factor<MODE_PREDICT> spamtree_model.cpp:1234-1358 predict_std            (phase P)
//   k_sample               spamtree_model.cpp:1011-1226 gibbs_sample_w_std     (phase B)
//   k_loglik               spamtree_model.cpp:781-826  get_loglik_w_std        (phase C)
//   k_stats / k_xb         spamtree_model.cpp:1374-1375, 1382, 1397-1400       (beta / tausq sufficient statistics)
//
// Design (DESIGN.md has the derivation): the reference materialises, per block u with ancestor rows PI_u,
// H_u = K_{u,pa} K_{pa,pa}^{-1}, the dense (P+m)^2 inverse Cholesky of K_{[pa,u]} and its Gram matrix.  The inverse
// Cholesky of an ancestor set is block lower triangular and its row panel for block a is [-Ri_a H_a | Ri_a]
// (tree_utils.cpp:204-206), so this build stores exactly ONE array per block, that panel ("Linv panel",
// m x (P+m), row-major), and every phase works from the chain of ancestor panels:
//   V = Linv_pa * K_{pa,u}            (one pass over the chain, ancestors last-to-first so V overwrites K in place)
//   T = V' * Linv_pa = H_u            (accumulated in the same pass)
//   R = K_uu - V'V,  Ri = chol(R)^{-1},  panel_u = [-Ri*T | Ri]
// Messages to ancestors are pushed as per-ancestor (m_a x m_a, m_a) pairs and summed hierarchically through
// direct children in a fixed order (no FP64 atomics -> bit-reproducible for any launch geometry).
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

// sqrt(a) for squared distances (a >= 0): v_rsq_f64 seed, one coupled Goldschmidt step, two residual corrections (the
// library's scheme without its range scaling).  a is clamped to 1e-300 from below, so coincident points give 1e-150
// instead of 0 (exp(-phi * 1e-150) == 1 exactly); squared distances above ~1e300 are outside the contract.
__device__ __forceinline__ double cov_sqrt(double a) {
  a = fmax(a, 1e-300);
  const double y = __builtin_amdgcn_rsq(a);
  double g = a * y, h = 0.5 * y;
  const double r = fma(-h, g, 0.5);
  g = fma(g, r, g); h = fma(h, r, h);
  double d = fma(-g, g, a);
  g = fma(d, h, g);
  d = fma(-g, g, a);
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

__global__ void k_normals(double *z, const long long *dev2model, long long n, unsigned iter, unsigned stream, unsigned long long seed) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) z[i] = philox_normal((unsigned long long)dev2model[i], iter, stream, seed);
}

// ---------------------------------------------------------------------------------------------------------------
// Phase A / P : per block, one workgroup.  BIG=false keeps K/V, T, the row stage and the m x m factors in LDS;
// BIG=true keeps them in a per-workgroup slice of a global scratch arena (any m, P).
// ---------------------------------------------------------------------------------------------------------------
struct FactorArgs {
  const Blk *blks;
  const int *anc_idx;
  const int *list;  // device block ids to process
  int nlist;
  const double *cx, *cy;
  const int *mv;
  const double *w_in;   // current w (device order)
  double *w_out;        // predict: where samples go
  const double *z;      // predict: normals (device order)
  double *panels;       // slot arena
  double *logdet_c, *loglik_c;
  int *errflag;         // atomicMin(level*16 + code)
  double *scratch;      // BIG
  long long scratch_stride;
  int maxP, maxM, maxMa, SR;
};

#define MODE_FACTOR 0
#define MODE_PREDICT 1

// limited_tree (/root/reference/src/spamtree_model.cpp:901-903, 1275-1278: Kxx_inv(u) = inv_sympd(K_uu), every block has
// ONE parent, /root/reference/src/tree_dep.cpp:133-186): the chain factor the children of u work with is the block's MARGINAL
// inverse Cholesky chol(K_uu)^{-1} (m x m, row-major, Blk::chain_off), not its conditional panel.  One workgroup per block.
struct MarginalArgs {
  const Blk *blks;
  const int *list;
  int nlist;
  const double *cx, *cy;
  const int *mv;
  double *panels;
  int *errflag;
  int maxM;
};
__global__ __launch_bounds__(NT) void k_marginal_invchol(MarginalArgs A, CovPar cp) {
  extern __shared__ double lds[];   // K (m x m) | L^{-1} (m x m)
  __shared__ int s_fail;
  const int tid = threadIdx.x;
  for (int li = blockIdx.x; li < A.nlist; li += gridDim.x) {
    const Blk B = A.blks[A.list[li]];
    const int m = B.m;
    double *K = lds, *Li = lds + (size_t)A.maxM * A.maxM;
    if (tid == 0) s_fail = 0;
    __syncthreads();
    for (int idx = tid; idx < m * m; idx += NT) {
      const int i = idx / m, j = idx - i * m;
      const long long ri = B.row0 + i, rj = B.row0 + j;
      K[idx] = (j <= i) ? cov_entry(cp, A.cx[ri], A.cy[ri], A.mv[ri], A.cx[rj], A.cy[rj], A.mv[rj]) : 0.0;
    }
    __syncthreads();
    chol_lower_inplace(K, m, &s_fail);
    tri_inverse_lower(K, Li, m);
    double *out = A.panels + B.chain_off;
    for (int idx = tid; idx < m * m; idx += NT) {
      const int i = idx / m, j = idx - i * m;
      out[idx] = (j <= i) ? Li[idx] : 0.0;
    }
    if (tid == 0 && s_fail) atomicMin(A.errflag, B.level * 16 + 2);   // errtype 2 (:919), reported at the block's level
    __syncthreads();
  }
}

template <bool BIG, int MODE>
__global__ __launch_bounds__(NT) void k_factor(FactorArgs A, CovPar cp) {
  extern __shared__ double lds[];
  __shared__ int s_anc[MAXJ], s_am[MAXJ], s_ao[MAXJ + 1];
  __shared__ long long s_arow[MAXJ], s_apan[MAXJ];
  __shared__ int s_fail;
  __shared__ double s_red[NT / 64];

  const int tid = threadIdx.x;
  const int maxP = A.maxP, maxM = A.maxM, maxMa = A.maxMa, SR = A.SR;
  // LDS carve
  double *sx = lds;
  double *sy = sx + (maxP + maxM);
  double *wv = sy + (maxP + maxM);
  double *hv = wv + (maxP + maxM);     // maxM
  double *ev = hv + maxM;              // maxM
  double *rd = ev + maxM;              // maxM
  double *stage = rd + maxM;           // SR * maxP
  int *smv = (int *)(stage + (size_t)SR * maxP);
  double *big0 = (double *)(smv + ((maxP + maxM + 1) & ~1));
  double *KV, *Tt, *Vp, *R, *Ri;
  if (BIG) {
    double *g = A.scratch + (size_t)blockIdx.x * A.scratch_stride;
    KV = g; Tt = KV + (size_t)maxP * maxM; Vp = Tt + (size_t)maxP * maxM; R = Vp + (size_t)maxMa * maxM; Ri = R + (size_t)maxM * maxM;
  } else {
    KV = big0; Tt = KV + (size_t)maxP * maxM; Vp = Tt + (size_t)maxP * maxM; R = Vp + (size_t)maxMa * maxM; Ri = R + (size_t)maxM * maxM;
  }

  for (int li = blockIdx.x; li < A.nlist; li += gridDim.x) {
    const int b = A.list[li];
    const Blk B = A.blks[b];
    const int m = B.m, P = B.P, J = B.nanc;
    __syncthreads();
    if (tid < J) {
      const int a = A.anc_idx[B.anc_ptr + tid];
      s_anc[tid] = a;
      s_am[tid] = A.blks[a].m;
      s_arow[tid] = A.blks[a].row0;
      s_apan[tid] = A.blks[a].chain_off;
    }
    if (tid == 0) s_fail = 0;
    __syncthreads();
    if (tid == 0) {
      int o = 0;
      for (int t = 0; t < J; ++t) { s_ao[t] = o; o += s_am[t]; }
      s_ao[J] = o;
    }
    __syncthreads();
    for (int t = 0; t < J; ++t) {
      const long long r0 = s_arow[t];
      const int oa = s_ao[t];
      for (int i = tid; i < s_am[t]; i += NT) {
        sx[oa + i] = A.cx[r0 + i]; sy[oa + i] = A.cy[r0 + i]; smv[oa + i] = A.mv[r0 + i]; wv[oa + i] = A.w_in[r0 + i];
      }
    }
    for (int i = tid; i < m; i += NT) {
      sx[P + i] = A.cx[B.row0 + i]; sy[P + i] = A.cy[B.row0 + i]; smv[P + i] = A.mv[B.row0 + i]; wv[P + i] = A.w_in[B.row0 + i];
    }
    __syncthreads();
    // K_{pa,u}  (covariance_functions.cpp:95-111 / :213-286), T = 0
    for (int idx = tid; idx < P * m; idx += NT) {
      const int k = idx / m, j = idx - k * m;
      KV[idx] = cov_entry(cp, sx[k], sy[k], smv[k], sx[P + j], sy[P + j], smv[P + j]);
      Tt[idx] = 0.0;
    }
    __syncthreads();
    // one pass over the ancestor chain, last ancestor first
    for (int t = J - 1; t >= 0; --t) {
      const int ma = s_am[t], oa = s_ao[t], Kb = oa + ma;
      const double *pa = A.panels + s_apan[t];
      for (int r0 = 0; r0 < ma; r0 += SR) {
        const int sr = min(SR, ma - r0);
        const double *src = pa + (size_t)r0 * Kb;
        for (int idx = tid; idx < sr * Kb; idx += NT) stage[idx] = src[idx];
        __syncthreads();
        for (int idx = tid; idx < sr * m; idx += NT) {
          const int i = idx / m, j = idx - i * m;
          const double *srow = stage + i * Kb;
          double acc = 0.0;
          for (int k = 0; k < Kb; ++k) acc += srow[k] * KV[k * m + j];
          Vp[(r0 + i) * m + j] = acc;
        }
        __syncthreads();
        for (int idx = tid; idx < m * Kb; idx += NT) {
          const int j = idx / Kb, k = idx - j * Kb;
          double acc = Tt[j * P + k];
          for (int i = 0; i < sr; ++i) acc += Vp[(r0 + i) * m + j] * stage[i * Kb + k];
          Tt[j * P + k] = acc;
        }
        __syncthreads();
      }
      for (int idx = tid; idx < ma * m; idx += NT) KV[oa * m + idx] = Vp[idx];
      __syncthreads();
    }
    // hv = H w_pa  (wave per row)
    {
      const int lane = tid & 63, wid = tid >> 6;
      for (int j = wid; j < m; j += NT / 64) {
        double acc = 0.0;
        for (int k = lane; k < P; k += 64) acc += Tt[j * P + k] * wv[k];
        acc = wave_sum(acc);
        if (lane == 0) hv[j] = acc;
      }
    }
    __syncthreads();

    if (MODE == MODE_PREDICT) {
      // spamtree_model.cpp:1306-1326: w_i = H_i w_pa + sqrt(max(K_ii - H_i K_{pa,i}, 0)) z_i
      for (int i = tid; i < m; i += NT) {
        double acc = cov_entry(cp, sx[P + i], sy[P + i], smv[P + i], sx[P + i], sy[P + i], smv[P + i]);
        for (int k = 0; k < P; ++k) acc -= KV[k * m + i] * KV[k * m + i];
        const double sd = (acc > 0.0) ? sqrt(acc) : 0.0;
        A.w_out[B.row0 + i] = hv[i] + sd * A.z[B.row0 + i];
      }
      continue;
    }

    double *pu = A.panels + B.panel_off;
    const int ld = B.ld;
    double wcore_part = 0.0, logdet_part = 0.0;
    if (B.isref) {
      // R = K_uu - V'V  (lower), chol, inverse
      for (int idx = tid; idx < m * m; idx += NT) {
        const int i = idx / m, j = idx - i * m;
        if (j <= i) {
          double acc = cov_entry(cp, sx[P + i], sy[P + i], smv[P + i], sx[P + j], sy[P + j], smv[P + j]);
          for (int k = 0; k < P; ++k) acc -= KV[k * m + i] * KV[k * m + j];
          R[idx] = acc;
        } else {
          R[idx] = 0.0;
        }
      }
      chol_lower_inplace(R, m, &s_fail);
      tri_inverse_lower(R, Ri, m);
      // panel_u = [ -Ri*T | Ri ]
      for (int idx = tid; idx < m * P; idx += NT) {
        const int i = idx / P, k = idx - i * P;
        double acc = 0.0;
        for (int j = 0; j <= i; ++j) acc += Ri[i * m + j] * Tt[j * P + k];
        pu[(size_t)i * ld + k] = -acc;
      }
      for (int idx = tid; idx < m * m; idx += NT) {
        const int i = idx / m, j = idx - i * m;
        pu[(size_t)i * ld + P + j] = Ri[idx];
      }
      // e = Ri (w_u - H w_pa)
      for (int i = tid; i < m; i += NT) {
        double acc = 0.0;
        for (int j = 0; j <= i; ++j) acc += Ri[i * m + j] * (wv[P + j] - hv[j]);
        wcore_part += acc * acc;
        logdet_part += log(Ri[i * m + i]);
      }
    } else {
      // non-reference level: rows conditionally independent (spamtree_model.cpp:923-963)
      for (int i = tid; i < m; i += NT) {
        double acc = cov_entry(cp, sx[P + i], sy[P + i], smv[P + i], sx[P + i], sy[P + i], smv[P + i]);
        for (int k = 0; k < P; ++k) acc -= KV[k * m + i] * KV[k * m + i];
        if (!(acc > 0.0)) s_fail = 1;
        const double r = 1.0 / sqrt(acc);
        rd[i] = r;
        pu[(size_t)i * ld + P] = r;
        const double e = r * (wv[P + i] - hv[i]);
        wcore_part += e * e;
        logdet_part += log(r);
      }
      __syncthreads();
      for (int idx = tid; idx < m * P; idx += NT) {
        const int i = idx / P, k = idx - i * P;
        pu[(size_t)i * ld + k] = -rd[i] * Tt[idx];
      }
    }
    const double wcore = block_sum(wcore_part, s_red);
    const double logdet = block_sum(logdet_part, s_red);
    __syncthreads();
    if (tid == 0) {
      A.logdet_c[b] = logdet;
      A.loglik_c[b] = (double)m * HL2PI - 0.5 * wcore;
      if (s_fail) atomicMin(A.errflag, B.level * 16 + (J == 0 ? 1 : (B.isref ? 2 : 3)));
    }
  }
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
__device__ __forceinline__ double wave_allsum(double x);
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


// wave_chol_eliminate for a 16 x 16 (or smaller) diagonal tile that sits inside a larger matrix: strides as parameters.
//   Am: tile's first element, row stride lda, lower triangle valid.  Bm: receives L^{-1} (lower triangle), row stride ldb.
//   mr <= 16 rows; rows >= mr behave as identity and are not written.  All 64 lanes must call.
__device__ __forceinline__ void wave_chol_eliminate_tile(const double *Am, int lda, double *Bm, int ldb, int mr, int *fail, int lane) {
  constexpr int MM = 16;
  double a[MM], b[MM];
  const bool row = lane < mr;
  const int lr = min(lane, MM - 1);
#pragma unroll
  for (int j = 0; j < MM; ++j) {
    a[j] = (row && j <= lane) ? Am[(size_t)lr * lda + j] : (j == lane ? 1.0 : 0.0);
    b[j] = j == lane ? 1.0 : 0.0;
  }
  double dd = 1.0;
  bool bad = false;
#pragma unroll
  for (int k = 0; k < MM; ++k) {
    if (k < mr) {   // wave-uniform
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
      if (j <= lane) Bm[(size_t)lane * ldb + j] = b[j] * rs;
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
__device__ unsigned long long g_stamps[16];
#define STAMP_DECL unsigned long long st_acc[16] = {0,0,0,0,0,0,0,0,0,0,0,0,0,0,0,0}; unsigned long long st_t0 = clock64(), st_t1 = 0;
#define STAMP(slot) do { st_t1 = clock64(); st_acc[slot] += st_t1 - st_t0; st_t0 = st_t1; } while (0)
__device__ int g_stamp_level = -1;   // >= 0: only workgroups of that tree level report (k_factor_quad)
#define STAMP_FLUSH_IF(c_) do { if (threadIdx.x == 0 && (c_)) { for (int q_ = 0; q_ < 16; ++q_) atomicAdd(&g_stamps[q_], st_acc[q_]); } } while (0)
#define STAMP_FLUSH STAMP_FLUSH_IF(g_stamp_level < 0)
#define STAMP_FLUSH_LEVEL(l_) STAMP_FLUSH_IF(g_stamp_level < 0 || g_stamp_level == (l_))
extern "C" int st_debug_stamp_level(int level) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamp_level), &level, sizeof(int)); return 0; }
extern "C" int st_debug_stamps(unsigned long long *out, int reset) {
  if (out) (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_stamps), sizeof(unsigned long long) * 16);
  if (reset) { unsigned long long z[16] = {0}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof(z)); }
  return 0;
}
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

struct FastArgs {
  const Blk *blks;
  const int *anc_idx;
  const Grp *grps;
  int ngrp;
  const double *cx, *cy;
  const int *mv;
  const double *w;
  double *panels;
  double *logdet_c, *loglik_c;
  int *errflag;
  const long long *gdesc;   // group descriptors of this launch's first group onwards
  int gd_stride;
  int Pm4, ldKV, ldS, SRm, stage_dbl;
};

#define FM_VPART 512

__global__ __launch_bounds__(NT, 2) void k_factor_mfma(FastArgs A, CovPar cp) {
  extern __shared__ double lds[];
  __shared__ int s_am[MAXJ], s_ao[MAXJ + 1];
  __shared__ long long s_arow[MAXJ], s_apan[MAXJ];
  __shared__ int s_fail;
  __shared__ double s_red[NT / 64];
  __shared__ long long s_bpan[32], s_brow[32];   // panel offset / first row of the group's blocks
  __shared__ int s_bld[32];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int jt = wid & 1, kh = wid >> 1;
  const int Pm4 = A.Pm4, ldKV = A.ldKV, ldS = A.ldS, SRm = A.SRm;
  double *KV = lds;
  double *stage = KV + (size_t)Pm4 * ldKV + 16;
  double *Vpart = stage + A.stage_dbl;
  double *zrow = Vpart - (ldS + 16);   // a row of zeros at the end of the stage area (never overwritten)
  double *colx = Vpart + FM_VPART, *coly = colx + 32, *colw = coly + 32, *hv = colw + 32, *rd = hv + 32;
  int *colmv = (int *)(rd + 32);
  int *colblk = colmv + 32;

  STAMP_DECL
  // workgroups are dealt round-robin over the 8 XCDs (each with its own L2): give every XCD a contiguous run of
  // groups so that siblings, which stream the same ancestor panels, meet in one L2
  int gidx = blockIdx.x;
  {
    const int per = A.ngrp >> 3;
    if (gidx < per * 8) gidx = (gidx & 7) * per + (gidx >> 3);
  }
  long long *s_gd = (long long *)stage;   // the group's descriptor lands in the (still unused) stage area: one round trip
  for (int i = tid; i < A.gd_stride; i += NT) s_gd[i] = A.gdesc[(size_t)gidx * A.gd_stride + i];
  __syncthreads();
  const GdHead B0 = gd_unpack(s_gd, tid, s_am, s_ao, s_arow, s_apan, nullptr, s_bpan, s_brow, s_bld, nullptr);
  const Grp G = {B0.row0, B0.blk0, B0.nblk, B0.M, B0.P};
  const int M = G.M, P = G.P, J = B0.nanc;
  const long long b0_panel_off = s_gd[8 + 4 * J];
  const int b0_ld = (int)s_gd[8 + 4 * J + 2];
  if (tid == 0) s_fail = 0;
  __syncthreads();
  // ---- prologue: coordinates (ancestors alias the stage area), K_{pa,u} into KV, pads zeroed
  {
    double *sx = stage, *sy = stage + Pm4;
    int *smv = (int *)(stage + 2 * (size_t)Pm4);
    for (int t = 0; t < J; ++t) {
      const long long r0 = s_arow[t];
      const int oa = s_ao[t];
      for (int i = tid; i < s_am[t]; i += NT) { sx[oa + i] = A.cx[r0 + i]; sy[oa + i] = A.cy[r0 + i]; smv[oa + i] = A.mv[r0 + i]; }
    }
    if (tid < 32) {
      const int j = tid;
      if (j < M) {
        const long long r = G.row0 + j;
        colx[j] = A.cx[r]; coly[j] = A.cy[r]; colw[j] = A.w[r]; colmv[j] = A.mv[r];
        int bi = 0;
        while (bi + 1 < G.nblk && r >= s_brow[bi + 1]) ++bi;
        colblk[j] = bi;
      } else {
        colx[j] = 0.0; coly[j] = 0.0; colw[j] = 0.0; colmv[j] = 0; colblk[j] = 0;
      }
    }
    for (int k = tid; k < ldS + 16; k += NT) zrow[k] = 0.0;
    __syncthreads();
    const float invld = 1.0f / (float)ldKV;
    for (int idx = tid; idx < Pm4 * ldKV + 16; idx += NT) {
      const int k = (int)(((float)idx + 0.5f) * invld), j = idx - k * ldKV;   // exact: idx < 2^14, ldKV <= 32
      KV[idx] = (k < P && j < M) ? cov_entry(cp, sx[k], sy[k], smv[k], colx[j], coly[j], colmv[j]) : 0.0;
    }
  }
  d4 acc[8];
#pragma unroll
  for (int n = 0; n < 8; ++n) acc[n] = (d4){0.0, 0.0, 0.0, 0.0};
  STAMP(0);

  // ---- one pass over the ancestor chain, last ancestor first, in sub-panels of <= 16 rows.  The next
  // sub-panel is fetched from global memory into registers while the matrix cores work on the current one.
  {
    double pre[16];   // rows wid, wid+4, wid+8, wid+12 of the sub-panel x 4 chunks of 64 columns
    auto sub_geom = [&](int t, int s, int &r0, int &sr, int &Kb) {
      const int ma = s_am[t];
      const int sr0 = ma > 16 ? (ma + 1) >> 1 : ma;
      r0 = s == 0 ? 0 : sr0;
      sr = s == 0 ? sr0 : ma - sr0;
      Kb = s_ao[t] + ma;
    };
    auto fetch = [&](int t, int s) {
      int r0, sr, Kb;
      sub_geom(t, s, r0, sr, Kb);
      const double *src = A.panels + s_apan[t] + (size_t)(r0 + wid) * Kb + lane;
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#ifdef FM_NOFETCH
          pre[rr * 4 + c] = 1e-3;
#else
          pre[rr * 4 + c] = (wid + 4 * rr < sr && lane + 64 * c < Kb) ? src[(size_t)(4 * rr) * Kb + 64 * c] : 0.0;
#endif
        }
      }
    };
    int t = J - 1, s = 0;
    if (t >= 0) fetch(t, s);
    d4 vt0 = (d4){0.0, 0.0, 0.0, 0.0}, vt1 = vt0;
    while (t >= 0) {
      const int ma = s_am[t], oa = s_ao[t];
      const int nsub = ma > 16 ? 2 : 1;
      int r0, sr, Kb;
      sub_geom(t, s, r0, sr, Kb);
      __syncthreads();  // everyone is done with the previous contents of `stage` (and with the prologue alias)
      STAMP(1);
      {
        double *dst = stage + (size_t)wid * ldS + lane;
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
#pragma unroll
          for (int c = 0; c < 4; ++c)
            if (wid + 4 * rr < sr && lane + 64 * c < Kb) dst[(size_t)(4 * rr) * ldS + 64 * c] = pre[rr * 4 + c];
        }
        if (tid < sr * 4) stage[(size_t)(tid >> 2) * ldS + Kb + (tid & 3)] = 0.0;   // k in [Kb, Kb+4) reads as zero
      }
      int tn = t, sn = s + 1;
      if (sn >= nsub) { tn = t - 1; sn = 0; }
      if (tn >= 0) fetch(tn, sn);
      __syncthreads();
      STAMP(2);
      // V_sub partial over this wave pair's half of K.  Rows >= sr read the zero row, columns in [Kb, Kb+4)
      // were zero-filled, so the loop body is two LDS reads and one MFMA.
      const int ns = (Kb + 3) >> 2, nh = (ns + 1) >> 1;
      const int st0 = kh ? nh : 0, st1 = kh ? ns : nh;
      d4 p = (d4){0.0, 0.0, 0.0, 0.0};
      {
        const double *ap = ((l15 < sr) ? stage + (size_t)l15 * ldS : zrow) + 4 * st0 + l4;
        const double *bp = KV + (size_t)(4 * st0 + l4) * ldKV + jt * 16 + l15;
        const int bstep = 4 * ldKV;
        int st = st0;
        for (; st + 4 <= st1; st += 4) {
          const double a0 = ap[0], a1 = ap[4], a2 = ap[8], a3 = ap[12];
          const double b0 = bp[0], b1 = bp[bstep], b2 = bp[2 * bstep], b3 = bp[3 * bstep];
          p = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, p, 0, 0, 0);
          p = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, p, 0, 0, 0);
          p = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, b2, p, 0, 0, 0);
          p = __builtin_amdgcn_mfma_f64_16x16x4f64(a3, b3, p, 0, 0, 0);
          ap += 16; bp += 4 * bstep;
        }
        for (; st < st1; ++st) {
          p = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[0], bp[0], p, 0, 0, 0);
          ap += 4; bp += bstep;
        }
      }
      STAMP(3);
      if (kh == 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) Vpart[jt * 256 + r * 64 + lane] = p[r];
      }
      __syncthreads();
      if (kh == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          p[r] += Vpart[jt * 256 + r * 64 + lane];
          Vpart[jt * 256 + r * 64 + lane] = p[r];
        }
      }
      __syncthreads();
      if (kh == 1) {
#pragma unroll
        for (int r = 0; r < 4; ++r) p[r] = Vpart[jt * 256 + r * 64 + lane];
      }
      if (s == 0) vt0 = p; else vt1 = p;
      STAMP(4);
      // T^T tiles (kt = kh, kh+2, ...) += Linv_sub^T * V_sub ; the V tile (C layout) is the B operand.
      {
        const int nst = (sr + 3) >> 2;
        const int kb0 = kh * 16 + l15;
        const double *r0p = ((l4 < sr) ? stage + (size_t)l4 * ldS : zrow) + kb0;
        const double *r1p = ((4 + l4 < sr) ? stage + (size_t)(4 + l4) * ldS : zrow) + kb0;
        const double *r2p = ((8 + l4 < sr) ? stage + (size_t)(8 + l4) * ldS : zrow) + kb0;
        const double *r3p = ((12 + l4 < sr) ? stage + (size_t)(12 + l4) * ldS : zrow) + kb0;
#pragma unroll
        for (int n = 0; n < 8; ++n) {
          const int kt = kh + 2 * n;
          if (kt * 16 < Kb) {
            const bool kok = kb0 + 32 * n < Kb;   // the boundary tile must not touch T columns of later panels
            double a0 = r0p[32 * n], a1 = r1p[32 * n], a2 = r2p[32 * n], a3 = r3p[32 * n];
            a0 = kok ? a0 : 0.0; a1 = kok ? a1 : 0.0; a2 = kok ? a2 : 0.0; a3 = kok ? a3 : 0.0;
            acc[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, p[0], acc[n], 0, 0, 0);
            if (nst > 1) acc[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, p[1], acc[n], 0, 0, 0);
            if (nst > 2) acc[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a2, p[2], acc[n], 0, 0, 0);
            if (nst > 3) acc[n] = __builtin_amdgcn_mfma_f64_16x16x4f64(a3, p[3], acc[n], 0, 0, 0);
          }
        }
      }
      // after the panel's last sub-panel its V rows replace the K rows they were computed from
      // (later panels read only rows < oa)
      if (s == nsub - 1 && kh == 0) {
        const int sr0 = ma > 16 ? (ma + 1) >> 1 : ma;
        const int j = jt * 16 + l15;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = l4 + 4 * r;
          if (j < ldKV) {
            if (i < sr0) KV[(size_t)(oa + i) * ldKV + j] = (nsub == 2) ? vt0[r] : p[r];
            if (nsub == 2 && i < ma - sr0) KV[(size_t)(oa + sr0 + i) * ldKV + j] = vt1[r];
          }
        }
      }
      STAMP(5);
      t = tn; s = sn;
    }
  }
  __syncthreads();
  STAMP(6);

  const bool refgrp = B0.isref != 0;
  double *R = stage, *Ri = stage + 32 * CH_LD;              // row stride CH_LD
  double *chcol = Ri + 32 * CH_LD, *chrs = chcol + 216;       // elimination scratch: 2 x 3 x 36 published entries, pivots
  if (refgrp) {
    // ---- R = K_uu - V'V : wave -> tile (it, jt2)
    const int it = wid >> 1, jt2 = wid & 1;
    if (it * 16 < M && jt2 * 16 < M) {
      d4 c = (d4){0.0, 0.0, 0.0, 0.0};
      const double *ap = KV + (size_t)l4 * ldKV + it * 16 + l15;
      const double *bp = KV + (size_t)l4 * ldKV + jt2 * 16 + l15;
      const int stp = 4 * ldKV;
      for (int st = 0; st < (Pm4 >> 2); ++st) {
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[0], bp[0], c, 0, 0, 0);
        ap += stp; bp += stp;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = it * 16 + l4 + 4 * r, j = jt2 * 16 + l15;
        if (i < M && j < M)
          R[i * CH_LD + j] = (j <= i) ? cov_entry(cp, colx[i], coly[i], colmv[i], colx[j], coly[j], colmv[j]) - c[r] : 0.0;
      }
    }
    for (int idx = tid; idx < 32 * CH_LD; idx += NT) Ri[idx] = (idx / CH_LD == idx % CH_LD) ? 1.0 : 0.0;
  } else {
    if (tid < M) {
      const int j = tid;
      double d = cov_entry(cp, colx[j], coly[j], colmv[j], colx[j], coly[j], colmv[j]);
      for (int k = 0; k < P; ++k) { const double v = KV[(size_t)k * ldKV + j]; d -= v * v; }
      if (!(d > 0.0)) s_fail = 1;
      rd[j] = 1.0 / sqrt(d);
    }
  }
  __syncthreads();
  STAMP(7);
  // ---- dump T^T into the KV buffer (same [k][ldKV] layout); pads zero
#pragma unroll
  for (int n = 0; n < 8; ++n) {
    const int kt = kh + 2 * n;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = kt * 16 + l4 + 4 * r, j = jt * 16 + l15;
      if (k < Pm4 && j < ldKV) KV[(size_t)k * ldKV + j] = (k < P && j < M) ? acc[n][r] : 0.0;
    }
  }
  double *wpa = Vpart;  // P <= 256 <= FM_VPART
  for (int t = 0; t < J; ++t)
    for (int i = tid; i < s_am[t]; i += NT) wpa[s_ao[t] + i] = A.w[s_arow[t] + i];
  __syncthreads();
  // ---- Ri = chol(R)^{-1}: workgroup-wide elimination in LDS; then hv = T w_pa
  if (refgrp) {   // element slots per thread: m (m + 1) <= 256 * slots
    if (M <= 22) team_chol_eliminate<2, NT>(R, Ri, M, M, chcol, &s_fail, tid);
    else if (M <= 27) team_chol_eliminate<3, NT>(R, Ri, M, M, chcol, &s_fail, tid);
    else team_chol_eliminate<5, NT>(R, Ri, M, M, chcol, &s_fail, tid);
  }
  {
    const int w0 = wid, nw = 4;
    for (int j = w0; j < M; j += nw) {
      double a = 0.0;
      for (int k = lane; k < P; k += 64) a += KV[(size_t)k * ldKV + j] * wpa[k];
      a = wave_sum(a);
      if (lane == 0) hv[j] = a;
    }
  }
  __syncthreads();

  STAMP(8);
  double wcore_part = 0.0, logdet_part = 0.0;
  if (refgrp) {
    double *pu = A.panels + b0_panel_off;
    const int ld = b0_ld;
    // ---- N = -Ri * T : tiles (it, kt), A[i][j] = -Ri[i][j] (lower), B[j][k] = T^T[k][j]
    const int nkt = (P + 15) >> 4, nit = (M + 15) >> 4;
    for (int tile = wid; tile < nit * nkt; tile += NT / 64) {
      const int it = tile % nit, kt = tile / nit;
      const int njs = (min(M, it * 16 + 16) + 3) >> 2;
      const int i = it * 16 + l15;
      const int krow = min(kt * 16 + l15, Pm4 - 1);
      d4 c = (d4){0.0, 0.0, 0.0, 0.0};
      for (int st = 0; st < njs; ++st) {
        const int j = 4 * st + l4;
        const double a = (i < M && j <= i) ? -Ri[i * CH_LD + j] : 0.0;
        const double b = KV[(size_t)krow * ldKV + j];
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int io = it * 16 + l4 + 4 * r, k = kt * 16 + l15;
        if (io < M && k < P) pu[(size_t)io * ld + k] = c[r];
      }
    }
    for (int idx = tid; idx < M * M; idx += NT) {
      const int i = idx / M, j = idx - i * M;
      pu[(size_t)i * ld + P + j] = (j <= i) ? Ri[i * CH_LD + j] : 0.0;
    }
    if (tid < M) {
      const int i = tid;
      double e = 0.0;
      for (int j = 0; j <= i; ++j) e += Ri[i * CH_LD + j] * (colw[j] - hv[j]);
      wcore_part = e * e;
      logdet_part = log(Ri[i * CH_LD + i]);
    }
    const double wcore = block_sum(wcore_part, s_red);
    const double logdet = block_sum(logdet_part, s_red);
    if (tid == 0) {
      A.logdet_c[G.blk0] = logdet;
      A.loglik_c[G.blk0] = (double)M * HL2PI - 0.5 * wcore;
      if (s_fail) atomicMin(A.errflag, B0.level * 16 + (J == 0 ? 1 : 2));
    }
  } else {
    // non-reference rows: panel row of column j = [ -r_j * T[j][:] | r_j ] in its own block
    for (int j = wid; j < M; j += NT / 64) {   // one wave per column: coalesced row of the block's panel
      const int bi = colblk[j];
      double *prow = A.panels + s_bpan[bi] + (size_t)(G.row0 + j - s_brow[bi]) * s_bld[bi];
      const double r = rd[j];
      for (int k = lane; k < P; k += 64) prow[k] = -r * KV[(size_t)k * ldKV + j];
      if (lane == 0) prow[P] = r;
    }
    if (tid < G.nblk) {
      const int bi = tid;
      double wc = 0.0, ldt = 0.0;
      int cnt = 0;
      for (int j = 0; j < M; ++j)
        if (colblk[j] == bi) {
          const double e = rd[j] * (colw[j] - hv[j]);
          wc += e * e;
          ldt += log(rd[j]);
          ++cnt;
        }
      A.logdet_c[G.blk0 + bi] = ldt;
      A.loglik_c[G.blk0 + bi] = (double)cnt * HL2PI - 0.5 * wc;
    }
    if (tid == 0 && s_fail) atomicMin(A.errflag, B0.level * 16 + 3);
  }
  STAMP(9);
  STAMP_FLUSH;
}


#include "chol_blocked.hpp"
#include "factor_quad.hpp"
#include "factor_big.hpp"
#include "factor_wide.hpp"
#include "factor_lchain.hpp"

// w = L^{-T} (L^{-1} b + z), S = L L' (spamtree_model.cpp:1054, 1086), for blocks too wide for one wave's registers (75-row
// blocks of the default multivariate tree), by ONE wave with S in LDS and NO workgroup barrier: the scratch-arena path pays
// three __syncthreads() per pivot for the factorisation and four per row for the two substitutions (about a millisecond per
// 75-row block, most of phase B at config #4).  Lane i owns rows i and i + 64 (m <= 128); right-looking elimination, the
// scaled pivot column goes through `lcol` (m doubles of LDS) so that a batch of eight trailing columns costs four wide
// uniform reads; the right-hand side rides along (forward substitution for free); the backward substitution walks L by rows.
//   S: m x m (m <= 80), row stride ms (odd: conflict-free column walks; >= m + 7), lower triangle valid, destroyed.
//   lcol: m + 8 doubles.  bv (LDS): in b, out w.
//   zg: the block's normals (global).  All 64 lanes of the wave must call; nobody else may touch S, lcol, bv meanwhile.
__device__ __forceinline__ void wave_chol_solve_lds(double *S, int ms, double *lcol, double *bv, const double *zg, int m, int *fail, int lane) {
  // rows 0 .. 63: lane i owns row i.  Rows 64 .. m - 1 (at most 16: m <= 80): lane (g, r) = (lane >> 4, lane & 15) works on
  // row 64 + r, and the four lane groups take DIFFERENT column batches of one trip (eleven rows on a row-per-lane mapping
  // would pay a whole wave-instruction stream for 17 % of its lanes)
  const int i0 = lane, i1 = 64 + (lane & 15), g1 = lane >> 4;
  const bool r0 = i0 < m, r1 = i1 < m;
  const bool two = m > 64;   // wave-uniform
  double c0 = r0 ? bv[i0] : 0.0, c1 = r1 ? bv[i1] : 0.0;   // c1, dr1, t1: replicated in the four lanes of a row
  double dr0 = 1.0, dr1 = 1.0;   // 1 / L_kk of this lane's rows
  bool bad = false;
  for (int j = lane; j < m + 8; j += 64) lcol[j] = 0.0;
  for (int k = 0; k < m; ++k) {
    const double d = S[(size_t)k * ms + k];
    bad = bad || !(d > 0.0);
    double rp = __builtin_amdgcn_rsq(d);            // 1 / sqrt(d): hardware seed + two Newton steps (relative error < 1e-16)
    rp = rp * fma(-0.5 * d * rp, rp, 1.5);
    rp = rp * fma(-0.5 * d * rp, rp, 1.5);
    const bool u0 = r0 && i0 > k, u1 = r1 && i1 > k;
    const double l0 = u0 ? S[(size_t)i0 * ms + k] * rp : 0.0;
    const double l1 = u1 ? S[(size_t)i1 * ms + k] * rp : 0.0;
    const double ck = readlane_f64(k < 64 ? c0 : c1, k & 63);   // (row 64 + r: lane r of group 0)
    const double yk = ck * rp;
    c0 = (i0 == k) ? yk : fma(-l0, yk, c0);
    c1 = (i1 == k) ? yk : fma(-l1, yk, c1);
    dr0 = (i0 == k) ? rp : dr0;
    dr1 = (i1 == k) ? rp : dr1;
    if (u0) { S[(size_t)i0 * ms + k] = l0; lcol[i0] = l0; }
    if (u1 && g1 == 0) { S[(size_t)i1 * ms + k] = l1; lcol[i1] = l1; }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // one wave: its LDS operations complete in order
    // trailing columns in batches of eight, WITHOUT per-element predicates: rows at or above the pivot carry l = 0 (no-ops),
    // entries above the diagonal and the columns past m - 1 that a batch reaches (row stride ms >= m + 7, lcol: m + 8) are
    // scribbled on and never read
    if (r0) {
      double *row = S + (size_t)i0 * ms;
      for (int j0 = k + 1; j0 < m; j0 += 8) {
        // (requesting the next batch before this one's stores would save about a tenth of the solve, but its 16 extra VGPRs
        // take the kernel past 128 and the leaf levels, which share it, from four waves per SIMD to three: measured, dropped)
        double lj[8], a0[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) lj[q] = lcol[j0 + q];
#pragma unroll
        for (int q = 0; q < 8; ++q) a0[q] = row[j0 + q];
#pragma unroll
        for (int q = 0; q < 8; ++q) row[j0 + q] = fma(-l0, lj[q], a0[q]);
      }
    }
    if (two) {   // wave-uniform
      if (r1) {
        double *row = S + (size_t)i1 * ms;
        for (int j0 = k + 1 + 8 * g1; j0 < m; j0 += 32) {
          double lj[8], a1[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) lj[q] = lcol[j0 + q];
#pragma unroll
          for (int q = 0; q < 8; ++q) a1[q] = row[j0 + q];
#pragma unroll
          for (int q = 0; q < 8; ++q) row[j0 + q] = fma(-l1, lj[q], a1[q]);
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  }
  if (bad && lane == 0) *fail = 1;
  // S now holds L below the diagonal (column k scaled by 1 / L_kk); c = L^{-1} b.  Backward: w_k = t_k / L_kk, t_j -= L_kj w_k
  double t0 = r0 ? c0 + zg[i0] : 0.0, t1 = r1 ? c1 + zg[i1] : 0.0;
  for (int k = m - 1; k >= 0; --k) {
    const double wk = readlane_f64(k < 64 ? t0 : t1, k & 63) * readlane_f64(k < 64 ? dr0 : dr1, k & 63);
    t0 = (i0 == k) ? wk : ((r0 && i0 < k) ? fma(-S[(size_t)k * ms + i0], wk, t0) : t0);
    if (two) t1 = (i1 == k) ? wk : ((r1 && i1 < k) ? fma(-S[(size_t)k * ms + i1], wk, t1) : t1);
  }
  if (r0) bv[i0] = t0;
  if (r1 && g1 == 0) bv[i1] = t1;
}

struct SampleArgs {
  const Blk *blks;
  const int *anc_idx;
  const int *dch_idx;
  const int *list;
  int nlist;
  const double *panels;  // param_data slot
  double *w;
  const double *y, *xb, *z;
  const int *mv;
  const unsigned char *obs;
  double *acc;           // message arena
  int *errflag;
  double *scratch;       // BIG: staged S matrix
  long long scratch_stride;
  int maxP, maxM, maxLd;
  int do_gram;
  int no_fwd;   // limited_tree: a block's record goes to its single parent only, nothing is forwarded from its children
  int lds_sq;   // BIG: 2 = the posterior precision lives in LDS after the vectors (maxM x (maxM | 1) + maxM doubles) and ONE wave
                // factorises and solves there (wave_chol_solve_lds); 0 = scratch arena + workgroup-wide loops
  double *s0;              // theta-only part Ri' Ri of every reference block's posterior precision, cached like the records'
  const long long *s0off;  // Gram parts: per block, offset into s0 (row stride m) or -1
  double tausq_inv[QMAX];
};

// NOREF: the level holds non-reference blocks only (the host knows): the reference branch -- whose blocked factorisation takes
// 224 VGPRs -- is compiled out, so that leaf levels keep four waves per SIMD
template <bool BIG, bool NOREF = false>
__global__ __launch_bounds__(NT) void k_sample(SampleArgs A) {
  extern __shared__ double lds[];
  __shared__ int s_am[MAXJ], s_ao[MAXJ + 1];
  __shared__ long long s_arow[MAXJ], s_aoff[MAXJ + 1];
  __shared__ int s_fail;
  __shared__ long long s_choff[16];    // record offsets of the first direct children (one read per block instead of one per entry)

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int maxP = A.maxP, maxM = A.maxM;
  double *wv = lds;                    // maxP + maxM
  double *tv = wv + (maxP + maxM);     // maxM   N w_pa
  double *ev = tv + maxM;              // maxM   Ri w_u + N w_pa
  double *bv = ev + maxM;              // maxM   rhs / solution
  double *av = bv + maxM;              // maxM   per-ancestor temp
  double *seg = av + maxM;             // MAXJ * maxM: seg[t][r] = sum_j N[r][oa_t + j] w_a[j], later ev[r] - seg[t][r]
  double *Np = seg + (size_t)MAXJ * maxM;   // !BIG: maxM * maxLd panel copy
  double *S = BIG ? (A.lds_sq ? Np : A.scratch + (size_t)blockIdx.x * A.scratch_stride) : (Np + (size_t)maxM * A.maxLd);

  STAMP_DECL
  int st_lev = 0;
  for (int li = blockIdx.x; li < A.nlist; li += gridDim.x) {
    const int b = A.list[li];
    const Blk B = A.blks[b];
    const int m = B.m, P = B.P, J = B.nanc, ld = B.ld;
    st_lev = B.level;
    __syncthreads();
    STAMP(7);
    if (tid < J) {
      const int a = A.anc_idx[B.anc_ptr + tid];
      s_am[tid] = A.blks[a].m;
      s_arow[tid] = A.blks[a].row0;
    }
    if (tid == 0) s_fail = 0;
    if (tid >= 64 && tid < 64 + min(B.ndch, 16)) s_choff[tid - 64] = A.blks[A.dch_idx[B.dch_ptr + tid - 64]].acc_off;
    __syncthreads();
    if (tid == 0) {
      int o = 0;
      long long ao = 0;
      for (int t = 0; t < J; ++t) { s_ao[t] = o; s_aoff[t] = ao; o += s_am[t]; ao += (long long)s_am[t] * s_am[t] + s_am[t]; }
      s_ao[J] = o; s_aoff[J] = ao;
    }
    __syncthreads();
    for (int t = 0; t < J; ++t) {
      const long long r0 = s_arow[t];
      for (int i = tid; i < s_am[t]; i += NT) wv[s_ao[t] + i] = A.w[r0 + i];
    }
    const double *pg = A.panels + B.panel_off;
    const double *N;  // m x ld, row-major: [ -Ri*H | Ri or r ]
    if (BIG) {
      N = pg;
    } else {
      for (int idx = tid; idx < m * ld; idx += NT) Np[idx] = pg[idx];
      N = Np;
    }
    __syncthreads();
    for (int i0 = 4 * wid; i0 < m; i0 += 4 * (NT / 64)) {   // four rows per trip: their loads travel together
      double a4[4] = {0.0, 0.0, 0.0, 0.0};
      for (int k = lane; k < P; k += 64) {
        const double wk = wv[k];
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (i0 + q < m) a4[q] += N[(size_t)(i0 + q) * ld + k] * wk;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const double r = wave_sum(a4[q]);
        if (lane == 0 && i0 + q < m) tv[i0 + q] = r;
      }
    }
    __syncthreads();

    STAMP(0);
    if (!NOREF && B.isref) {
      const double *Ri = N + P;  // Ri[i][j] = N[i*ld + P + j]
      // Sigi_tot = Ri'Ri + sum_children Sigi_children + diag(tausq_inv)      (:1044-1051)
      const int ms = (BIG && A.lds_sq) ? ((m + 7) | 1) : m;   // row stride of S
      const long long so = (BIG && A.s0off) ? A.s0off[b] : -1;
#pragma unroll 4
      for (int idx = tid; idx < m * m; idx += NT) {
        const int i = idx / m, j = idx - i * m;
        if (j <= i) {
          double acc = 0.0;
          if (so >= 0 && !A.do_gram) acc = A.s0[so + idx];   // Ri' Ri is a function of theta only
          else {
            for (int k = i; k < m; ++k) acc += Ri[(size_t)k * ld + i] * Ri[(size_t)k * ld + j];
            if (so >= 0) A.s0[so + idx] = acc;
          }
          for (int c = 0; c < B.ndch; ++c) {
            const long long co = c < 16 ? s_choff[c] : A.blks[A.dch_idx[B.dch_ptr + c]].acc_off;
            acc += A.acc[co + B.acc_len + idx];
          }
          if (i == j) acc += A.tausq_inv[A.mv[B.row0 + i]];
          if (BIG && A.lds_sq) Np[(size_t)i * ms + j] = acc; else S[(size_t)i * ms + j] = acc;
        } else {
          if (BIG && A.lds_sq) Np[(size_t)i * ms + j] = 0.0; else S[(size_t)i * ms + j] = 0.0;
        }
      }
      // Smu_tot = A_u' w_pa + sum_children Smu_children + tausq_inv*(y - XB)   (:1062-1077)
      for (int i = tid; i < m; i += NT) {
        double acc = 0.0;
        // - Ri' (N w_pa), walking Ri by ROWS (thread i reads entry i of row k: coalesced; same summation order as the column walk)
        for (int k0 = 0; k0 < m; k0 += 8) {
          double x[8];
#pragma unroll
          for (int q = 0; q < 8; ++q) x[q] = (k0 + q < m && k0 + q >= i) ? Ri[(size_t)(k0 + q) * ld + i] : 0.0;
#pragma unroll
          for (int q = 0; q < 8; ++q)
            if (k0 + q < m && k0 + q >= i) acc -= x[q] * tv[k0 + q];
        }
        for (int c = 0; c < B.ndch; ++c) {
          const long long co = c < 16 ? s_choff[c] : A.blks[A.dch_idx[B.dch_ptr + c]].acc_off;
          acc += A.acc[co + B.acc_len + m * m + i];
        }
        const long long r = B.row0 + i;
        acc += A.tausq_inv[A.mv[r]] * (A.y[r] - A.xb[r]);
        bv[i] = acc;
      }
      if (BIG && A.lds_sq) {
        // w_u = L^{-T} (L^{-1} Smu + z) by wave 0 alone, S in LDS (no workgroup barrier inside)
        __syncthreads();
        STAMP(1);
        // (Np, not S: S is `lds_sq ? LDS : scratch arena`, a generic pointer -- the compiler would emit FLAT loads and stores,
        // six times slower than ds_read / ds_write here)
        if (m > 32) block_chol_solve_mfma(Np, ms, Np + (size_t)maxM * ((maxM + 7) | 1), bv, A.z + B.row0, m, &s_fail);   // (workgroup-uniform)
        else if (wid == 0) wave_chol_solve_lds(Np, ms, Np + (size_t)maxM * ((maxM + 7) | 1), bv, A.z + B.row0, m, &s_fail, lane);
      } else {
      chol_lower_inplace(S, m, &s_fail);
      // w_u = L^{-T} (L^{-1} Smu + z)   (= Sigi_chol' (Sigi_chol Smu + z), :1086)
      for (int k = 0; k < m; ++k) {
        __syncthreads();
        const double xk = bv[k] / S[k * m + k];
        __syncthreads();
        if (tid == 0) bv[k] = xk;
        for (int i = k + 1 + tid; i < m; i += NT) bv[i] -= S[i * m + k] * xk;
      }
      __syncthreads();
      for (int i = tid; i < m; i += NT) bv[i] += A.z[B.row0 + i];
      for (int k = m - 1; k >= 0; --k) {
        __syncthreads();
        const double xk = bv[k] / S[k * m + k];
        __syncthreads();
        if (tid == 0) bv[k] = xk;
        for (int i = tid; i < k; i += NT) bv[i] -= S[k * m + i] * xk;
      }
      }
      __syncthreads();
      STAMP(2);
      for (int i = tid; i < m; i += NT) {
        wv[P + i] = bv[i];
        A.w[B.row0 + i] = bv[i];
      }
      __syncthreads();
      for (int i = tid; i < m; i += NT) {
        double acc = tv[i];
        for (int j = 0; j <= i; ++j) acc += Ri[(size_t)i * ld + j] * wv[P + j];
        ev[i] = acc;
      }
    } else {
      // non-reference rows (:1091-1155)
      for (int i = tid; i < m; i += NT) {
        const long long r = B.row0 + i;
        const double ri = N[(size_t)i * ld + P];
        const double tsq = A.tausq_inv[A.mv[r]];
        const double sig = ri * ri + tsq;
        if (!(sig > 0.0)) s_fail = 1;
        const double mu = -ri * tv[i] + tsq * (A.y[r] - A.xb[r]);
        const double c = 1.0 / sqrt(sig);
        const double wi = c * c * mu + c * A.z[r];
        wv[P + i] = wi;
        A.w[r] = wi;
        ev[i] = ri * wi + tv[i];
      }
    }
    __syncthreads();
    STAMP(3);
    // messages to every ancestor (:1158-1207), summed with the direct children's accumulated messages
    if (!A.do_gram) {
      // Gram parts cached (Q4): only the vectors.  All ancestors at once, as in k_sample_lean: thread (row r, ancestor t)
      // sums its segment N[r][oa_t ..] w_a, then thread k (a chain column) accumulates -sum_r N[r][k] (ev[r] - seg_t(k)[r])
      for (int idx = tid; idx < m * J; idx += NT) {
        const int r = idx / J, t = idx - r * J;
        const int ma = s_am[t], oa = s_ao[t];
        const double *row = N + (size_t)r * ld + oa;
        const double *wa = wv + oa;
        double a = 0.0;
        for (int j0 = 0; j0 < ma; j0 += 8) {
          double x[8];
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) x[jj] = (j0 + jj < ma) ? row[j0 + jj] : 0.0;
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) a += x[jj] * ((j0 + jj < ma) ? wa[j0 + jj] : 0.0);
        }
        seg[t * maxM + r] = ev[r] - a;
      }
      __syncthreads();
      STAMP(4);
      double *rec = A.acc + B.acc_off;
      for (int k = tid; k < P; k += NT) {
        int t = 0;
        while (t + 1 < J && k >= s_ao[t + 1]) ++t;
        const int ma = s_am[t], i = k - s_ao[t];
        const double *avt = seg + t * maxM;
        double a = 0.0;
        for (int r0 = 0; r0 < m; r0 += 8) {
          double x[8];
#pragma unroll
          for (int rr = 0; rr < 8; ++rr) x[rr] = (r0 + rr < m) ? N[(size_t)(r0 + rr) * ld + k] : 0.0;
#pragma unroll
          for (int rr = 0; rr < 8; ++rr) a -= x[rr] * ((r0 + rr < m) ? avt[r0 + rr] : 0.0);
        }
        for (int c = 0; c < (A.no_fwd ? 0 : B.ndch); ++c) {
          const Blk C = A.blks[A.dch_idx[B.dch_ptr + c]];
          a += A.acc[C.acc_off + s_aoff[t] + ma * ma + i];
        }
        rec[s_aoff[t] + ma * ma + i] = a;
      }
      STAMP(5);
      if (tid == 0 && s_fail) atomicMin(A.errflag, B.level * 16 + (B.isref ? 10 : 11));
      continue;
    }
    long long off = 0;
    for (int t = 0; t < J; ++t) {
      const int ma = s_am[t], oa = s_ao[t];
      for (int r = tid; r < m; r += NT) {
        double acc = ev[r];
        for (int j = 0; j < ma; ++j) acc -= N[(size_t)r * ld + oa + j] * wv[oa + j];
        av[r] = acc;
      }
      __syncthreads();
      double *out = A.acc + B.acc_off + off;
      for (int idx = A.do_gram ? tid : ma * ma + tid; idx < ma * ma + ma; idx += NT) {
        double acc = 0.0;
        if (idx < ma * ma) {
          const int i = idx / ma, j = idx - i * ma;
          for (int r = 0; r < m; ++r) acc += N[(size_t)r * ld + oa + i] * N[(size_t)r * ld + oa + j];
        } else {
          const int i = idx - ma * ma;
          for (int r = 0; r < m; ++r) acc -= N[(size_t)r * ld + oa + i] * av[r];
        }
        for (int c = 0; c < (A.no_fwd ? 0 : B.ndch); ++c) {
          const Blk C = A.blks[A.dch_idx[B.dch_ptr + c]];
          acc += A.acc[C.acc_off + off + idx];
        }
        out[idx] = acc;
      }
      off += (long long)ma * ma + ma;
      __syncthreads();
    }
    if (tid == 0 && s_fail) atomicMin(A.errflag, B.level * 16 + (B.isref ? 10 : 11));
  }
  STAMP_FLUSH_LEVEL(st_lev);
}


// The theta-only parts of the generic path's message records -- Sigma_a = N_a' N_a per ancestor a, plus the direct children's
// (spamtree_model.cpp:1162, 1190-1192) -- and of the posterior precision (Ri' Ri) on the FP64 matrix cores,
// ahead of a sweep that then takes k_sample's cached branch.  k_sample's own do_gram branch builds them with one thread per
// entry and a strided global walk per product: 35 ms for the leaf level of config #4 (16 384 blocks x 7 ancestors x 75 x 75
// entries x 36 rows) against 2.2 ms for the sweep itself.  One workgroup per block; a task = one 16 x 16 tile (it >= jt) of
// one ancestor's Gram matrix, tasks dealt over the four waves; both MFMA operands are rows of the block's panel, straight
// from global memory / L2 (16 consecutive doubles per row: whole 128-byte segments).  Fixed summation order.
struct GramBigArgs {
  const Blk *blks;
  const int *anc_idx, *dch_idx;
  const int *list;
  int nlist;
  const double *panels;
  double *acc;
  double *s0;
  const long long *s0off;
  int no_fwd;
};

__global__ __launch_bounds__(NT) void k_gram_big(GramBigArgs A) {
  __shared__ int s_am[MAXJ + 1], s_ao[MAXJ + 1], s_t0[MAXJ + 2];
  __shared__ long long s_aoff[MAXJ + 1];
  __shared__ long long s_choff[16];
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = A.list[blockIdx.x];
  const Blk B = A.blks[b];
  const int m = B.m, P = B.P, J = B.nanc, ld = B.ld;
  const long long so = (B.isref && A.s0off) ? A.s0off[b] : -1;
  if (tid < J) s_am[tid] = A.blks[A.anc_idx[B.anc_ptr + tid]].m;
  if (tid >= 64 && tid < 64 + min(B.ndch, 16)) s_choff[tid - 64] = A.blks[A.dch_idx[B.dch_ptr + tid - 64]].acc_off;
  __syncthreads();
  if (tid == 0) {
    int o = 0, tasks = 0;
    long long ao = 0;
    for (int t = 0; t < J; ++t) {
      const int nt = (s_am[t] + 15) >> 4;
      s_ao[t] = o; s_aoff[t] = ao; s_t0[t] = tasks;
      o += s_am[t]; ao += (long long)s_am[t] * s_am[t] + s_am[t]; tasks += nt * (nt + 1) / 2;
    }
    s_t0[J] = tasks;
    if (so >= 0) { s_am[J] = m; s_ao[J] = P; s_aoff[J] = 0; const int nt = (m + 15) >> 4; tasks += nt * (nt + 1) / 2; }
    s_t0[J + 1] = tasks;
  }
  __syncthreads();
  const double *N = A.panels + B.panel_off;
  double *rec = A.acc + B.acc_off;
  const int ntask = s_t0[J + 1], ns = (m + 3) >> 2;
  const int nch = A.no_fwd ? 0 : B.ndch;
  for (int e = wid; e < ntask; e += NT / 64) {
    int t = 0;
    while (e >= s_t0[t + 1]) ++t;
    int it = 0, pe = e - s_t0[t];
    while ((it + 1) * (it + 2) / 2 <= pe) ++it;
    const int jt = pe - it * (it + 1) / 2;
    const int ma = s_am[t], oa = s_ao[t];
    const int ci = 16 * it + l15, cj = 16 * jt + l15;
    const double *ap = N + (size_t)l4 * ld + oa + min(ci, ma - 1), *bp = N + (size_t)l4 * ld + oa + min(cj, ma - 1);
    d4 c = (d4){0.0, 0.0, 0.0, 0.0};
    int st = 0;
    for (; st + 4 <= ns; st += 4) {   // eight operand loads in flight per lane
      double a4[4], b4[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const bool rok = 4 * (st + q) + l4 < m;
        a4[q] = (rok && ci < ma) ? ap[(size_t)4 * (st + q) * ld] : 0.0;
        b4[q] = (rok && cj < ma) ? bp[(size_t)4 * (st + q) * ld] : 0.0;
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) c = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[q], b4[q], c, 0, 0, 0);
    }
    for (; st < ns; ++st) {
      const bool rok = 4 * st + l4 < m;
      const double a1 = (rok && ci < ma) ? ap[(size_t)4 * st * ld] : 0.0, b1 = (rok && cj < ma) ? bp[(size_t)4 * st * ld] : 0.0;
      c = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, c, 0, 0, 0);
    }
    // C layout: entry (i = 16 it + 4 q + l4, j = 16 jt + l15); the mirrored entry of an off-diagonal tile gets the same value
    const bool isS0 = t == J;
    double *out = isS0 ? A.s0 + so : rec + s_aoff[t];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int i = 16 * it + 4 * q + l4, j = cj;
      if (i < ma && j < ma) {
        double v = c[q];
        if (!isS0) {
          for (int ch = 0; ch < nch; ++ch) {
            const long long co = ch < 16 ? s_choff[ch] : A.blks[A.dch_idx[B.dch_ptr + ch]].acc_off;
            v += A.acc[co + s_aoff[t] + (size_t)i * ma + j];
          }
        }
        out[(size_t)i * ma + j] = v;
        if (it != jt) out[(size_t)j * ma + i] = v;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Phase B, fast path: one workgroup per column group (same groups as k_factor_mfma).  A sibling group of
// non-reference blocks is treated as ONE block with a diagonal Ri: the Gram of the stacked panel rows is the sum
// of the siblings' messages, so the group writes one message record (at its first block) instead of one per block.
// Gram matrices N_a' N_a run on the FP64 matrix cores; the m x m posterior Cholesky and both triangular solves
// run in the registers of wave 0.
// ---------------------------------------------------------------------------------------------------------------
struct SampleFastArgs {
  const Blk *blks;
  const int *anc_idx, *dch_idx;
  const Grp *grps;
  int ngrp;
  const double *panels;
  double *w;
  const double *y, *xb, *z;
  const int *mv;
  double *acc;
  int *errflag;
  const long long *gdesc;   // group descriptors of this launch's first group onwards
  int gd_stride;
  int ldN, Mr4, Mrows, maxP, av_dbl;   // Mrows: staged panel rows (the level's largest group)   // av_dbl: doubles of the per-ancestor vectors / elimination scratch (>= 32 J, >= 224)
  int do_gram;   // 0: the Gram parts of the records are still valid for this theta, rewrite only the vectors
  int no_fwd;    // limited_tree: nothing is forwarded from the children's records
  double tausq_inv[QMAX];
};

__global__ __launch_bounds__(NT, 3) void k_sample_mfma(SampleFastArgs A) {
  extern __shared__ double lds[];
  __shared__ int s_am[MAXJ], s_ao[MAXJ + 1];
  __shared__ long long s_arow[MAXJ], s_aoff[MAXJ + 1];
  __shared__ long long s_bpan[32], s_brow[32];
  __shared__ int s_bld[32];
  __shared__ int s_fail;
  __shared__ long long s_coff[64];               // message records of the direct children
  __shared__ int s_nch;
  __shared__ long long s_gd[GD_MAXW];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  const int ldN = A.ldN, Mr4 = A.Mr4;
  double *Np = lds;                              // maxM x ldN (no pad rows: the Gram tiles mask rows >= M)
  double *wv = Np + (size_t)A.Mrows * ldN + 32;  // maxP + 32 : ancestors' w, then the group's new w
  double *tv = wv + A.maxP + 32, *ev = tv + 32, *bv = ev + 32, *tsq = bv + 32, *yx = tsq + 32, *zc = yx + 32;
  double *av = zc + 32;                          // MAXJ x 32
  int *colblk = (int *)(av + A.av_dbl);          // 32 ints
  double *S = av + A.av_dbl + 16;                // reference levels only: maxM x CH_LD
  double *Li = S;                                // chol(S)^{-1} replaces S (the elimination reads S once, writes at the end)

  int gidx = blockIdx.x;
  {
    const int per = A.ngrp >> 3;
    if (gidx < per * 8) gidx = (gidx & 7) * per + (gidx >> 3);
  }
  STAMP_DECL
  for (int i = tid; i < A.gd_stride; i += NT) s_gd[i] = A.gdesc[(size_t)gidx * A.gd_stride + i];   // the group's descriptor: one round trip
  __syncthreads();
  const GdHead B0 = gd_unpack(s_gd, tid, s_am, s_ao, s_arow, nullptr, s_aoff, s_bpan, s_brow, s_bld, s_coff);
  const Grp G = {B0.row0, B0.blk0, B0.nblk, B0.M, B0.P};
  const int M = G.M, P = G.P, J = B0.nanc;
  const bool refgrp = B0.isref != 0;
  if (tid == 0) { s_fail = 0; s_nch = B0.ndch; }
  if (tid < 32) {
    const int j = tid;
    if (j < M) {
      const long long r = G.row0 + j;
      tsq[j] = A.tausq_inv[A.mv[r]]; yx[j] = A.y[r] - A.xb[r]; zc[j] = A.z[r];
      int bi = 0;
      while (bi + 1 < G.nblk && r >= s_gd[8 + 4 * J + 3 * (bi + 1) + 1]) ++bi;
      colblk[j] = bi;
    } else {
      tsq[j] = 0.0; yx[j] = 0.0; zc[j] = 0.0; colblk[j] = 0;
    }
  }
  __syncthreads();
  for (int k = tid; k < P; k += NT) {
    int t = 0;
    while (t + 1 < J && k >= s_ao[t + 1]) ++t;
    wv[k] = A.w[s_arow[t] + (k - s_ao[t])];
  }
  STAMP(0);
  // panel rows -> LDS (row j of the group = one panel row of its block); pad rows / columns zero.
  // Each wave takes rows wid, wid+4, ...; all loads of four rows are issued before the first LDS store.
  const int rowlen = P + (refgrp ? M : 1);
  for (int jb = 0; jb < Mr4; jb += 16) {
    double tmp[4][4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int j = jb + wid + 4 * rr;
      const int jc = min(j, M - 1);
      const int bi = colblk[jc];
      const double *src = A.panels + s_bpan[bi] + (size_t)(G.row0 + jc - s_brow[bi]) * s_bld[bi];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int k = lane + 64 * c;
        tmp[rr][c] = (j < M && k < rowlen) ? src[k] : 0.0;
      }
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int j = jb + wid + 4 * rr;
      if (j < M) {
        double *dst = Np + (size_t)j * ldN;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int k = lane + 64 * c;
          if (k < ldN) dst[k] = tmp[rr][c];
        }
        for (int k = 256 + lane; k < ldN; k += 64) dst[k] = 0.0;
      }
    }
  }
  for (int j = wid; j < M; j += NT / 64) {          // rows longer than 256 columns (P + M > 256)
    const int bi = colblk[j];
    const double *src = A.panels + s_bpan[bi] + (size_t)(G.row0 + j - s_brow[bi]) * s_bld[bi];
    for (int k = 256 + lane; k < rowlen; k += 64) Np[(size_t)j * ldN + k] = src[k];
  }
  __syncthreads();
  STAMP(1);
  for (int j = wid; j < M; j += NT / 64) {
    double a = 0.0;
    const double *row = Np + (size_t)j * ldN;
    for (int k = lane; k < P; k += 64) a += row[k] * wv[k];
    a = wave_sum(a);
    if (lane == 0) tv[j] = a;
  }
  __syncthreads();
  STAMP(2);
  if (refgrp) {
    const double *Ri = Np + P;   // Ri[i][j] = Np[i*ldN + P + j]
    for (int idx = tid; idx < M * M; idx += NT) {
      const int i = idx / M, j = idx - i * M;
      double a = 0.0;
      if (j <= i) {
        double ch[4];   // the children's records: four loads in flight, fixed summation order
        for (int c0 = 0; c0 < s_nch; c0 += 4) {
#pragma unroll
          for (int cc = 0; cc < 4; ++cc) ch[cc] = (c0 + cc < s_nch) ? A.acc[s_coff[min(c0 + cc, s_nch - 1)] + B0.acc_len + idx] : 0.0;
          if (c0 == 0) for (int k = i; k < M; ++k) a += Ri[(size_t)k * ldN + i] * Ri[(size_t)k * ldN + j];
#pragma unroll
          for (int cc = 0; cc < 4; ++cc) a += ch[cc];
        }
        if (s_nch == 0) for (int k = i; k < M; ++k) a += Ri[(size_t)k * ldN + i] * Ri[(size_t)k * ldN + j];
        if (i == j) a += tsq[i];
      }
      S[i * CH_LD + j] = a;
    }
    if (tid < M) {
      const int i = tid;
      double a = 0.0;
      for (int k = i; k < M; ++k) a -= Ri[(size_t)k * ldN + i] * tv[k];
      double ch[4];
      for (int c0 = 0; c0 < s_nch; c0 += 4) {
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) ch[cc] = (c0 + cc < s_nch) ? A.acc[s_coff[min(c0 + cc, s_nch - 1)] + B0.acc_len + M * M + i] : 0.0;
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) a += ch[cc];
      }
      bv[i] = a + tsq[i] * yx[i];
    }
  }
  STAMP(3);
  // Gram part of the message records, [ N_a' N_a ] + the children's records (spamtree_model.cpp:1158-1207); it does not
  // depend on the draw and is skipped while it is still valid for the accepted theta
  double *rec = A.acc + B0.acc_off;
  const int nsteps = Mr4 >> 2;
  if (A.do_gram) {
    for (int u = wid; u < J * 4; u += NT / 64) {
      const int t = u >> 2, it = (u >> 1) & 1, jt = u & 1;
      const int ma = s_am[t], oa = s_ao[t];
      if (it * 16 >= ma || jt * 16 >= ma) continue;
      d4 c = (d4){0.0, 0.0, 0.0, 0.0};
      const double *ap = Np + (size_t)l4 * ldN + oa + it * 16 + l15;
      const double *bp = Np + (size_t)l4 * ldN + oa + jt * 16 + l15;
      for (int st = 0; st < nsteps; ++st) {
        const bool rok = 4 * st + l4 < M;   // rows >= M are not staged; columns past the ancestor's m only feed discarded entries
        const double av_ = rok ? ap[0] : 0.0, bv_ = rok ? bp[0] : 0.0;
        c = __builtin_amdgcn_mfma_f64_16x16x4f64(av_, bv_, c, 0, 0, 0);
        ap += 4 * ldN; bp += 4 * ldN;
      }
      double *out = rec + s_aoff[t];
      // children's records: all loads of a chunk of four children are issued together (fixed summation order)
      double chv[4] = {0.0, 0.0, 0.0, 0.0};
      const int nfw = A.no_fwd ? 0 : s_nch;
      for (int c0 = 0; c0 < nfw; c0 += 4) {
        double ld4[4][4];
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = it * 16 + l4 + 4 * r, j = jt * 16 + l15;
            ld4[cc][r] = (c0 + cc < nfw && i < ma && j < ma) ? A.acc[s_coff[min(c0 + cc, nfw - 1)] + s_aoff[t] + i * ma + j] : 0.0;
          }
        }
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
#pragma unroll
          for (int r = 0; r < 4; ++r) chv[r] += ld4[cc][r];
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = it * 16 + l4 + 4 * r, j = jt * 16 + l15;
        if (i < ma && j < ma) out[i * ma + j] = c[r] + chv[r];
      }
    }
  }
  if (refgrp) {
    // w_u = L^{-T} (L^{-1} Smu + z) with Li = L^{-1} from the elimination of [S | I]: two small matrix-vector products
    // instead of a forward and a backward substitution (m barrier steps each)
    if (M <= 27) {
      // one wave, registers only (wave_chol_solve): the other waves wait
      __syncthreads();   // S, bv complete
      if (tid < 64) wave_chol_solve<27>(S, bv, zc, wv + P, M, &s_fail, tid);
    } else {
      team_chol_eliminate<5, NT>(S, Li, M, M, av, &s_fail, tid);
      if (tid < M) {
        double a = zc[tid];
        for (int j = 0; j <= tid; ++j) a += Li[tid * CH_LD + j] * bv[j];
        ev[tid] = a;
      }
      __syncthreads();
      if (tid < M) {
        double a = 0.0;
        for (int i = tid; i < M; ++i) a += Li[i * CH_LD + tid] * ev[i];
        wv[P + tid] = a;
      }
    }
  }
  __syncthreads();
  STAMP(4);
  if (refgrp) {
    const double *Ri = Np + P;
    if (tid < M) {
      const int i = tid;
      A.w[G.row0 + i] = wv[P + i];
      double a = tv[i];
      for (int j = 0; j <= i; ++j) a += Ri[(size_t)i * ldN + j] * wv[P + j];
      ev[i] = a;
    }
  } else {
    if (tid < M) {
      const int j = tid;
      const double rj = Np[(size_t)j * ldN + P];
      const double sig = rj * rj + tsq[j];
      if (!(sig > 0.0)) s_fail = 1;
      const double mu = -rj * tv[j] + tsq[j] * yx[j];
      const double c = 1.0 / sqrt(sig);
      const double wj = c * c * mu + c * zc[j];
      wv[P + j] = wj;
      A.w[G.row0 + j] = wj;
      ev[j] = rj * wj + tv[j];
    }
  }
  __syncthreads();
  // av[t][r] = ev[r] - sum_j N[r][oa_t + j] w_a[j]  for every ancestor t
  for (int idx = tid; idx < J * 32; idx += NT) {
    const int t = idx >> 5, r = idx & 31;
    double a = 0.0;
    if (r < M) {
      a = ev[r];
      const double *row = Np + (size_t)r * ldN + s_ao[t];
      const double *wa = wv + s_ao[t];
      for (int j = 0; j < s_am[t]; ++j) a -= row[j] * wa[j];
    }
    av[idx] = a;
  }
  __syncthreads();
  STAMP(5);
  // vector part of the records: -N_a' av_a + the children's
  for (int idx = tid; idx < J * 32; idx += NT) {
    const int t = idx >> 5, i = idx & 31;
    const int ma = s_am[t], oa = s_ao[t];
    if (i < ma) {
      double a = 0.0;
      double ch[4];   // the children's vectors: requested before the dot product, added after it in a fixed order
      const int nch = A.no_fwd ? 0 : s_nch;
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) ch[cc] = (cc < nch) ? A.acc[s_coff[min(cc, max(nch - 1, 0))] + s_aoff[t] + ma * ma + i] : 0.0;
      for (int r = 0; r < M; ++r) a -= Np[(size_t)r * ldN + oa + i] * av[t * 32 + r];
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) a += ch[cc];
      for (int c0 = 4; c0 < nch; c0 += 4) {
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) ch[cc] = (c0 + cc < nch) ? A.acc[s_coff[min(c0 + cc, nch - 1)] + s_aoff[t] + ma * ma + i] : 0.0;
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) a += ch[cc];
      }
      rec[s_aoff[t] + ma * ma + i] = a;
    }
  }
  STAMP(6);
  STAMP_FLUSH;
  if (tid == 0 && s_fail) atomicMin(A.errflag, B0.level * 16 + (refgrp ? 10 : 11));
}

// ---------------------------------------------------------------------------------------------------------------
// Phase B, the theta-only part of the messages on its own: Gram part of a group's records, N_a' N_a + the children's
// (spamtree_model.cpp:1162, 1190-1192: G_u[a, a]; SURVEY.md Q4: it depends on the accepted theta only).  Launched per level,
// leaves first, on the first sweep after a factorisation of the accepted slot; the sweep itself then always takes the lean
// kernels.  The panel is NOT staged: a wave owns one 16 x 16 tile of one ancestor's Gram matrix and reads its MFMA operands
// straight from global memory / L2 (16 consecutive doubles per panel row and lane group), every load of the tile in
// flight before the first MFMA; LDS holds the descriptor only, so eight workgroups share a CU.  Same arithmetic and
// summation order as the Gram section of k_sample_mfma (bit-identical records).
__global__ __launch_bounds__(NT, 6) void k_gram(SampleFastArgs A) {
  __shared__ int s_am[MAXJ], s_ao[MAXJ + 1];
  __shared__ long long s_arow[MAXJ], s_aoff[MAXJ + 1];
  __shared__ long long s_bpan[32], s_brow[32];
  __shared__ int s_bld[32];
  __shared__ long long s_coff[64];
  __shared__ long long s_gd[GD_MAXW];
  __shared__ long long s_rowoff[32];   // panel offset of the group's row r

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int l15 = lane & 15, l4 = lane >> 4;
  int gidx = blockIdx.x;
  {
    const int per = A.ngrp >> 3;
    if (gidx < per * 8) gidx = (gidx & 7) * per + (gidx >> 3);
  }
  for (int i = tid; i < A.gd_stride; i += NT) s_gd[i] = A.gdesc[(size_t)gidx * A.gd_stride + i];
  __syncthreads();
  const GdHead B0 = gd_unpack(s_gd, tid, s_am, s_ao, s_arow, nullptr, s_aoff, s_bpan, s_brow, s_bld, s_coff);
  const int M = B0.M, J = B0.nanc;
  __syncthreads();
  if (tid < 32) {
    long long off = 0;
    if (tid < M) {
      const long long r = B0.row0 + tid;
      int bi = 0;
      while (bi + 1 < B0.nblk && r >= s_brow[bi + 1]) ++bi;
      off = s_bpan[bi] + (r - s_brow[bi]) * s_bld[bi];
    }
    s_rowoff[tid] = off;
  }
  __syncthreads();
  const int nsteps = (M + 3) >> 2;          // <= 8
  const int nfw = A.no_fwd ? 0 : B0.ndch;
  double *rec = A.acc + B0.acc_off;
  for (int u = wid; u < J * 4; u += NT / 64) {
    const int t = u >> 2, it = (u >> 1) & 1, jt = u & 1;
    const int ma = s_am[t], oa = s_ao[t];
    if (it * 16 >= ma || jt * 16 >= ma) continue;
    double av_[8], bv_[8];
#pragma unroll
    for (int st = 0; st < 8; ++st) {
      const int r = 4 * st + l4;
      const bool rok = st < nsteps && r < M;   // columns past the ancestor's m only feed discarded entries
      const double *row = A.panels + s_rowoff[min(r, 31)] + oa + l15;
      av_[st] = rok ? row[it * 16] : 0.0;
      bv_[st] = rok ? row[jt * 16] : 0.0;
    }
    d4 c = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int st = 0; st < 8; ++st)
      if (st < nsteps) c = __builtin_amdgcn_mfma_f64_16x16x4f64(av_[st], bv_[st], c, 0, 0, 0);
    double *out = rec + s_aoff[t];
    double chv[4] = {0.0, 0.0, 0.0, 0.0};
    for (int c0 = 0; c0 < nfw; c0 += 4) {   // children's records: chunks of four in flight, fixed summation order
      double ld4[4][4];
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = it * 16 + l4 + 4 * r, j = jt * 16 + l15;
          ld4[cc][r] = (c0 + cc < nfw && i < ma && j < ma) ? A.acc[s_coff[min(c0 + cc, nfw - 1)] + s_aoff[t] + i * ma + j] : 0.0;
        }
      }
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) {
#pragma unroll
        for (int r = 0; r < 4; ++r) chv[r] += ld4[cc][r];
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = it * 16 + l4 + 4 * r, j = jt * 16 + l15;
      if (i < ma && j < ma) out[i * ma + j] = c[r] + chv[r];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Phase B, fast path, sweeps that keep the cached Gram parts (do_gram == 0: every sweep between two accepted theta).
// Same results as k_sample_mfma up to rounding, but the panel is never staged in LDS: it is read twice from global
// memory (L2 / Infinity Cache the second time) with a thread mapping chosen per pass --
//   pass 1: thread (row r, ancestor t) sums its 25-or-so products N[r][oa_t + j] w_a[j]: the segment sums give both
//           tv = N w_pa (their sum over t) and, later, av_t = ev - N_t w_t, with no cross-lane reduction;
//   pass 2: thread k (a chain column) accumulates -sum_r N[r][k] av_t(k)[r]: coalesced rows, no reduction either.
// LDS holds vectors only (plus the m x m posterior precision of reference blocks): ~6-19 KB instead of 50-55 KB, so
// 6-8 workgroups share a CU and their latency chains overlap.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(NT, 5) void k_sample_lean(SampleFastArgs A) {
  extern __shared__ double lds[];
  __shared__ int s_am[MAXJ], s_ao[MAXJ + 1];
  __shared__ long long s_arow[MAXJ], s_aoff[MAXJ + 1];
  __shared__ long long s_bpan[32], s_brow[32], s_rowoff[32];
  __shared__ int s_bld[32], s_cb[32];
  __shared__ int s_fail;
  __shared__ long long s_coff[64];
  __shared__ int s_nch;
  __shared__ long long s_gd[GD_MAXW];

  const int tid = threadIdx.x;
  double *wv = lds;                                  // maxP + 32 : ancestors' w, then the group's new w
  double *seg = wv + A.maxP + 32;                    // av_dbl : seg[t][r], later av[t][r]; elimination scratch in between
  double *tv = seg + A.av_dbl + 16, *ev = tv + 32, *bv = ev + 32, *tsq = bv + 32, *yx = tsq + 32, *zc = yx + 32, *rjv = zc + 32;
  double *Rc = rjv + 32;                             // reference levels only: Ri, 32 x CH_LD
  double *S = Rc + 32 * CH_LD;                       // 32 x CH_LD: posterior precision, then its inverse Cholesky factor

  int gidx = blockIdx.x;
  {
    const int per = A.ngrp >> 3;
    if (gidx < per * 8) gidx = (gidx & 7) * per + (gidx >> 3);
  }
  for (int i = tid; i < A.gd_stride; i += NT) s_gd[i] = A.gdesc[(size_t)gidx * A.gd_stride + i];   // the group's descriptor: one round trip
  __syncthreads();
  const GdHead B0 = gd_unpack(s_gd, tid, s_am, s_ao, s_arow, nullptr, s_aoff, s_bpan, s_brow, s_bld, s_coff);
  const Grp G = {B0.row0, B0.blk0, B0.nblk, B0.M, B0.P};
  const int M = G.M, P = G.P, J = B0.nanc;
  const bool refgrp = B0.isref != 0;
  if (tid == 0) { s_fail = 0; s_nch = B0.ndch; }
  if (tid >= 32 && tid < 64) {
    const int j = tid - 32;
    double t_ = 0.0, y_ = 0.0, z_ = 0.0, r_ = 0.0;
    int bi = 0;
    long long ro = 0;
    if (j < M) {
      const long long r = G.row0 + j;
      t_ = A.tausq_inv[A.mv[r]]; y_ = A.y[r] - A.xb[r]; z_ = A.z[r];
      const long long *gb = s_gd + 8 + 4 * J;   // per block: panel offset, first row, ld
      while (bi + 1 < G.nblk && r >= gb[3 * (bi + 1) + 1]) ++bi;
      ro = gb[3 * bi] + (r - gb[3 * bi + 1]) * gb[3 * bi + 2];
      if (!refgrp) r_ = A.panels[ro + P];
    }
    tsq[j] = t_; yx[j] = y_; zc[j] = z_; rjv[j] = r_; s_cb[j] = bi; s_rowoff[j] = ro;
  }
  __syncthreads();
  for (int k = tid; k < P; k += NT) {
    int t = 0;
    while (t + 1 < J && k >= s_ao[t + 1]) ++t;
    wv[k] = A.w[s_arow[t] + (k - s_ao[t])];
  }
  if (refgrp) {   // Ri -> LDS (rows of the panel's last M columns)
    for (int idx = tid; idx < M * M; idx += NT) {
      const int i = idx / M, j = idx - i * M;
      Rc[i * CH_LD + j] = (j <= i) ? A.panels[s_rowoff[i] + P + j] : 0.0;
    }
  }
  __syncthreads();
  // ---- pass 1: segment sums seg[t][r] = sum_j N[r][oa_t + j] w_a[j]
  for (int idx = tid; idx < M * J; idx += NT) {
    const int r = idx / J, t = idx - r * J;
    const int ma = s_am[t], oa = s_ao[t];
    const double *row = A.panels + s_rowoff[r] + oa;
    const double *wa = wv + oa;
    double a = 0.0;
    for (int j0 = 0; j0 < ma; j0 += 16) {   // two batches of loads for the usual 25-row ancestor
      double x[16];
#pragma unroll
      for (int jj = 0; jj < 16; ++jj) x[jj] = (j0 + jj < ma) ? row[j0 + jj] : 0.0;
#pragma unroll
      for (int jj = 0; jj < 16; ++jj) a += x[jj] * ((j0 + jj < ma) ? wa[j0 + jj] : 0.0);
    }
    seg[t * 32 + r] = a;
  }
  __syncthreads();
  if (tid < M) {
    double a = 0.0;
    for (int t = 0; t < J; ++t) a += seg[t * 32 + tid];
    tv[tid] = a;
  }
  __syncthreads();
  if (refgrp) {
    for (int idx = tid; idx < M * M; idx += NT) {
      const int i = idx / M, j = idx - i * M;
      double a = 0.0;
      if (j <= i) {
        double ch[4];   // the children's records: four loads in flight, fixed summation order
        for (int c0 = 0; c0 < s_nch; c0 += 4) {
#pragma unroll
          for (int cc = 0; cc < 4; ++cc) ch[cc] = (c0 + cc < s_nch) ? A.acc[s_coff[min(c0 + cc, s_nch - 1)] + B0.acc_len + idx] : 0.0;
          if (c0 == 0) for (int k = i; k < M; ++k) a += Rc[k * CH_LD + i] * Rc[k * CH_LD + j];
#pragma unroll
          for (int cc = 0; cc < 4; ++cc) a += ch[cc];
        }
        if (s_nch == 0) for (int k = i; k < M; ++k) a += Rc[k * CH_LD + i] * Rc[k * CH_LD + j];
        if (i == j) a += tsq[i];
      }
      S[i * CH_LD + j] = a;
    }
    if (tid < M) {
      const int i = tid;
      double a = 0.0;
      for (int k = i; k < M; ++k) a -= Rc[k * CH_LD + i] * tv[k];
      double ch[4];
      for (int c0 = 0; c0 < s_nch; c0 += 4) {
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) ch[cc] = (c0 + cc < s_nch) ? A.acc[s_coff[min(c0 + cc, s_nch - 1)] + B0.acc_len + M * M + i] : 0.0;
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) a += ch[cc];
      }
      bv[i] = a + tsq[i] * yx[i];
    }
    // w_u = L^{-T} (L^{-1} Smu + z) with Li = L^{-1} from the elimination of [S | I] (scratch: the pivot cells live after
    // the segment sums, which stay intact)
    double *pub = seg + 32 * J;
    if (M <= 27) {
      // one wave, registers only: elimination with the right-hand side riding along, backward substitution by wave sums
      __syncthreads();   // S, bv complete
      if (tid < 64) wave_chol_solve<27>(S, bv, zc, wv + P, M, &s_fail, tid);
      __syncthreads();
    } else {
      team_chol_eliminate<5, NT>(S, S, M, M, pub, &s_fail, tid);
      if (tid < M) {
        double a = zc[tid];
        for (int j = 0; j <= tid; ++j) a += S[tid * CH_LD + j] * bv[j];
        ev[tid] = a;
      }
      __syncthreads();
      if (tid < M) {
        double a = 0.0;
        for (int i = tid; i < M; ++i) a += S[i * CH_LD + tid] * ev[i];
        wv[P + tid] = a;
      }
      __syncthreads();
    }
    if (tid < M) {
      const int i = tid;
      A.w[G.row0 + i] = wv[P + i];
      double a = tv[i];
      for (int j = 0; j <= i; ++j) a += Rc[i * CH_LD + j] * wv[P + j];
      ev[i] = a;
    }
  } else {
    if (tid < M) {
      const int j = tid;
      const double rj = rjv[j];
      const double sig = rj * rj + tsq[j];
      if (!(sig > 0.0)) s_fail = 1;
      const double mu = -rj * tv[j] + tsq[j] * yx[j];
      const double c = 1.0 / sqrt(sig);
      const double wj = c * c * mu + c * zc[j];
      A.w[G.row0 + j] = wj;
      ev[j] = rj * wj + tv[j];
    }
  }
  __syncthreads();
  // av[t][r] = ev[r] - seg[t][r]
  for (int idx = tid; idx < J * 32; idx += NT) {
    const int r = idx & 31;
    seg[idx] = (r < M) ? ev[r] - seg[idx] : 0.0;
  }
  __syncthreads();
  // ---- pass 2: vector part of the records, -N_a' av_a + the children's
  double *rec = A.acc + B0.acc_off;
  for (int k = tid; k < P; k += NT) {
    int t = 0;
    while (t + 1 < J && k >= s_ao[t + 1]) ++t;
    const int ma = s_am[t], i = k - s_ao[t];
    const double *avt = seg + t * 32;
    double ch[4];   // the children's vectors: requested first, added after the dot product in a fixed order
    const int nch = A.no_fwd ? 0 : s_nch;
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) ch[cc] = (cc < nch) ? A.acc[s_coff[min(cc, max(nch - 1, 0))] + s_aoff[t] + ma * ma + i] : 0.0;
    double a = 0.0;
    for (int r0 = 0; r0 < M; r0 += 8) {
      double x[8];
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) x[rr] = (r0 + rr < M) ? A.panels[s_rowoff[min(r0 + rr, M - 1)] + k] : 0.0;
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) a -= x[rr] * avt[min(r0 + rr, 31)];
    }
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) a += ch[cc];
    for (int c0 = 4; c0 < nch; c0 += 4) {
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) ch[cc] = (c0 + cc < nch) ? A.acc[s_coff[min(c0 + cc, nch - 1)] + s_aoff[t] + ma * ma + i] : 0.0;
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) a += ch[cc];
    }
    rec[s_aoff[t] + ma * ma + i] = a;
  }
  if (tid == 0 && s_fail) atomicMin(A.errflag, B0.level * 16 + (refgrp ? 10 : 11));
}

// ---------------------------------------------------------------------------------------------------------------
// Phase B, reference blocks of at most 27 rows, ONE BLOCK PER WAVE (four independent blocks per workgroup, no workgroup
// barrier anywhere).  k_sample_lean gives a block 256 threads and eleven barriers for what is a chain of short dependent
// steps (descriptor -> rows' data -> panel pass 1 -> children's records -> 25-pivot solve -> panel pass 2): a 55 us latency
// chain per block with five of them in flight per CU.  Here a wave walks the same chain alone -- lane i owns row i: its
// segment sums, row i of the posterior precision built straight into the registers the elimination works on
// (wave_chol_solve_core), its draw -- with 11 KB of LDS, so twelve blocks are in flight per CU.  Same arithmetic and
// summation orders as k_sample_lean (identical draws).  LDS operations of one wave execute in order: a wave-level
// s_waitcnt separates the phases.
// ---------------------------------------------------------------------------------------------------------------
#define WSYNC() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
__global__ __launch_bounds__(NT, 3) void k_sample_wave(SampleFastArgs A) {
  extern __shared__ double lds[];
  __shared__ int s_failw[NT / 64];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int gidx = blockIdx.x * (NT / 64) + wid;
  if (gidx >= A.ngrp) return;   // no workgroup barrier below: a wave without a block simply leaves
  double *base = lds + (size_t)wid * A.ldN;          // this wave's LDS region (A.ldN doubles)
  long long *s_gd = (long long *)base;               // the block's descriptor
  double *wv = base + A.gd_stride;                   // maxP + 32 : ancestors' w, then the block's new w
  double *seg = wv + A.maxP + 32;                    // av_dbl : seg[t][r], later av[t][r]
  double *tv = seg + A.av_dbl, *ev = tv + 32;
  double *Rc = ev + 32;                              // Mrows x CH_LD: Ri
  if (lane == 0) s_failw[wid] = 0;
  for (int i = lane; i < A.gd_stride; i += 64) s_gd[i] = A.gdesc[(size_t)gidx * A.gd_stride + i];
  WSYNC();
  auto slo = [](long long v) { return __builtin_amdgcn_readfirstlane((int)(v & 0xffffffffLL)); };
  auto shi = [](long long v) { return __builtin_amdgcn_readfirstlane((int)(v >> 32)); };
  auto sll = [&](long long v) { return ((long long)shi(v) << 32) | (unsigned int)slo(v); };
  const long long row0 = sll(s_gd[0]), acc_off = sll(s_gd[1]);
  const int M = slo(s_gd[2]), P = shi(s_gd[2]), J = slo(s_gd[3]), level = shi(s_gd[4]);
  const int nch = slo(s_gd[5]), acc_len = shi(s_gd[5]);
  const long long *gb = s_gd + 8 + 4 * J;            // the block: panel offset, first row, ld
  const long long bpan = sll(gb[0]);
  const int bld = (int)sll(gb[2]);
  const long long *coff = gb + 3;                    // message records of the direct children
  auto am_of = [&](int t) { return (int)(s_gd[8 + 4 * t] & 0xffffffffLL); };
  auto ao_of = [&](int t) { return (int)(s_gd[8 + 4 * t] >> 32); };
  const bool row = lane < M;
  const int li = min(lane, 31);
  double tsq = 0.0, yx = 0.0, zc = 0.0;
  if (row) { const long long r = row0 + lane; tsq = A.tausq_inv[A.mv[r]]; yx = A.y[r] - A.xb[r]; zc = A.z[r]; }
  for (int k = lane; k < P; k += 64) {
    int t = 0;
    for (int j = 1; j < J; ++j) t += k >= ao_of(j) ? 1 : 0;   // independent compares, not a search loop of dependent LDS reads
    wv[k] = A.w[s_gd[8 + 4 * t + 1] + (k - ao_of(t))];
  }
  for (int idx = lane; idx < M * M; idx += 64) {     // Ri -> LDS (the panel's last M columns)
    const int i = idx / M, j = idx - i * M;
    Rc[i * CH_LD + j] = (j <= i) ? A.panels[bpan + (size_t)i * bld + P + j] : 0.0;
  }
  WSYNC();
  // ---- pass 1: segment sums seg[t][r] = sum_j N[r][oa_t + j] w_a[j]
  for (int idx = lane; idx < M * J; idx += 64) {
    const int r = idx / J, t = idx - r * J;
    const int ma = am_of(t), oa = ao_of(t);
    const double *prow = A.panels + bpan + (size_t)r * bld + oa;
    const double *wa = wv + oa;
    double a = 0.0;
    for (int j0 = 0; j0 < ma; j0 += 16) {
      double x[16];
#pragma unroll
      for (int jj = 0; jj < 16; ++jj) x[jj] = (j0 + jj < ma) ? prow[j0 + jj] : 0.0;
#pragma unroll
      for (int jj = 0; jj < 16; ++jj) a += x[jj] * ((j0 + jj < ma) ? wa[j0 + jj] : 0.0);
    }
    seg[t * 32 + r] = a;
  }
  WSYNC();
  double tvi = 0.0;
  if (row) { for (int t = 0; t < J; ++t) tvi += seg[t * 32 + lane]; tv[lane] = tvi; }
  WSYNC();
  // ---- row `lane` of the posterior precision Ri'Ri + the children's Gram parts + tausq_inv, straight into registers
  double a[27];
#pragma unroll
  for (int j = 0; j < 27; ++j) a[j] = 0.0;
  double bv = 0.0;
  for (int k = 0; k < M; ++k) {                      // Ri[k][i] = 0 for k < i: the leading terms add exact zeros
    const double rk = Rc[k * CH_LD + li];
    bv -= rk * tv[k];
#pragma unroll
    for (int j = 0; j < 27; ++j) a[j] += rk * Rc[k * CH_LD + j];
  }
  for (int c = 0; c < nch; ++c) {                    // the children's records, fixed order
    const double *rc = A.acc + coff[c] + acc_len;
    double ch[27];
#pragma unroll
    for (int j = 0; j < 27; ++j) ch[j] = (row && j <= lane) ? rc[li * M + j] : 0.0;
    const double cv = row ? rc[M * M + li] : 0.0;
#pragma unroll
    for (int j = 0; j < 27; ++j) a[j] += ch[j];
    bv += cv;
  }
#pragma unroll
  for (int j = 0; j < 27; ++j) {
    if (j == lane) a[j] += tsq;
    a[j] = (row && j <= lane) ? a[j] : (j == lane ? 1.0 : 0.0);
  }
  bv = row ? bv + tsq * yx : 0.0;
  // ---- w_u = L^{-T} (L^{-1} b + z): elimination with the right-hand side riding along, backward substitution by wave sums
  const double wnew = wave_chol_solve_core<27>(a, bv, zc, M, &s_failw[wid], lane);
  if (row) { A.w[row0 + lane] = wnew; wv[P + lane] = wnew; }
  WSYNC();
  if (row) {
    double e = tvi;
    for (int j = 0; j <= lane; ++j) e += Rc[lane * CH_LD + j] * wv[P + j];
    ev[lane] = e;
  }
  WSYNC();
  for (int idx = lane; idx < J * 32; idx += 64) {    // av[t][r] = ev[r] - seg[t][r]
    const int r = idx & 31;
    seg[idx] = (r < M) ? ev[r] - seg[idx] : 0.0;
  }
  WSYNC();
  // ---- pass 2: vector part of the records, -N_a' av_a + the children's
  double *rec = A.acc + acc_off;
  const int nfw = A.no_fwd ? 0 : nch;
  for (int k = lane; k < P; k += 64) {
    int t = 0;
    for (int j = 1; j < J; ++j) t += k >= ao_of(j) ? 1 : 0;   // independent compares, not a search loop of dependent LDS reads
    const int ma = am_of(t), i = k - ao_of(t);
    const long long aoff = s_gd[8 + 4 * t + 3];
    const double *avt = seg + t * 32;
    double ch[4];
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) ch[cc] = (cc < nfw) ? A.acc[coff[min(cc, max(nfw - 1, 0))] + aoff + ma * ma + i] : 0.0;
    double acc = 0.0;
    for (int r0 = 0; r0 < M; r0 += 8) {
      double x[8];
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) x[rr] = (r0 + rr < M) ? A.panels[bpan + (size_t)min(r0 + rr, M - 1) * bld + k] : 0.0;
#pragma unroll
      for (int rr = 0; rr < 8; ++rr) acc -= x[rr] * avt[min(r0 + rr, 31)];
    }
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) acc += ch[cc];
    for (int c0 = 4; c0 < nfw; c0 += 4) {
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) ch[cc] = (c0 + cc < nfw) ? A.acc[coff[min(c0 + cc, nfw - 1)] + aoff + ma * ma + i] : 0.0;
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) acc += ch[cc];
    }
    rec[aoff + ma * ma + i] = acc;
  }
  WSYNC();
  if (lane == 0 && s_failw[wid]) atomicMin(A.errflag, level * 16 + 10);
}

// ---------------------------------------------------------------------------------------------------------------
// Phase B, fast path, leaf (non-reference) groups on sweeps that keep the cached Gram parts.  The rows of a leaf group
// are independent given the ancestors (diagonal Ri), so a wave owns whole rows: lanes hold the row's columns (coalesced
// loads, registers only), the per-ancestor segment sums come from masked butterfly reductions, the draw, the residual
// and the row's contribution -N[r][k] av_t(k)[r] to every chain column follow without leaving the wave; only the
// column sums over the four waves go through LDS.  One pass over the panel, ~10 KB of LDS, one barrier pair.
// ---------------------------------------------------------------------------------------------------------------
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

__global__ __launch_bounds__(NT, 5) void k_sample_leaf(SampleFastArgs A) {
  extern __shared__ double lds[];
  __shared__ int s_am[MAXJ], s_ao[MAXJ + 1];
  __shared__ long long s_arow[MAXJ], s_aoff[MAXJ + 1];
  __shared__ long long s_bpan[32], s_brow[32], s_rowoff[32];
  __shared__ int s_bld[32];
  __shared__ int s_fail;
  __shared__ long long s_coff[64];
  __shared__ int s_nch;
  __shared__ long long s_gd[GD_MAXW];
  __shared__ double s_seg[NT / 64][MAXJ];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  double *wv = lds;                       // maxP + 32
  double *red = wv + A.maxP + 32;         // 4 x 256: per-wave column sums
  double *tsq = red + 4 * 256, *yx = tsq + 32, *zc = yx + 32;

  int gidx = blockIdx.x;
  {
    const int per = A.ngrp >> 3;
    if (gidx < per * 8) gidx = (gidx & 7) * per + (gidx >> 3);
  }
  for (int i = tid; i < A.gd_stride; i += NT) s_gd[i] = A.gdesc[(size_t)gidx * A.gd_stride + i];   // the group's descriptor: one round trip
  __syncthreads();
  const GdHead B0 = gd_unpack(s_gd, tid, s_am, s_ao, s_arow, nullptr, s_aoff, s_bpan, s_brow, s_bld, s_coff);
  const Grp G = {B0.row0, B0.blk0, B0.nblk, B0.M, B0.P};
  const int M = G.M, P = G.P, J = B0.nanc;
  if (tid == 0) { s_fail = 0; s_nch = B0.ndch; }
  if (tid >= 32 && tid < 64) {
    const int j = tid - 32;
    double t_ = 0.0, y_ = 0.0, z_ = 0.0;
    long long ro = 0;
    if (j < M) {
      const long long r = G.row0 + j;
      t_ = A.tausq_inv[A.mv[r]]; y_ = A.y[r] - A.xb[r]; z_ = A.z[r];
      int bi = 0;
      const long long *gb = s_gd + 8 + 4 * J;   // per block: panel offset, first row, ld
      while (bi + 1 < G.nblk && r >= gb[3 * (bi + 1) + 1]) ++bi;
      ro = gb[3 * bi] + (r - gb[3 * bi + 1]) * gb[3 * bi + 2];
    }
    tsq[j] = t_; yx[j] = y_; zc[j] = z_; s_rowoff[j] = ro;
  }
  __syncthreads();
  // this lane's columns k = lane + 64 c (P + 1 <= 256 columns: the host routes longer chains to k_sample_mfma)
  int tk[4];
  double wk[4], acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int k = lane + 64 * c;
    const int t = anc_of(s_ao, J, k);
    tk[c] = k < P ? t : -1;
    wk[c] = k < P ? A.w[s_arow[t] + (k - s_ao[t])] : 0.0;
    acc[c] = 0.0;
  }
  const int lastc = P >> 6, lastl = P & 63;   // where column P (the row's r_j) lives
#pragma unroll 1
  for (int b = 0; b < 2; ++b) {
    double v[4][4];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int j = wid + 4 * (4 * b + rr);
      const double *src = A.panels + s_rowoff[min(j, M - 1)];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int k = lane + 64 * c;
        v[rr][c] = (j < M && k <= P) ? src[k] : 0.0;
      }
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int j = wid + 4 * (4 * b + rr);
      if (j < M) {   // wave-uniform
        // segment sums (every lane gets them), tv = their sum in ancestor order
        double tvj = 0.0;
        for (int t = 0; t < J; ++t) {
          double x = 0.0;
#pragma unroll
          for (int c = 0; c < 4; ++c) x += (tk[c] == t) ? v[rr][c] * wk[c] : 0.0;
          x = wave_allsum(x);
          if (lane == 0) s_seg[wid][t] = x;
          tvj += x;
        }
        double rj = 0.0;
#pragma unroll
        for (int c = 0; c < 4; ++c) if (c == lastc) rj = __shfl(v[rr][c], lastl, 64);
        const double sig = rj * rj + tsq[j];
        if (!(sig > 0.0) && lane == 0) s_fail = 1;
        const double mu = -rj * tvj + tsq[j] * yx[j];
        const double cc = 1.0 / sqrt(sig);
        const double wj = cc * cc * mu + cc * zc[j];
        if (lane == 0) A.w[G.row0 + j] = wj;
        const double evj = rj * wj + tvj;
        // this row's share of the vector records: -N[j][k] (ev_j - seg_t(k)[j])
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (tk[c] >= 0) acc[c] -= v[rr][c] * (evj - s_seg[wid][tk[c]]);
      }
    }
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) red[wid * 256 + lane + 64 * c] = acc[c];
  __syncthreads();
  double *rec = A.acc + B0.acc_off;
  for (int k = tid; k < P; k += NT) {
    const int t = anc_of(s_ao, J, k);
    const int ma = s_am[t], i = k - s_ao[t];
    double a = ((red[k] + red[256 + k]) + red[512 + k]) + red[768 + k];
    for (int cc = 0; cc < s_nch; ++cc) a += A.acc[s_coff[cc] + s_aoff[t] + ma * ma + i];
    rec[s_aoff[t] + ma * ma + i] = a;
  }
  if (tid == 0 && s_fail) atomicMin(A.errflag, B0.level * 16 + 11);
}

// ---------------------------------------------------------------------------------------------------------------
// Phase C: residual + quadratic form per block (spamtree_model.cpp:781-826)
// ---------------------------------------------------------------------------------------------------------------
struct LoglikArgs {
  const Blk *blks;
  const int *anc_idx;
  const int *list;
  int nlist;
  const double *panels;
  const double *w;
  double *loglik_c;
  int maxP, maxM;
};

__global__ __launch_bounds__(NT) void k_loglik(LoglikArgs A) {
  extern __shared__ double lds[];
  __shared__ int s_am[MAXJ], s_ao[MAXJ + 1];
  __shared__ long long s_arow[MAXJ];
  __shared__ double s_red[NT / 64];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  double *wv = lds;
  double *tv = wv + (A.maxP + A.maxM);
  for (int li = blockIdx.x; li < A.nlist; li += gridDim.x) {
    const int b = A.list[li];
    const Blk B = A.blks[b];
    const int m = B.m, P = B.P, J = B.nanc, ld = B.ld;
    __syncthreads();
    if (tid < J) {
      const int a = A.anc_idx[B.anc_ptr + tid];
      s_am[tid] = A.blks[a].m;
      s_arow[tid] = A.blks[a].row0;
    }
    __syncthreads();
    if (tid == 0) {
      int o = 0;
      for (int t = 0; t < J; ++t) { s_ao[t] = o; o += s_am[t]; }
      s_ao[J] = o;
    }
    __syncthreads();
    for (int t = 0; t < J; ++t)
      for (int i = tid; i < s_am[t]; i += NT) wv[s_ao[t] + i] = A.w[s_arow[t] + i];
    for (int i = tid; i < m; i += NT) wv[P + i] = A.w[B.row0 + i];
    __syncthreads();
    const double *N = A.panels + B.panel_off;
    for (int i = wid; i < m; i += NT / 64) {
      double acc = 0.0;
      for (int k = lane; k < P; k += 64) acc += N[(size_t)i * ld + k] * wv[k];
      acc = wave_sum(acc);
      if (lane == 0) tv[i] = acc;
    }
    __syncthreads();
    double part = 0.0;
    for (int i = tid; i < m; i += NT) {
      double acc = tv[i];
      if (B.isref) {
        for (int j = 0; j <= i; ++j) acc += N[(size_t)i * ld + P + j] * wv[P + j];
      } else {
        acc += N[(size_t)i * ld + P] * wv[P + i];
      }
      part += acc * acc;
    }
    const double wcore = block_sum(part, s_red);
    if (tid == 0) A.loglik_c[b] = (double)m * HL2PI - 0.5 * wcore;
  }
}

// Phase C for the column-group levels: one workgroup per group (a reference block, or <= 32 rows of sibling leaf blocks
// which share their ancestors' w).  Every wave requests all of its panel rows before the first use (up to 8 rows x 5
// pieces of 64 columns in flight per lane); a row of the panel is [ N_i | Ri_i 0 ] (reference) or [ N_i | r_i ] (leaf), so
// the residual e_i = Ri (w_u - H w_pa) is one dot product of the row with [ w_pa ; w_u ].
struct LoglikGrpArgs {
  const Blk *blks;
  const int *anc_idx;
  const Grp *grps;
  const int *list;   // group indices
  int nlist;
  const double *panels;
  const double *w;
  double *loglik_c;
  int maxP;
  const long long *gdesc;   // all group descriptors (indexed by the absolute group index in `list`)
  int gd_stride;
};

__global__ __launch_bounds__(NT, 8) void k_loglik_grp(LoglikGrpArgs A) {
  extern __shared__ double lds[];
  __shared__ int s_am[MAXJ], s_ao[MAXJ + 1];
  __shared__ long long s_arow[MAXJ];
  __shared__ long long s_bpan[32], s_brow[32];
  __shared__ int s_bld[32], s_cb[32];
  __shared__ double s_e2[32];
  __shared__ long long s_gd[GD_MAXW];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  double *wv = lds;   // maxP + 32
  {
    const long long *gd = A.gdesc + (size_t)A.list[blockIdx.x] * A.gd_stride;
    for (int i = tid; i < A.gd_stride; i += NT) s_gd[i] = gd[i];   // the group's descriptor: one round trip
  }
  __syncthreads();
  const GdHead B0 = gd_unpack(s_gd, tid, s_am, s_ao, s_arow, nullptr, nullptr, s_bpan, s_brow, s_bld, nullptr);
  const Grp G = {B0.row0, B0.blk0, B0.nblk, B0.M, B0.P};
  const int M = G.M, P = G.P, J = B0.nanc;
  const bool refgrp = B0.isref != 0;
  if (tid >= 128 && tid < 128 + 32) wv[P + tid - 128] = (tid - 128 < M) ? A.w[G.row0 + tid - 128] : 0.0;
  if (tid >= 32 && tid < 64) {
    const int j = tid - 32;
    int bi = 0;
    if (j < M) { const long long r = G.row0 + j; while (bi + 1 < G.nblk && r >= s_gd[8 + 4 * J + 3 * (bi + 1) + 1]) ++bi; }
    s_cb[j] = bi;
  }
  __syncthreads();
  // this wave's rows wid, wid + 4, ... in two batches of four rows; all of a batch's loads are issued before anything
  // waits.  <= 64 VGPRs: eight workgroups per CU keep the memory system busy.
  const int rowlen = P + (refgrp ? M : 1);
#pragma unroll 1
  for (int b = 0; b < 2; ++b) {
    double v[4][5];
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int j = wid + 4 * (4 * b + rr), jc = min(j, M - 1);
      const int bi = s_cb[jc];
      const double *src = A.panels + s_bpan[bi] + (size_t)(G.row0 + jc - s_brow[bi]) * s_bld[bi];
#pragma unroll
      for (int c = 0; c < 5; ++c) {
        const int k = lane + 64 * c;
        v[rr][c] = (j < M && k < rowlen) ? src[k] : 0.0;
      }
    }
    if (b == 0) {
      for (int k = tid; k < P; k += NT) {
        int t = 0;
        while (t + 1 < J && k >= s_ao[t + 1]) ++t;
        wv[k] = A.w[s_arow[t] + (k - s_ao[t])];
      }
      __syncthreads();
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int j = wid + 4 * (4 * b + rr);
      double acc = 0.0;
#pragma unroll
      for (int c = 0; c < 5; ++c) {
        const int k = lane + 64 * c;
        const double wk = k < P ? wv[k] : (refgrp ? (k < rowlen ? wv[k] : 0.0) : wv[P + min(j, 31)]);
        acc += v[rr][c] * wk;
      }
      acc = wave_sum(acc);
      if (lane == 0 && j < 32) s_e2[j] = acc * acc;
    }
  }
  __syncthreads();
  if (tid < G.nblk) {
    double wc = 0.0;
    int cnt = 0;
    for (int j = 0; j < M; ++j)
      if (s_cb[j] == tid) { wc += s_e2[j]; ++cnt; }
    A.loglik_c[G.blk0 + tid] = (double)cnt * HL2PI - 0.5 * wc;
  }
}

// fixed-shape deterministic sums of two arrays: out[0] = sum a, out[1] = sum b.  Stage 1: SUM2_WG workgroups, each a
// contiguous chunk (thread-strided partial sums, LDS tree); stage 2: one wave adds the SUM2_WG partials in order.
#define SUM2_WG 64
__global__ __launch_bounds__(NT) void k_sum2_partial(const double *a, const double *b, int n, double *partial) {
  __shared__ double sa[NT], sb[NT];
  const int chunk = (n + SUM2_WG - 1) / SUM2_WG;
  const int lo = blockIdx.x * chunk, hi = min(n, lo + chunk);
  double xa = 0.0, xb = 0.0;
  for (int i = lo + threadIdx.x; i < hi; i += NT) { xa += a[i]; xb += b[i]; }
  sa[threadIdx.x] = xa; sb[threadIdx.x] = xb;
  __syncthreads();
  for (int s2 = NT / 2; s2 > 0; s2 >>= 1) {
    if ((int)threadIdx.x < s2) { sa[threadIdx.x] += sa[threadIdx.x + s2]; sb[threadIdx.x] += sb[threadIdx.x + s2]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { partial[2 * blockIdx.x] = sa[0]; partial[2 * blockIdx.x + 1] = sb[0]; }
}
__global__ void k_sum2_final(const double *partial, double *out) {
  if (threadIdx.x < 2) {
    double s2 = 0.0;
    for (int g = 0; g < SUM2_WG; ++g) s2 += partial[2 * g + threadIdx.x];
    out[threadIdx.x] = s2;
  }
}
static void launch_sum2(hipStream_t st, const double *a, const double *b, int n, double *partial, double *out) {
  hipLaunchKernelGGL(k_sum2_partial, dim3(SUM2_WG), dim3(NT), 0, st, a, b, n, partial);
  hipLaunchKernelGGL(k_sum2_final, dim3(1), dim3(64), 0, st, partial, out);
}

// XB = X * Bcoeff[:, mv]   (spamtree_model.cpp:127, 1382); X is column-major n x p in device row order
__global__ void k_xb(const double *X, const int *mv, const double *B, long long n, int p, double *xb) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double *bj = B + (size_t)p * mv[i];
  double acc = 0.0;
  for (int j = 0; j < p; ++j) acc += X[(size_t)j * n + i] * bj[j];
  xb[i] = acc;
}

// partial sums for beta / tausq: per workgroup nq = p*q + q values; stage 2 reduces in workgroup order.
#define STATS_WG 1024
__global__ __launch_bounds__(NT) void k_stats(const double *X, const double *y, const double *w, const double *xb, const int *mv,
                                               const unsigned char *obs, const long long *partner, long long n, int p, int q,
                                               double *partial) {
  __shared__ double s_red[NT / 64];
  const int nq = p * q + q;
  double acc[QMAX * 8 + QMAX];  // p <= 8 enforced on the host for this kernel
  for (int k = 0; k < nq; ++k) acc[k] = 0.0;
  const long long chunk = (n + gridDim.x - 1) / gridDim.x;
  const long long lo = (long long)blockIdx.x * chunk, hi = min(n, lo + chunk);
  for (long long i = lo + threadIdx.x; i < hi; i += NT) {
    if (!obs[i]) continue;
    const int v = mv[i];
    const double rw = y[i] - w[partner[i]];
    for (int j = 0; j < p; ++j) acc[v * p + j] += X[(size_t)j * n + i] * rw;
    const double e = y[i] - xb[i] - w[i];
    acc[p * q + v] += e * e;
  }
  for (int k = 0; k < nq; ++k) {
    const double s = block_sum(acc[k], s_red);
    if (threadIdx.x == 0) partial[(size_t)blockIdx.x * nq + k] = s;
  }
}
// one workgroup per statistic: fixed-shape tree over the STATS_WG partial sums (deterministic)
__global__ __launch_bounds__(NT) void k_stats_final(const double *partial, int nwg, int nq, double *out) {
  __shared__ double sm[NT];
  const int k = blockIdx.x;
  double s = 0.0;
  for (int g = threadIdx.x; g < nwg; g += NT) s += partial[(size_t)g * nq + k];
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int o = NT / 2; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) sm[threadIdx.x] += sm[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[k] = sm[0];
}

__global__ void k_yhat(const double *xb, const double *w, const double *noise, const int *mv, long long n, const double *tsq_inv_q,
                       double *yhat) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) yhat[i] = xb[i] + w[i] + noise[i] / sqrt(tsq_inv_q[mv[i]]);
}


// ---------------------------------------------------------------------------------------------------------------
// "Next" rows of SURVEY.md section 8f: the exported CrossCovarianceAG10 (covariance_functions.cpp:301-355) and
// running posterior means of w / yhat over saved iterations (the use of list_mean, list_mean.cpp:10-40)
// ---------------------------------------------------------------------------------------------------------------
__global__ void k_cross_cov(const double *c1, const int *mv1, long long n1, const double *c2, const int *mv2, long long n2, CovPar cp,
                            double *out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long j = blockIdx.y;
  if (i < n1 && j < n2) out[j * n1 + i] = cov_entry(cp, c1[i], c1[n1 + i], mv1[i], c2[j], c2[n2 + j], mv2[j]);
}
// Posterior quantiles per row over the saved draws (list_qtile / prctile_stl, /root/reference/src/list_mean.cpp:62-137):
// draws[d * n + row], d < keep.  A workgroup sorts the draws of R rows in LDS (bitonic, rows padded to Kpad = 2^k with +inf)
// and applies the reference's interpolation rule between the two order statistics around r = q * keep.
struct QtArgs {
  const double *draws;
  long long n;
  int keep, Kpad, R;
  double q;
  double *out;
};
__global__ __launch_bounds__(NT) void k_qtile(QtArgs A) {
  extern __shared__ double lds[];
  const int tid = threadIdx.x, R = A.R, K = A.Kpad;
  const long long row0 = (long long)blockIdx.x * R;
  for (int idx = tid; idx < R * K; idx += NT) {
    const int d = idx / R, r = idx - d * R;   // R consecutive rows of one draw: contiguous in memory
    double v = __builtin_inf();
    if (d < A.keep && row0 + r < A.n) v = A.draws[(size_t)d * A.n + row0 + r];
    lds[(size_t)r * K + d] = v;
  }
  __syncthreads();
  const int half = K >> 1;
  for (int k = 2; k <= K; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int p = tid; p < R * half; p += NT) {
        const int r = p / half, i = p - r * half;
        const int i1 = 2 * j * (i / j) + (i % j), i2 = i1 + j;
        double *a = lds + (size_t)r * K;
        const double x = a[i1], y = a[i2];
        const bool up = (i1 & k) == 0;
        if ((x > y) == up) { a[i1] = y; a[i2] = x; }
      }
      __syncthreads();
    }
  }
  if (tid < R && row0 + tid < A.n) {
    const double *a = lds + (size_t)tid * K;
    const int len = A.keep;
    // prctile_stl: r = percent / 100 * len with percent = q * 100 (cqtile); every product rounded on its own -- a fused
    // q * len - 1 would see 0.025 * 40 as 1 + 5.6e-17 and pick the other pair of order statistics
    double r = A.q * 100.0, lower, upper;
    asm volatile("" : "+v"(r));   // (an empty asm after each step keeps the optimiser from contracting across it)
    r = r / 100.0;
    asm volatile("" : "+v"(r));
    r = r * (double)len;
    asm volatile("" : "+v"(r));
    if (r >= len / 2.0) {
      const int lo = (int)fmax(r - 1.0, 0.0);
      lower = a[lo];
      upper = lo < len - 1 ? a[lo + 1] : lower;
    } else {
      const int up = (int)ceil(fmax(r - 1.0, 0.0));
      upper = a[up];
      lower = up > 0 ? a[up - 1] : upper;
    }
    const int k = (int)(r + 0.5);                    // implicit floor
    r = r - k;
    A.out[row0 + tid] = (0.5 - r) * lower + (0.5 + r) * upper;
  }
}

__global__ void k_axpy_sum(double *acc, const double *x, long long n) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) acc[i] += x[i];
}

// ===============================================================================================================
// host side
// ===============================================================================================================
static thread_local std::string g_create_error;

template <typename T>
struct DevBuf {
  T *p = nullptr;
  size_t n = 0;
  hipError_t alloc(size_t count) {
    n = count;
    if (count == 0) { p = nullptr; return hipSuccess; }
    return hipMalloc((void **)&p, count * sizeof(T));
  }
  hipError_t upload(const std::vector<T> &v) {
    hipError_t e = alloc(v.size());
    if (e != hipSuccess || v.empty()) return e;
    return hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
  }
  void free() { if (p) (void)hipFree(p); p = nullptr; n = 0; }
};

struct LevelInfo {
  int first = 0, count = 0;   // into lvl_list
  int isref = 1;
  int maxP = 0, maxM = 0, maxMa = 0, maxLd = 0;
  bool big_factor = false, big_sample = false, sample_sq = false;   // sample_sq: k_sample<true> keeps S and chol(S)^-1 in LDS
  size_t lds_factor = 0, lds_sample = 0, lds_loglik = 0;
  double alg_bytes_A = 0, alg_bytes_B = 0, alg_bytes_C = 0, alg_bytes_msg = 0;
  double flops_A = 0, flops_B = 0, flops_C = 0;
  // MFMA fast path of phase A (column groups)
  bool fast = false;
  int grp_first = 0, grp_count = 0, Pm4 = 0, ldKV = 2, ldS = 2, SRm = 1, stage_dbl = 0;
  size_t lds_fast = 0;
  int ldN = 2, Mr4 = 4, Mrows = 1, av_dbl = 224;
  size_t lds_sfast = 0, lds_slean = 0;
  bool bigmfma = false;            // generic level whose phase A takes k_factor_bigmfma
  int wide_first = 0, wide_count = 0, wide_maxN = 0;   // sibling groups of this rank's run (k_factor_wide); count 0: not used
  size_t lds_wide = 0;
  int bm_ldS = 0;
  size_t lds_bigmfma = 0;
  int lchain = 0;                  // non-reference level on k_factor_lchain<lchain> (0: not used)
  int lc_first = 0, lc_count = 0;  // its slabs (this rank's run) in d_lcslabs
  int quad_first = 0, quad_count = 0, qown_lo = 0, qown_n = 0, q_ldS = 0, q_nkx = 0;   // k_factor_quad (q_nkx = 0: not eligible)
  size_t lds_quad = 0;
  int own_lo = 0, own_n = 0, gown_lo = 0, gown_n = 0;   // this rank's run of the level's block list / group list
};

struct st_handle_s {
  std::string err;
  int device = 0;
  hipStream_t stream = nullptr;
  int quirks = 1, force_generic = 0;
  long long n_all = 0, n_blocks = 0;
  int q = 1, p = 1, d = 2, n_groups = 0, n_actual_groups = 0;
  long long n_obs = 0;
  size_t lds_limit = 65536;
  int sm_count = 256;

  std::vector<long long> dev2model, model2dev;       // rows
  std::vector<int> blk_model2dev;                    // blocks
  std::vector<Blk> blks;                             // device block order
  std::vector<int> anc_idx, dch_idx, lvl_list, pred_list, all_obs_list;
  std::vector<Grp> grps;
  DevBuf<Grp> d_grps;
  std::vector<Quad> quads;
  DevBuf<Quad> d_quads;
  std::vector<WideGrp> wgrps;                 // sibling groups of the wide levels (k_factor_wide)
  DevBuf<WideGrp> d_wgrps;
  int wide_on = 1;                            // SPAMTREE_WIDE=0: k_factor_bigmfma (one block per workgroup) instead
  std::vector<LcSlab> lcslabs;                // k_factor_lchain: slabs of sibling groups
  DevBuf<LcSlab> d_lcslabs;
  DevBuf<double> d_lcrow;                     // per-row e^2 | log r of the lchain levels (2 n)
  DevBuf<double> d_s0;                        // Ri' Ri of the reference blocks on the generic phase-B path (theta-only, cached with the Gram parts)
  DevBuf<long long> d_s0off;                  // per block: offset into d_s0, -1 = none
  bool c_pending = false;                     // st_sample_w_loglik_begin: the sweep's failure word and log-density are on their way to pin[8..10]
  int c_rc = 0; double c_ll = 0.0;            // ... or (multi-GPU / communicator attached) already here
  int gram_big = 1;                           // SPAMTREE_GRAM_BIG=0: the generic sweep kernel rebuilds the records' Gram parts itself (one thread per entry)
  int lchain_on = 1;                          // SPAMTREE_LCHAIN=0: non-reference long-chain levels stay on k_factor_wide / k_factor_bigmfma
  std::vector<long long> gdesc;               // group descriptors (GdHead layout), gd_stride words per group
  DevBuf<long long> d_gdesc;
  int gd_stride = 8;
  int quad_nu = 4;
  // multi-GPU sharding
  int rank = 0, world = 1, cut = 0;
  std::vector<int> blk_owner;                 // device block -> owning rank, -1 = replicated
  std::vector<int> own_obs_list;              // observed blocks this rank evaluates in phase C
  DevBuf<int> d_ownobs;
  std::vector<int> own_grp_list, own_obs_slow; // the same set split: column groups of the fast levels / blocks of the others
  DevBuf<int> d_owngrp, d_ownslow;
  DevBuf<unsigned char> d_rowmask, d_blkmask; // 1 = this rank contributes the entry to a sum-with-zeros exchange
  DevBuf<double> d_comm;                      // 2*n_blocks + 64 doubles
  DevBuf<double> d_gather;                    // all-gather of w: world x gather_cnt (a rank's owned rows in device order + its failure word)
  DevBuf<int> d_gidx;                         // device row of every slot of d_gather (-1: padding / the failure word)
  int gather_cnt = 1;
  DevBuf<double> d_gerr;                      // the ranks' failure words after the all-gather (64)
  // phase A of the latency-bound top levels ahead of time (st_factor_begin): they depend on theta only -- except for the
  // blocks' quadratic forms, redone with the current w afterwards -- and run on a second stream under the sweep
  hipStream_t stream2 = nullptr;
  hipEvent_t ev_top = nullptr, ev_main = nullptr;
  DevBuf<int> d_err2, d_toplist;
  int n_toplist = 0, g_top = 0;
  bool async_top = false, top_pending = false, prof_suspend = false;
  hipEvent_t ev_stats = nullptr;
  bool stats_on_stream2 = false;   // the statistics kernels of the current (w, XB) are in flight on the second stream
  bool stats_prefetched = false;   // ... and their results follow them to pin[20 ..] on that stream
  int top_phys = -1;
  std::vector<double> top_theta;
  long long top_off = 0, top_len = 0;         // message records of the cut level inside `acc`
  std::vector<std::pair<long long, long long>> top_zero;   // sub-ranges of it owned by other ranks
  bool ext_stream = false;
  DevBuf<double> d_sum_w, d_sum_yhat;         // running sums over saved iterations (st_summary_*)
  long long n_summary = 0;
  DevBuf<double> d_draws_w, d_draws_yhat;     // st_summary_reserve: the saved draws themselves, [keep][n_all] (quantiles)
  long long draws_cap = 0, n_draws = 0;
  int factor_gen = 1;
  int sample_lean = 1;                        // sweeps with cached Gram parts take k_sample_lean (SPAMTREE_SAMPLE_LEAN=0: never)
  int sample_wave = 1;                        // reference blocks of <= 27 rows: one block per wave (SPAMTREE_SAMPLE_WAVE=0: k_sample_lean)
  int split_gram = 1;                         // sweeps that rebuild the Gram parts: k_gram + lean kernels (SPAMTREE_SPLIT_GRAM=0: k_sample_mfma)
  bool stats_valid = false;                   // d_stats matches the current w and XB
  bool host_stats_valid = false;              // ... and host_stats holds a copy of it
  std::vector<double> host_stats;
  double *pin = nullptr;                      // 64 doubles of pinned host memory for the small device-to-host reads
  bool gram_valid = false;                    // message Gram parts in `acc` match the accepted theta (slot 0)
  bool cache_gram = true;
  bool limited = false;               // limited_tree: single parents, marginal chain factors (k_marginal_invchol)
  std::vector<int> twin_list;         // limited_tree: device ids of the blocks that own a chain panel
  DevBuf<int> d_twin;
  int twin_maxM = 1;
  ncclComm_t comm = nullptr;                  // native RCCL communicator (st_comm_init); null = exchanges are the caller's
  std::vector<LevelInfo> levels;
  LevelInfo pred_info;
  std::vector<double> xtx;
  std::vector<long long> n_obs_q;

  DevBuf<double> d_cx, d_cy, d_y, d_X, d_w, d_xb, d_z, d_B, d_panels[2], d_acc, d_logdet[2], d_loglik[2], d_scalars, d_partial,
      d_stats, d_scratch, d_tmp_n, d_tsq;
  DevBuf<int> d_mv, d_anc, d_dch, d_lvl, d_pred, d_allobs, d_err;
  DevBuf<unsigned char> d_obs;
  DevBuf<long long> d_dev2model, d_partner;
  DevBuf<Blk> d_blks;
  size_t panel_total = 0, acc_total = 0;
  long long scratch_stride = 0;
  int scratch_wgs = 0;
  int slot_map[2] = {0, 1};    // logical slot (0 param, 1 alter) -> physical arena
  double tausq_inv[QMAX];
  std::vector<double> theta[2];
  bool z_valid = false;

  // profiling
  int prof = 0;   // 0 off, 1 every kernel family, 2 phase A only (the roofline measurement at the lowest cost)
  double prof_ms[ST_N_KERNEL_FAMILIES] = {0};
  long long prof_n[ST_N_KERNEL_FAMILIES] = {0};
  std::vector<double> prof_level_ms;   // phase-A time per level, accumulated
  std::vector<long long> prof_level_n;
  struct ProfRec { hipEvent_t a, b; int fam, level, count; };   // count: kernel launches inside the bracket
  std::vector<ProfRec> prof_pending;
  std::vector<hipEvent_t> ev_free;
};

#define HCHK(h, call)                                                                                         \
  do {                                                                                                        \
    hipError_t e_ = (call);                                                                                   \
    if (e_ != hipSuccess) {                                                                                   \
      (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);                                           \
      return ST_ERR_HIP;                                                                                      \
    }                                                                                                         \
  } while (0)

static int fail_create(st_handle_s *h, int code, const std::string &msg) {
  g_create_error = msg;
  if (h) {
    st_destroy(h);
  }
  return code;
}

// Launch timing with HIP events on the launch stream, harvested lazily (no host sync inside the measured region).
static hipEvent_t prof_event(st_handle_s *h) {
  if (!h->ev_free.empty()) { hipEvent_t e = h->ev_free.back(); h->ev_free.pop_back(); return e; }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}
static void prof_harvest(st_handle_s *h) {
  for (auto &r : h->prof_pending) {
    float ms = 0.f;
    if (hipEventSynchronize(r.b) == hipSuccess && hipEventElapsedTime(&ms, r.a, r.b) == hipSuccess) {
      h->prof_ms[r.fam] += ms;
      h->prof_n[r.fam] += r.count;
      if (r.level >= 0 && r.level < (int)h->prof_level_ms.size()) { h->prof_level_ms[r.level] += ms; h->prof_level_n[r.level] += 1; }
    }
    h->ev_free.push_back(r.a);
    h->ev_free.push_back(r.b);
  }
  h->prof_pending.clear();
}
struct ProfScope {
  st_handle_s *h;
  st_handle_s::ProfRec r;
  // mode 1: every launch is bracketed; mode 2: only the whole-phase bracket of phase A (level == -2), one pair of events
  hipStream_t st;
  ProfScope(st_handle_s *h_, int fam, int level = -1, int count = 1, hipStream_t st_ = nullptr) : h(h_), st(st_ ? st_ : h_->stream) {
    r.fam = fam; r.level = level; r.count = count; r.a = r.b = nullptr;
    const bool on = level == -2 ? (h->prof == 2 && !h->prof_suspend) : h->prof == 1;
    if (on) { r.a = prof_event(h); r.b = prof_event(h); (void)hipEventRecord(r.a, st); }
  }
  ~ProfScope() {
    if (r.a && r.b) {
      (void)hipEventRecord(r.b, st);
      h->prof_pending.push_back(r);
      if (h->prof_pending.size() > 8192) prof_harvest(h);
    }
  }
};

// w or XB is about to change: the cached statistics die; a reduction still in flight on the second stream finishes first
static void invalidate_stats(st_handle_s *h) {
  if (h->stats_on_stream2) {
    (void)hipSetDevice(h->device);
    (void)hipStreamWaitEvent(h->stream, h->ev_stats, 0);
    h->stats_on_stream2 = false;
  }
  h->stats_valid = false; h->host_stats_valid = false;
}

static size_t lds_factor_bytes(int maxP, int maxM, int maxMa, int SR, bool big) {
  size_t dbl = (size_t)3 * (maxP + maxM) + 3 * (size_t)maxM + (size_t)SR * maxP;
  size_t bytes = dbl * 8 + (size_t)((maxP + maxM + 1) & ~1) * 4;
  if (!big) bytes += ((size_t)2 * maxP * maxM + (size_t)maxMa * maxM + (size_t)2 * maxM * maxM) * 8;
  return bytes + 64;
}
static size_t scratch_factor_doubles(int maxP, int maxM, int maxMa) {
  return (size_t)2 * maxP * maxM + (size_t)maxMa * maxM + (size_t)2 * maxM * maxM;
}
static size_t lds_sample_sq_bytes(int maxM) { return ((size_t)maxM * ((maxM + 7) | 1) + maxM + 16) * 8; }   // S (odd row stride) + the pivot column
static size_t lds_sample_bytes(int maxP, int maxM, int maxLd, bool big) {
  size_t dbl = (size_t)(maxP + maxM) + 4 * (size_t)maxM + (size_t)MAXJ * maxM;   // ... + segment sums seg[t][r]
  if (!big) dbl += (size_t)maxM * maxLd + (size_t)maxM * maxM;
  return dbl * 8 + 64;
}
static size_t lds_loglik_bytes(int maxP, int maxM) { return ((size_t)maxP + 2 * (size_t)maxM) * 8 + 64; }

extern "C" const char *st_last_error(st_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }
extern "C" void st_set_create_error(const char *msg) { g_create_error = msg ? msg : ""; }   // other translation units of the library

extern "C" int st_destroy(st_handle h) {
  if (!h) return ST_OK;
  (void)hipSetDevice(h->device);
  h->d_cx.free(); h->d_cy.free(); h->d_y.free(); h->d_X.free(); h->d_w.free(); h->d_xb.free(); h->d_z.free(); h->d_B.free();
  h->d_panels[0].free(); h->d_panels[1].free(); h->d_acc.free();
  for (int s = 0; s < 2; ++s) { h->d_logdet[s].free(); h->d_loglik[s].free(); }
  h->d_scalars.free(); h->d_partial.free(); h->d_stats.free(); h->d_scratch.free(); h->d_tmp_n.free(); h->d_tsq.free();
  h->d_mv.free(); h->d_anc.free(); h->d_dch.free(); h->d_lvl.free(); h->d_pred.free(); h->d_allobs.free(); h->d_err.free();
  h->d_twin.free(); h->d_wgrps.free(); h->d_lcslabs.free(); h->d_lcrow.free(); h->d_s0.free(); h->d_s0off.free(); h->d_obs.free(); h->d_dev2model.free(); h->d_partner.free(); h->d_blks.free(); h->d_grps.free(); h->d_quads.free(); h->d_gdesc.free();
  h->d_ownobs.free(); h->d_owngrp.free(); h->d_ownslow.free(); h->d_rowmask.free(); h->d_blkmask.free(); h->d_comm.free(); h->d_gather.free(); h->d_gidx.free(); h->d_gerr.free(); h->d_err2.free(); h->d_toplist.free();
  if (h->ev_top) (void)hipEventDestroy(h->ev_top);
  if (h->ev_main) (void)hipEventDestroy(h->ev_main);
  if (h->ev_stats) (void)hipEventDestroy(h->ev_stats);
  if (h->stream2) (void)hipStreamDestroy(h->stream2); h->d_sum_w.free(); h->d_sum_yhat.free(); h->d_draws_w.free(); h->d_draws_yhat.free();
  prof_harvest(h);
  for (auto e : h->ev_free) (void)hipEventDestroy(e);
  if (h->comm) (void)ncclCommDestroy(h->comm);
  if (h->pin) (void)hipHostFree(h->pin);
  if (h->stream && !h->ext_stream) (void)hipStreamDestroy(h->stream);
  delete h;
  return ST_OK;
}

static int create_impl(const st_problem *pb, const st_options *opt, st_handle *out, bool plan_only, int64_t *owner_out, int32_t *cut_out);

extern "C" int st_create(const st_problem *pb, const st_options *opt, st_handle *out) {
  return create_impl(pb, opt, out, false, nullptr, nullptr);
}

// Pure host: which rank owns each block (0-based block id) when `world` processes share one problem; -1 = replicated
// on every rank (the levels above the cut).  No GPU needed: lets the sharding plan be tested on CPU.
extern "C" int st_shard_plan(const st_problem *pb, int32_t world, int64_t *owner, int32_t *cut_level) {
  st_options opt = {0, 1, 0, world, 0, 0};
  st_handle dummy = nullptr;
  return create_impl(pb, &opt, &dummy, true, owner, cut_level);
}

static int create_impl(const st_problem *pb, const st_options *opt, st_handle *out, bool plan_only, int64_t *owner_out, int32_t *cut_out) {
  if (!pb || !out) { g_create_error = "st_create: null argument"; return ST_ERR_USAGE; }
  *out = nullptr;
  if (pb->d != 2) { g_create_error = "only d=2 is reachable from spamtree() (R/spamtree_fit.R:58-60)"; return ST_ERR_UNSUPPORTED; }
  if (pb->q < 1 || pb->q > QMAX) { g_create_error = "q out of range"; return ST_ERR_UNSUPPORTED; }
  if (pb->p < 1 || pb->p > 8) { g_create_error = "p must be in 1..8"; return ST_ERR_UNSUPPORTED; }
  if (opt && (opt->world < 1 || opt->rank < 0 || opt->rank >= opt->world || opt->world > 64)) { g_create_error = "bad rank/world"; return ST_ERR_USAGE; }
  st_handle_s *h = new st_handle_s();
  h->rank = opt ? opt->rank : 0;
  h->world = opt ? opt->world : 1;
  h->device = opt ? opt->device : 0;
  h->quirks = opt ? opt->reference_quirks : 1;
  h->force_generic = opt ? opt->force_generic : 0;
  h->cache_gram = !(opt && (opt->reserved & 1));
  h->limited = opt && (opt->reserved & 2);
  if (h->limited && h->world > 1) return fail_create(h, ST_ERR_UNSUPPORTED, "limited_tree with world > 1");
  const long long n = pb->n_all, nb = pb->n_blocks;
  h->n_all = n; h->n_blocks = nb; h->q = pb->q; h->p = pb->p; h->d = pb->d; h->n_groups = pb->n_groups;
  for (int j = 0; j < QMAX; ++j) h->tausq_inv[j] = 1.0;

  // ---- block census (na_study :303-313), levels (make_gibbs_groups :194-301)
  std::vector<int> m_of(nb), obs_of(nb, 0), grp_of(nb);
  std::vector<long long> labels(pb->block_groups, pb->block_groups + nb);
  std::sort(labels.begin(), labels.end());
  labels.erase(std::unique(labels.begin(), labels.end()), labels.end());
  if ((int)labels.size() > pb->n_groups) return fail_create(h, ST_ERR_TOPOLOGY, "more levels in block_groups than entries in res_is_ref");
  std::vector<char> row_seen(n, 0);
  for (long long u = 0; u < nb; ++u) {
    m_of[u] = (int)(pb->indexing_ptr[u + 1] - pb->indexing_ptr[u]);
    grp_of[u] = (int)(std::lower_bound(labels.begin(), labels.end(), pb->block_groups[u]) - labels.begin());
    for (long long k = pb->indexing_ptr[u]; k < pb->indexing_ptr[u + 1]; ++k) {
      const long long r = pb->indexing_idx[k];
      if (r < 0 || r >= n || row_seen[r]) return fail_create(h, ST_ERR_TOPOLOGY, "indexing is not a partition of the rows");
      row_seen[r] = 1;
      if (std::isfinite(pb->y[r])) obs_of[u]++;
    }
  }
  for (long long r = 0; r < n; ++r)
    if (!row_seen[r]) return fail_create(h, ST_ERR_TOPOLOGY, "row without a block");
  const int G = (int)labels.size();
  std::vector<int> grp_has_obs(G, 0);
  for (long long u = 0; u < nb; ++u)
    if (obs_of[u] > 0) grp_has_obs[grp_of[u]] = 1;
  int n_actual = 0;
  for (int g = 0; g < G; ++g) n_actual += grp_has_obs[g];
  for (int g = 0; g < n_actual; ++g)
    if (!grp_has_obs[g]) return fail_create(h, ST_ERR_TOPOLOGY, "an empty level precedes an observed one");
  h->n_actual_groups = n_actual;

  // ---- device block order: by (level, id); rows contiguous per block
  std::vector<int> order(nb);
  std::iota(order.begin(), order.end(), 0);
  std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return grp_of[a] < grp_of[b]; });
  h->blk_model2dev.assign(nb, -1);
  {
    // inside a level, blocks with the same last parent (identical ancestor chain) are made contiguous, so a
    // workgroup can take several sibling leaf blocks as one column group
    int i0 = 0;
    while (i0 < nb) {
      int i1 = i0;
      while (i1 < nb && grp_of[order[i1]] == grp_of[order[i0]]) ++i1;
      auto key = [&](int u) -> long long {
        const long long p0 = pb->parents_ptr[u], p1 = pb->parents_ptr[u + 1];
        if (p1 == p0) return -1;
        const long long a = pb->parents_idx[p1 - 1];
        return (a >= 0 && a < nb) ? (long long)h->blk_model2dev[a] : -1;
      };
      std::stable_sort(order.begin() + i0, order.begin() + i1, [&](int a, int b) { return key(a) < key(b); });
      for (int i = i0; i < i1; ++i) h->blk_model2dev[order[i]] = i;
      i0 = i1;
    }
  }
  h->dev2model.resize(n); h->model2dev.resize(n);
  h->blks.resize(nb);
  long long row = 0, panel_total = 0, acc_total = 0;
  for (int i = 0; i < nb; ++i) {
    const int u = order[i];
    Blk &B = h->blks[i];
    B.row0 = row; B.m = m_of[u]; B.level = grp_of[u]; B.model_id = u; B.nobs = obs_of[u];
    for (long long k = pb->indexing_ptr[u]; k < pb->indexing_ptr[u + 1]; ++k) {
      if (k > pb->indexing_ptr[u] && pb->indexing_idx[k] <= pb->indexing_idx[k - 1])
        return fail_create(h, ST_ERR_TOPOLOGY, "indexing(u) must be ascending");
      h->dev2model[row] = pb->indexing_idx[k];
      h->model2dev[pb->indexing_idx[k]] = row;
      ++row;
    }
  }
  // ---- ancestors: chain property anc(u) = anc(last parent) + [last parent]
  for (int i = 0; i < nb; ++i) {
    const int u = order[i];
    Blk &B = h->blks[i];
    const long long p0 = pb->parents_ptr[u], p1 = pb->parents_ptr[u + 1];
    B.nanc = (int)(p1 - p0);
    if (B.nanc > MAXJ) return fail_create(h, ST_ERR_UNSUPPORTED, "more than ST_MAX_ANCESTORS ancestors");
    B.anc_ptr = (int)h->anc_idx.size();
    int P = 0;
    for (long long k = p0; k < p1; ++k) {
      const long long a = pb->parents_idx[k];
      if (a < 0 || a >= nb) return fail_create(h, ST_ERR_TOPOLOGY, "parent id out of range");
      if (k > p0 && a <= pb->parents_idx[k - 1]) return fail_create(h, ST_ERR_TOPOLOGY, "parents(u) must be ascending");
      if (grp_of[a] >= grp_of[u]) return fail_create(h, ST_ERR_TOPOLOGY, "parent on the same or a deeper level");
      if (pb->res_is_ref[grp_of[a]] != 1) return fail_create(h, ST_ERR_TOPOLOGY, "parent on a non-reference level");
      if (obs_of[a] == 0) return fail_create(h, ST_ERR_TOPOLOGY, "ancestor block without observations");
      h->anc_idx.push_back(h->blk_model2dev[a]);
      P += m_of[a];
    }
    B.P = P;
    if (h->limited) {
      if (B.nanc > 1) return fail_create(h, ST_ERR_TOPOLOGY, "limited_tree: a block has more than one parent (make_edges_limited gives one)");
    } else if (B.nanc > 0) {
      const long long last = pb->parents_idx[p1 - 1];
      const long long q0 = pb->parents_ptr[last], q1 = pb->parents_ptr[last + 1];
      bool ok = (q1 - q0) == (p1 - p0 - 1);
      for (long long k = 0; ok && k < q1 - q0; ++k) ok = pb->parents_idx[q0 + k] == pb->parents_idx[p0 + k];
      if (!ok) return fail_create(h, ST_ERR_UNSUPPORTED, "parents(u) is not parents(last parent)+[last parent]: for make_edges_limited's single-parent lists set the limited_tree bit of st_options");
    }
    const bool observed = B.nobs > 0;
    B.isref = (observed && B.level < pb->n_groups && pb->res_is_ref[B.level] == 1) ? 1 : 0;
    B.ld = B.P + (B.isref ? B.m : 1);
    B.panel_off = -1; B.acc_off = 0; B.acc_len = 0;
    B.chain_off = -1;
    if (observed) {
      B.panel_off = panel_total;
      panel_total += (long long)B.m * B.ld;
      B.chain_off = B.panel_off;
      if (h->limited) {
        B.chain_off = -1;
        if (B.isref) {   // every observed reference block may be somebody's parent (observed or prediction children)
          B.chain_off = panel_total;
          panel_total += (long long)B.m * B.m;
          h->twin_list.push_back(i);
          h->twin_maxM = std::max(h->twin_maxM, B.m);
        }
      }
    }
  }
  // acc layout + direct children
  std::vector<std::vector<int>> dch(nb);
  for (int i = 0; i < nb; ++i) {
    Blk &B = h->blks[i];
    if (B.nobs == 0) continue;
    long long len = 0;
    for (int t = 0; t < B.nanc; ++t) {
      const int ma = h->blks[h->anc_idx[B.anc_ptr + t]].m;
      len += (long long)ma * ma + ma;
    }
    if (len > INT_MAX) return fail_create(h, ST_ERR_UNSUPPORTED, "message record too large");
    B.acc_len = (int)len;
    B.acc_off = acc_total;
    acc_total += len;
    if (B.nanc > 0) dch[h->anc_idx[B.anc_ptr + B.nanc - 1]].push_back(i);
  }
  h->panel_total = (size_t)panel_total;
  h->acc_total = (size_t)acc_total;

  // ---- multi-GPU ownership: whole subtrees below a cut level go to one rank, the levels
  // above the cut are replicated.  cut = first reference level (not the last observed one) with >= 2*world
  // observed blocks; its blocks, contiguous in device order, are split into `world` runs of equal weight
  // (weight = sum over the subtree of m*P^2, the factorisation cost).
  h->blk_owner.assign(nb, -1);
  h->cut = n_actual;            // nothing sharded unless a cut is found
  if (h->world > 1) {
    std::vector<int> cnt(G, 0);
    for (int i = 0; i < nb; ++i) if (h->blks[i].nobs > 0) cnt[h->blks[i].level]++;
    for (int g = 0; g + 1 < n_actual; ++g)
      if (pb->res_is_ref[g] == 1 && cnt[g] >= 2 * h->world) { h->cut = g; break; }
    if (h->cut < n_actual) {
      const int cut = h->cut;
      std::vector<int> root_of(nb, -1);   // device index of the cut-level ancestor (or self)
      std::vector<double> wsub(nb, 0.0);
      for (int i = 0; i < nb; ++i) {
        const Blk &B = h->blks[i];
        if (B.level < cut) continue;
        int r = -1;
        if (B.level == cut) r = i;
        else for (int t = 0; t < B.nanc; ++t) { const int a = h->anc_idx[B.anc_ptr + t]; if (h->blks[a].level == cut) r = a; }
        if (r < 0) return fail_create(h, ST_ERR_TOPOLOGY, "block below the cut level without an ancestor on it");
        root_of[i] = r;
        wsub[r] += (double)B.m * ((double)B.P * B.P + 1.0);
      }
      std::vector<int> roots;
      double tot = 0;
      for (int i = 0; i < nb; ++i) if (h->blks[i].level == cut && h->blks[i].nobs > 0) { roots.push_back(i); tot += wsub[i]; }
      double acc_w = 0;
      std::vector<int> root_owner(nb, 0);
      for (size_t k = 0; k < roots.size(); ++k) {
        int r = (int)std::floor((acc_w + 0.5 * wsub[roots[k]]) / tot * h->world);
        r = std::min(std::max(r, 0), h->world - 1);
        if (k > 0) r = std::max(r, root_owner[roots[k - 1]]);   // keep runs contiguous
        root_owner[roots[k]] = r;
        acc_w += wsub[roots[k]];
      }
      for (int i = 0; i < nb; ++i) if (root_of[i] >= 0) h->blk_owner[i] = root_owner[root_of[i]];
    }
  }
  if (plan_only) {
    if (owner_out) for (int i = 0; i < nb; ++i) owner_out[h->blks[i].model_id] = h->blk_owner[i];
    if (cut_out) *cut_out = h->cut;
    delete h;
    return ST_OK;
  }

  // ---- level lists (u_by_block_groups) and per-level launch geometry
  hipDeviceProp_t prop;
  if (hipSetDevice(h->device) != hipSuccess || hipGetDeviceProperties(&prop, h->device) != hipSuccess)
    return fail_create(h, ST_ERR_HIP, "no usable HIP device (the product path has no CPU fallback)");
  h->sm_count = prop.multiProcessorCount;
  {
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, h->device) == hipSuccess && v > 0) h->lds_limit = (size_t)v;
    if (h->lds_limit > 160 * 1024) h->lds_limit = 160 * 1024;
    h->quad_nu = 4;   // units per workgroup of k_factor_quad (2 per workgroup with two workgroups per CU measured slower)
    { const char *e = getenv("SPAMTREE_WIDE"); h->wide_on = (e && e[0] == '0') ? 0 : ((e && e[0] == '2') ? 2 : 1); }
    { const char *e = getenv("SPAMTREE_LCHAIN"); h->lchain_on = (e && e[0] == '0') ? 0 : 1; }
    { const char *e = getenv("SPAMTREE_GRAM_BIG"); h->gram_big = (e && e[0] == '0') ? 0 : 1; }
  }
  h->levels.resize(n_actual);
  auto geometry = [&](LevelInfo &L, const std::vector<int> &list, bool is_pred) {
    for (int b : list) {
      const Blk &B = h->blks[b];
      L.maxP = std::max(L.maxP, B.P); L.maxM = std::max(L.maxM, B.m); L.maxLd = std::max(L.maxLd, B.ld);
      for (int t = 0; t < B.nanc; ++t) L.maxMa = std::max(L.maxMa, h->blks[h->anc_idx[B.anc_ptr + t]].m);
      // algorithmic bytes / flops, SURVEY.md section 8d
      const double m = B.m, P = B.P, tri = P * (P + 1) / 2, rim = B.isref ? m * (m + 1) / 2 : m;
      double trisum = 0;
      for (int t = 0; t < B.nanc; ++t) { const double ma = h->blks[h->anc_idx[B.anc_ptr + t]].m; trisum += ma * (ma + 1) / 2; }
      if (!is_pred) {
        L.alg_bytes_A += (8.0 * 2 + 8) * (m + P) + (h->q > 1 ? 4 * (m + P) : 0) + 8 * tri + 8 * m * P + 8 * rim + 16;
        L.alg_bytes_B += 8 * m * P + 8 * rim + 8 * P + 40 * m;
        L.alg_bytes_C += 8 * m * P + 8 * rim + 8 * (m + P) + 8;
        L.alg_bytes_msg += 2 * 8 * (P + trisum);
        L.flops_A += 2 * m * P * P + (B.isref ? 2 * m * m * P + m * m * m : 0);
        L.flops_B += (B.isref ? 2.0 / 3 * m * m * m : 0) + 4 * m * P;
        for (int t = 0; t < B.nanc; ++t) { const double ma = h->blks[h->anc_idx[B.anc_ptr + t]].m; L.flops_B += 2 * ma * ma * m; }
        L.flops_C += 2 * m * P + (B.isref ? m * m : 0);
      }
    }
    L.maxMa = std::max(L.maxMa, 1);
    const int SR = 8;
    L.lds_factor = lds_factor_bytes(L.maxP, L.maxM, L.maxMa, SR, false);
    L.big_factor = h->force_generic || L.lds_factor > h->lds_limit;
    if (L.big_factor) L.lds_factor = lds_factor_bytes(L.maxP, L.maxM, L.maxMa, 4, true);
    L.lds_sample = lds_sample_bytes(L.maxP, L.maxM, L.maxLd, false);
    L.big_sample = h->force_generic || L.lds_sample > h->lds_limit || (L.isref && L.maxM > 32 && L.maxM <= 80);   // wide reference blocks: the
    // scratch-arena kernel has the blocked matrix-core solve (the LDS-panel kernel factorises with three barriers per pivot)
    if (L.big_sample) {
      L.lds_sample = lds_sample_bytes(L.maxP, L.maxM, L.maxLd, true);
      // the posterior precision in LDS, factorised and solved by ONE wave without workgroup barriers (wave_chol_solve_lds):
      // every reference level where it fits (config #4: 74 KB, two workgroups per CU).  [The earlier LDS variant -- S and
      // chol(S)^-1, 100 KB, one barrier per pivot -- only paid on levels of at most 2 x CUs blocks.]
      if (L.isref && L.maxM <= 80 && L.lds_sample + lds_sample_sq_bytes(L.maxM) <= h->lds_limit) {
        L.lds_sample += lds_sample_sq_bytes(L.maxM); L.sample_sq = true;
      }
    }
    L.lds_loglik = lds_loglik_bytes(L.maxP, L.maxM);
  };
  for (int g = 0; g < n_actual; ++g) {
    LevelInfo &L = h->levels[g];
    L.first = (int)h->lvl_list.size();
    L.isref = (int)pb->res_is_ref[g];
    // reference order inside a level: block_names order (make_gibbs_groups :238-246); order is irrelevant on device
    for (long long i = 0; i < nb; ++i) {
      const long long u = pb->block_names[i] - 1;
      if (u < 0 || u >= nb) return fail_create(h, ST_ERR_TOPOLOGY, "block_names out of range");
      if (grp_of[u] == g && obs_of[u] > 0) h->lvl_list.push_back(h->blk_model2dev[u]);
    }
    L.count = (int)h->lvl_list.size() - L.first;
    std::sort(h->lvl_list.begin() + L.first, h->lvl_list.end());
    std::vector<int> list(h->lvl_list.begin() + L.first, h->lvl_list.end());
    geometry(L, list, false);
    // column groups for the MFMA path: a reference block alone, or consecutive sibling non-reference blocks
    {
      L.grp_first = (int)h->grps.size();
      bool ok = !h->force_generic && L.maxP <= 256 && L.maxMa <= 32;
      int maxM = 0, maxKb = 0, maxSub = 1;
      size_t i = 0;
      while (ok && i < list.size()) {
        const Blk &B = h->blks[list[i]];
        Grp G;
        G.row0 = B.row0; G.blk0 = list[i]; G.nblk = 1; G.M = B.m; G.P = B.P;
        if (B.m > 32) { ok = false; break; }
        size_t j = i + 1;
        if (!B.isref) {
          const int lastp = B.nanc ? h->anc_idx[B.anc_ptr + B.nanc - 1] : -1;
          while (j < list.size() && G.nblk < 32) {
            const Blk &C = h->blks[list[j]];
            const int lp = C.nanc ? h->anc_idx[C.anc_ptr + C.nanc - 1] : -1;
            if (C.isref || lp != lastp || list[j] != list[j - 1] + 1 || G.M + C.m > 32) break;
            G.M += C.m; G.nblk += 1; ++j;
          }
        }
        maxM = std::max(maxM, G.M);
        for (int t = 0; t < B.nanc; ++t) {
          const int ma = h->blks[h->anc_idx[B.anc_ptr + t]].m;
          maxSub = std::max(maxSub, ma > 16 ? (ma + 1) / 2 : ma);
        }
        maxKb = std::max(maxKb, B.P);
        h->grps.push_back(G);
        i = j;
      }
      L.grp_count = (int)h->grps.size() - L.grp_first;
      if (ok) {
        L.Pm4 = (L.maxP + 3) & ~3;
        L.ldKV = std::max(2, (maxM + 1) & ~1);
        int ldS = std::max(2, maxKb + 4);                       // 4 zero-filled pad columns per staged row
        while ((ldS & 1) || ((ldS >> 1) & 1) == 0) ++ldS;   // 2 * odd: conflict-free A-operand reads
        L.ldS = ldS; L.SRm = maxSub;
        size_t st = (size_t)L.SRm * L.ldS + 16;
        st = std::max(st, (size_t)2 * L.Pm4 + L.Pm4 / 2 + 2);   // prologue alias: ancestor x, y, outcome ids
        st = std::max(st, (size_t)2 * 32 * CH_LD + 216 + 36);     // epilogue alias: R, Ri (stride CH_LD), elimination scratch
        st = ((st + 1) & ~(size_t)1) + (size_t)L.ldS + 16;       // + the zero row at the end
        L.stage_dbl = (int)((st + 1) & ~(size_t)1);
        L.lds_fast = ((size_t)L.Pm4 * L.ldKV + 16 + L.stage_dbl + FM_VPART + 5 * 32) * 8 + 64 * 4 + 64;
        ok = L.lds_fast <= h->lds_limit;
      }
      if (ok) {
        L.Mr4 = std::max(4, (maxM + 3) & ~3);
        L.Mrows = std::max(1, maxM);
        L.ldN = L.maxLd | 1;              // odd stride >= the longest panel row
        int maxJ = 0;
        for (int b : list) maxJ = std::max(maxJ, h->blks[b].nanc);
        L.av_dbl = std::max(32 * maxJ, 224);
        const size_t dbl = (size_t)maxM * L.ldN + 32 + (size_t)L.maxP + 32 + 6 * 32 + (size_t)L.av_dbl + 16 + (L.isref ? (size_t)maxM * CH_LD : 0) + 16;
        L.lds_sfast = dbl * 8 + 64 * 4 + 64;
        L.lds_slean = ((size_t)L.maxP + 32 + (size_t)L.av_dbl + 224 + 16 + 7 * 32 + (L.isref ? 2 * 32 * CH_LD : 0) + 16) * 8;
        ok = L.lds_sfast <= h->lds_limit;
      }
      L.fast = ok;
      if (!ok) { h->grps.resize(L.grp_first); L.grp_count = 0; }
      if (!ok && !h->force_generic && L.maxM <= 80 && L.maxP <= BM_MAXP && L.maxP > 0) {
        int ldS = L.maxP + 24;
        while ((ldS & 1) || ((ldS >> 1) & 1) == 0) ++ldS;
        L.bm_ldS = ldS;
        const size_t work = std::max((size_t)17 * ldS + 16 * 80 + BM_KS * 5 * 256, (size_t)2 * L.maxM * L.maxM + 64);   // stage + zero row + V tile + partial V tiles | R, Ri of the epilogue
        L.lds_bigmfma = ((size_t)3 * (L.maxP + L.maxM) + 3 * (size_t)L.maxM + work) * 8 + (size_t)((L.maxP + L.maxM + 1) & ~1) * 4 + 64;
        L.bigmfma = L.lds_bigmfma <= h->lds_limit;
        // non-reference blocks, <= 64 columns, every block behind at least one ancestor: k_factor_lchain (K in registers, the
        // chain factor streamed through LDS twice); a property of the level, the same on every rank
        if (L.bigmfma && h->lchain_on && !h->limited && !L.isref && L.maxM <= 64 && L.maxP <= 544) {
          bool all_anc = true;
          for (int b : list) all_anc = all_anc && h->blks[b].nanc >= 1 && !h->blks[b].isref;
          if (all_anc) L.lchain = L.maxP <= 384 ? 96 : 136;
        }
      }
    }
    if (L.lds_factor > h->lds_limit || L.lds_sample > h->lds_limit || L.lds_loglik > h->lds_limit)
      return fail_create(h, ST_ERR_UNSUPPORTED, "block too large for the LDS-resident vectors");
  }
  // direct children that hold a message record: every observed block of a generic level, the first block of each
  // column group of a fast level (the group's record is the sum over its sibling blocks)
  {
    std::vector<char> holder(nb, 0);
    for (int g = 0; g < n_actual; ++g) {
      const LevelInfo &L = h->levels[g];
      if (L.fast) for (int k = 0; k < L.grp_count; ++k) holder[h->grps[L.grp_first + k].blk0] = 1;
      else for (int k = 0; k < L.count; ++k) holder[h->lvl_list[L.first + k]] = 1;
    }
    for (int i = 0; i < nb; ++i) {
      Blk &B = h->blks[i];
      B.dch_ptr = (int)h->dch_idx.size();
      B.ndch = 0;
      if (!dch[i].empty() && !B.isref) return fail_create(h, ST_ERR_TOPOLOGY, "a non-reference block has observed children");
      for (int c : dch[i]) if (holder[c]) { h->dch_idx.push_back(c); B.ndch++; }
      if (B.ndch > 64 && B.nobs > 0 && h->levels[B.level].fast)
        return fail_create(h, ST_ERR_UNSUPPORTED, "more than 64 direct child groups under one block");
    }
  }
  // this rank's runs per level, exchange masks, cut-level record region
  std::vector<unsigned char> rowmask(n, 0), blkmask(nb, 0);
  for (int g = 0; g < n_actual; ++g) {
    LevelInfo &L = h->levels[g];
    L.own_lo = 0; L.own_n = L.count; L.gown_lo = 0; L.gown_n = L.grp_count;
    if (g >= h->cut) {
      int lo = L.count, hi = 0;
      for (int k = 0; k < L.count; ++k)
        if (h->blk_owner[h->lvl_list[L.first + k]] == h->rank) { lo = std::min(lo, k); hi = std::max(hi, k + 1); }
      L.own_lo = lo < hi ? lo : 0; L.own_n = lo < hi ? hi - lo : 0;
      for (int k = L.own_lo; k < L.own_lo + L.own_n; ++k)
        if (h->blk_owner[h->lvl_list[L.first + k]] != h->rank) return fail_create(h, ST_ERR_TOPOLOGY, "a rank's blocks are not contiguous in a level");
      if (L.fast) {
        int glo = L.grp_count, ghi = 0;
        for (int k = 0; k < L.grp_count; ++k) {
          const Grp &Gr = h->grps[L.grp_first + k];
          const bool mine = h->blk_owner[Gr.blk0] == h->rank;
          for (int b2 = 0; b2 < Gr.nblk; ++b2)
            if ((h->blk_owner[Gr.blk0 + b2] == h->rank) != mine) return fail_create(h, ST_ERR_TOPOLOGY, "a column group straddles two ranks");
          if (mine) { glo = std::min(glo, k); ghi = std::max(ghi, k + 1); }
        }
        L.gown_lo = glo < ghi ? glo : 0; L.gown_n = glo < ghi ? ghi - glo : 0;
      }
    }
    // sibling groups for k_factor_wide (levels on the wide-block path): consecutive blocks of this rank's run with the same
    // last parent (= the same chain; device order keeps siblings and their rows contiguous), at most WG_MAXB blocks and
    // WG_MAXN columns per group
    L.wide_first = (int)h->wgrps.size(); L.wide_count = 0; L.wide_maxN = 0;
    // measured at config #4 (577^2 x 3 outcomes): the leaf level 32.7 -> 28.2 ms, the 75-column reference level 13.9 -> 15.1 ms
    // (two blocks per group: more passes than staging saved), levels with fewer groups than CUs lose parallelism -- so only
    // big non-reference levels take it (SPAMTREE_WIDE=2 forces it on every eligible level: tests)
    if (L.bigmfma && !L.lchain && h->wide_on && !h->limited && (h->wide_on == 2 || (!L.isref && L.count >= 2 * h->sm_count))) {   // L.count, not the rank's share: the two kernels round
      // differently, and a level must take the same one on every rank of every world size (bit-identical sharded runs)
      int k = L.own_lo;
      const int kend = L.own_lo + L.own_n;
      while (k < kend) {
        const int b0 = h->lvl_list[L.first + k];
        const Blk &B0 = h->blks[b0];
        const int lastp = B0.nanc ? h->anc_idx[B0.anc_ptr + B0.nanc - 1] : -1;
        WideGrp Gw; Gw.first = k - L.own_lo; Gw.count = 1;
        int Ncols = B0.m;
        while (k + Gw.count < kend && Gw.count < WG_MAXB) {
          const int b1 = h->lvl_list[L.first + k + Gw.count];
          const Blk &B1 = h->blks[b1];
          const int lp1 = B1.nanc ? h->anc_idx[B1.anc_ptr + B1.nanc - 1] : -2;
          if (lp1 != lastp || B1.nanc != B0.nanc || B1.isref != B0.isref || b1 != b0 + Gw.count || Ncols + B1.m > WG_MAXN ||
              B1.row0 != B0.row0 + Ncols) break;
          Ncols += B1.m; ++Gw.count;
        }
        L.wide_maxN = std::max(L.wide_maxN, Ncols);
        h->wgrps.push_back(Gw);
        ++L.wide_count;
        k += Gw.count;
      }
      int ldS = L.maxP + 24;
      while ((ldS & 1) || ((ldS >> 1) & 1) == 0) ++ldS;
      L.bm_ldS = ldS;
      const size_t work = std::max((size_t)17 * ldS + 16 * 16 * WG_JT, (size_t)2 * L.maxM * L.maxM + 64);
      L.lds_wide = ((size_t)3 * (L.maxP + L.wide_maxN) + 2 * (size_t)L.wide_maxN + work) * 8 + (size_t)((L.maxP + L.wide_maxN + 1) & ~1) * 4 + 64;
      if (L.lds_wide > h->lds_limit) { h->wgrps.resize(L.wide_first); L.wide_count = 0; }
    }
    // slabs for k_factor_lchain: sibling groups (consecutive blocks of this rank's run with the same last parent, contiguous
    // rows AND panels, one row stride) cut into runs of <= 4 column tiles, as equal as possible (9 tiles -> 3 + 3 + 3)
    L.lc_first = (int)h->lcslabs.size(); L.lc_count = 0;
    if (L.lchain) {
      int k = L.own_lo;
      const int kend = L.own_lo + L.own_n;
      while (k < kend) {
        const int b0 = h->lvl_list[L.first + k];
        const Blk &B0 = h->blks[b0];
        const int lastp = h->anc_idx[B0.anc_ptr + B0.nanc - 1];
        int cnt = 1, Ncols = B0.m;
        while (k + cnt < kend && cnt < 16) {
          const int b1 = h->lvl_list[L.first + k + cnt];
          const Blk &B1 = h->blks[b1];
          if (b1 != b0 + cnt || B1.nanc != B0.nanc || h->anc_idx[B1.anc_ptr + B1.nanc - 1] != lastp || B1.P != B0.P || B1.ld != B0.ld ||
              B1.row0 != B0.row0 + Ncols || B1.panel_off != B0.panel_off + (long long)Ncols * B0.ld) break;
          Ncols += B1.m; ++cnt;
        }
        const int JT = (Ncols + 15) / 16, nsl = (JT + 3) / 4, tps = (JT + nsl - 1) / nsl;
        for (int s0 = 0; s0 < Ncols; s0 += 16 * tps) {
          LcSlab S;
          S.row0 = B0.row0 + s0; S.pan0 = B0.panel_off + (long long)s0 * B0.ld; S.blk0 = b0;
          S.ncol = std::min(16 * tps, Ncols - s0); S.ld = B0.ld; S.pad = 0;
          h->lcslabs.push_back(S);
          ++L.lc_count;
        }
        k += cnt;
      }
    }
    // quads for k_factor_quad: runs of up to quad_nu column groups of one rank that share their ancestor chain
    // (reference levels) or the chain without its last ancestor (leaf levels: cousins)
    L.quad_first = (int)h->quads.size(); L.quad_count = 0; L.q_nkx = 0; L.qown_lo = 0; L.qown_n = 0;
    if (L.fast && L.maxP > 0 && L.maxP <= 200 && L.maxMa <= 32) {
      bool mixed = false;
      int qlo = INT_MAX, qhi = 0;
      // pass 0 ignores ownership: its quad count decides eligibility, so that every rank of every world size takes
      // the same kernel for a level (results are then bit-identical across world sizes); pass 1 builds this rank's quads
      int nq_any = 0;
      // units per workgroup on THIS rank: a sharded level with few owned groups takes smaller quads, so that its
      // workgroups still cover the CUs (a workgroup of 2 / 1 units lives about 0.72 / 0.5 as long as one of 4; results do
      // not depend on the grouping: every unit's arithmetic is its own)
      int nu_max = h->quad_nu;
      {
        int owned = 0;
        for (int k2 = 0; k2 < L.grp_count; ++k2) {
          const Grp &Gq = h->grps[L.grp_first + k2];
          if (g < h->cut || h->blk_owner[Gq.blk0] == h->rank) ++owned;
        }
        double best = 1e300;
        const int cand[3] = {4, 2, 1};
        const double tl[3] = {1.0, 0.72, 0.5};
        for (int c = 0; c < 3; ++c) {
          if (cand[c] > h->quad_nu) continue;
          const double rounds = std::ceil((double)std::max(owned, 1) / (double)(cand[c] * h->sm_count));
          if (rounds * tl[c] < best - 1e-9) { best = rounds * tl[c]; nu_max = cand[c]; }
        }
        const char *e = getenv("SPAMTREE_QUAD_UNITS");   // tests: force the units per workgroup (1, 2 or 4)
        if (e && atoi(e) >= 1 && atoi(e) <= h->quad_nu) nu_max = atoi(e);
      }
      for (int pass = 0; pass < 2; ++pass) {
        int k = 0;
        while (k < L.grp_count) {
          const Grp &G0 = h->grps[L.grp_first + k];
          const Blk &B0 = h->blks[G0.blk0];
          const int J = B0.nanc, Jc = B0.isref ? J : std::max(J - 1, 0);
          if ((B0.isref != 0) != (L.isref != 0)) mixed = true;
          Quad Qd;
          Qd.g0 = k; Qd.nu = 1; Qd.Jc = Jc; Qd.Pc = 0;
          for (int t = 0; t < Jc; ++t) Qd.Pc += h->blks[h->anc_idx[B0.anc_ptr + t]].m;
          while (Qd.nu < (pass == 0 ? h->quad_nu : nu_max) && k + Qd.nu < L.grp_count) {
            const Grp &G1 = h->grps[L.grp_first + k + Qd.nu];
            const Blk &B1 = h->blks[G1.blk0];
            if (B1.nanc != J || B1.isref != B0.isref) break;
            if (pass == 1 && h->blk_owner[G1.blk0] != h->blk_owner[G0.blk0]) break;
            bool same = true;
            for (int t = 0; t < Jc && same; ++t) same = h->anc_idx[B1.anc_ptr + t] == h->anc_idx[B0.anc_ptr + t];
            if (!same) break;
            ++Qd.nu;
          }
          if (pass == 0) ++nq_any;
          else {
            const bool mine = g < h->cut || h->blk_owner[G0.blk0] == h->rank;
            if (mine) { qlo = std::min(qlo, L.quad_count); qhi = std::max(qhi, L.quad_count + 1); }
            h->quads.push_back(Qd);
            L.quad_count++;
          }
          k += Qd.nu;
        }
      }
      L.qown_lo = qlo < qhi ? qlo : 0; L.qown_n = qlo < qhi ? qhi - qlo : 0;
      const int need = (L.maxP + 3) / 4;
      L.q_nkx = need <= 32 ? 32 : (need <= 38 ? 38 : (need <= 44 ? 44 : 50));
      const int ldS = quad_lds_stride(L.q_nkx);   // the kernel's compile-time row stride (>= maxP + 24)
      L.q_ldS = ldS;
      L.lds_quad = ((size_t)h->quad_nu * 16 * ldS + ldS + (L.isref ? (size_t)h->quad_nu * 512 : (size_t)2 * h->quad_nu * QUAD_LEAF_KH * 64)) * 8;   // arena, zero row, V exchange / covariance scratch
      int min_groups = 2 * h->sm_count;   // smaller levels do not fill the chip with quads: k_factor_mfma's 4x more workgroups win
      { const char *e = getenv("SPAMTREE_QUAD_MIN"); if (e) min_groups = atoi(e); }
      if (L.grp_count < 2 * nq_any || mixed || L.grp_count < min_groups) L.q_nkx = 0;   // mostly singletons: nothing to share
      if (L.isref && L.q_nkx == 50) L.q_nkx = 0;                   // that instantiation spills registers: k_factor_mfma is faster
    }
  }
  for (int i = 0; i < nb; ++i) {
    const Blk &B = h->blks[i];
    const bool mine = (h->blk_owner[i] == h->rank) || (h->blk_owner[i] < 0 && h->rank == 0);
    blkmask[i] = mine ? 1 : 0;
    if (mine) for (int r2 = 0; r2 < B.m; ++r2) rowmask[B.row0 + r2] = 1;
    if (B.nobs > 0 && (h->blk_owner[i] < 0 || h->blk_owner[i] == h->rank)) h->own_obs_list.push_back(i);
  }
  if (h->cut < n_actual) {
    const LevelInfo &L = h->levels[h->cut];
    long long lo = -1, hi = -1;
    for (int k = 0; k < L.count; ++k) {
      const Blk &B = h->blks[h->lvl_list[L.first + k]];
      if (lo < 0) lo = B.acc_off;
      hi = B.acc_off + B.acc_len;
      if (h->blk_owner[h->lvl_list[L.first + k]] != h->rank && B.acc_len > 0) h->top_zero.push_back({B.acc_off, (long long)B.acc_len});
    }
    h->top_off = std::max(0LL, lo); h->top_len = hi > lo ? hi - lo : 0;
  }
  for (int i = 0; i < nb; ++i) {
    if (h->blks[i].nobs > 0) h->all_obs_list.push_back(i);
    else {
      if (h->blks[i].nanc == 0) return fail_create(h, ST_ERR_TOPOLOGY, "prediction block without parents");
      h->pred_list.push_back(i);
    }
  }
  geometry(h->pred_info, h->pred_list, true);

  // ---- row data in device order
  std::vector<double> cx(n), cy(n), y(n), X((size_t)n * pb->p);
  std::vector<int> mv(n);
  std::vector<unsigned char> obs(n);
  h->n_obs_q.assign(pb->q, 0);
  for (long long i = 0; i < n; ++i) {
    const long long r = h->dev2model[i];
    cx[i] = pb->coords[r]; cy[i] = pb->coords[n + r];
    const long long v = pb->mv_id[r] - 1;
    if (v < 0 || v >= pb->q) return fail_create(h, ST_ERR_USAGE, "mv_id out of range");
    mv[i] = (int)v;
    const bool ok = std::isfinite(pb->y[r]);
    obs[i] = ok ? 1 : 0;
    y[i] = ok ? pb->y[r] : 0.0;                                  // spamtree_model.cpp:146
    if (ok) { h->n_obs_q[v]++; h->n_obs++; }
    for (int j = 0; j < pb->p; ++j) X[(size_t)j * n + i] = pb->X[(size_t)j * n + r];
  }
  // Q3 partner rows (spamtree_model.cpp:1375): the t-th available row is paired with w[t]
  std::vector<long long> partner(n);
  {
    std::vector<long long> rank_av(n, -1);
    long long t = 0;
    for (long long r = 0; r < n; ++r)
      if (std::isfinite(pb->y[r])) rank_av[r] = t++;
    for (long long i = 0; i < n; ++i) {
      const long long r = h->dev2model[i];
      partner[i] = (h->quirks && rank_av[r] >= 0) ? h->model2dev[rank_av[r]] : i;
    }
  }
  // XtX(j) over observed rows of outcome j (:151-155)
  h->xtx.assign((size_t)pb->p * pb->p * pb->q, 0.0);
  for (long long r = 0; r < n; ++r) {
    if (!std::isfinite(pb->y[r])) continue;
    const int v = (int)(pb->mv_id[r] - 1);
    for (int a = 0; a < pb->p; ++a)
      for (int b2 = 0; b2 < pb->p; ++b2)
        h->xtx[(size_t)v * pb->p * pb->p + (size_t)b2 * pb->p + a] += pb->X[(size_t)a * n + r] * pb->X[(size_t)b2 * n + r];
  }

#define CCHK(call)                                                                                            \
  do {                                                                                                        \
    hipError_t e_ = (call);                                                                                   \
    if (e_ != hipSuccess) return fail_create(h, ST_ERR_HIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
  } while (0)
  CCHK(hipStreamCreate(&h->stream));
  CCHK(hipHostMalloc((void **)&h->pin, 64 * sizeof(double), hipHostMallocDefault));
  CCHK(h->d_cx.upload(cx)); CCHK(h->d_cy.upload(cy)); CCHK(h->d_y.upload(y)); CCHK(h->d_X.upload(X));
  CCHK(h->d_mv.upload(mv)); CCHK(h->d_obs.upload(obs)); CCHK(h->d_partner.upload(partner));
  CCHK(h->d_dev2model.upload(h->dev2model));
  {
    // device copy: acc_len = where, inside a child's record, the part FOR this block starts (after the block's own
    // ancestors in the full tree; at 0 when every block has a single parent)
    std::vector<Blk> db = h->blks;
    if (h->limited) for (Blk &B : db) B.acc_len = 0;
    CCHK(h->d_blks.upload(db));
    if (h->limited) { std::vector<int> t = h->twin_list; if (t.empty()) t.push_back(0); CCHK(h->d_twin.upload(t)); }
  }
  { std::vector<int> a = h->anc_idx; if (a.empty()) a.push_back(0); CCHK(h->d_anc.upload(a)); }
  { std::vector<int> a = h->dch_idx; if (a.empty()) a.push_back(0); CCHK(h->d_dch.upload(a)); }
  CCHK(h->d_lvl.upload(h->lvl_list));
  { std::vector<Grp> g = h->grps; if (g.empty()) g.push_back(Grp{0, 0, 0, 0, 0}); CCHK(h->d_grps.upload(g)); }
  {
    // group descriptors: the flattened metadata of every column group (layout: GdHead / gd_unpack)
    int stride = 8;
    for (const Grp &G : h->grps) {
      const Blk &B0 = h->blks[G.blk0];
      stride = std::max(stride, 8 + 4 * B0.nanc + 3 * G.nblk + std::min(B0.ndch, 64));
    }
    stride = (stride + 1) & ~1;
    if (stride > GD_MAXW) return fail_create(h, ST_ERR_UNSUPPORTED, "group descriptor too long");
    h->gd_stride = stride;
    h->gdesc.assign(std::max<size_t>(1, h->grps.size()) * (size_t)stride, 0);
    auto pack = [](long long lo, long long hi) { return (lo & 0xffffffffLL) | (hi << 32); };
    for (size_t g = 0; g < h->grps.size(); ++g) {
      const Grp &G = h->grps[g];
      const Blk &B0 = h->blks[G.blk0];
      long long *w = h->gdesc.data() + g * (size_t)stride;
      const int nch = std::min(B0.ndch, 64);
      w[0] = G.row0; w[1] = B0.acc_off; w[2] = pack(G.M, G.P); w[3] = pack(B0.nanc, G.nblk); w[4] = pack(B0.isref, B0.level);
      w[5] = pack(nch, h->limited ? 0 : B0.acc_len); w[6] = pack(G.blk0, 0);
      long long ao = 0, aoff = 0;
      for (int t = 0; t < B0.nanc; ++t) {
        const Blk &Ba = h->blks[h->anc_idx[B0.anc_ptr + t]];
        long long *a = w + 8 + 4 * t;
        a[0] = pack(Ba.m, ao); a[1] = Ba.row0; a[2] = Ba.chain_off; a[3] = aoff;
        ao += Ba.m; aoff += (long long)Ba.m * Ba.m + Ba.m;
      }
      w[7] = aoff;
      for (int b = 0; b < G.nblk; ++b) {
        const Blk &Bb = h->blks[G.blk0 + b];
        long long *q = w + 8 + 4 * B0.nanc + 3 * b;
        q[0] = Bb.panel_off; q[1] = Bb.row0; q[2] = Bb.ld;
      }
      for (int c = 0; c < nch; ++c) w[8 + 4 * B0.nanc + 3 * G.nblk + c] = h->blks[h->dch_idx[B0.dch_ptr + c]].acc_off;
    }
    CCHK(h->d_gdesc.upload(h->gdesc));
  }
  { std::vector<Quad> g = h->quads; if (g.empty()) g.push_back(Quad{0, 0, 0, 0}); CCHK(h->d_quads.upload(g)); }
  { std::vector<WideGrp> g = h->wgrps; if (g.empty()) g.push_back(WideGrp{0, 0}); CCHK(h->d_wgrps.upload(g)); }
  if (!h->lcslabs.empty()) { CCHK(h->d_lcslabs.upload(h->lcslabs)); CCHK(h->d_lcrow.alloc((size_t)2 * h->n_all)); }
  {
    std::vector<long long> s0off((size_t)(nb > 0 ? nb : 1), -1);
    size_t tot = 0;
    for (int g = 0; g < n_actual; ++g) {
      const LevelInfo &L = h->levels[g];
      if (!L.big_sample || !L.isref) continue;
      for (int k = 0; k < L.count; ++k) {
        const int b = h->lvl_list[L.first + k];
        s0off[b] = (long long)tot;
        tot += (size_t)h->blks[b].m * h->blks[b].m;
      }
    }
    CCHK(h->d_s0off.upload(s0off));
    CCHK(h->d_s0.alloc(std::max(tot, (size_t)1)));
  }
  { std::vector<int> a = h->pred_list; if (a.empty()) a.push_back(0); CCHK(h->d_pred.upload(a)); }
  CCHK(h->d_allobs.upload(h->all_obs_list));
  { std::vector<int> a = h->own_obs_list; if (a.empty()) a.push_back(0); CCHK(h->d_ownobs.upload(a)); }
  for (int g = 0; g < n_actual; ++g) {
    const LevelInfo &L = h->levels[g];
    if (L.fast) for (int k = 0; k < L.gown_n; ++k) h->own_grp_list.push_back(L.grp_first + L.gown_lo + k);
  }
  for (int b : h->own_obs_list) if (!h->levels[h->blks[b].level].fast) h->own_obs_slow.push_back(b);
  { std::vector<int> a = h->own_grp_list; if (a.empty()) a.push_back(0); CCHK(h->d_owngrp.upload(a)); }
  { std::vector<int> a = h->own_obs_slow; if (a.empty()) a.push_back(0); CCHK(h->d_ownslow.upload(a)); }
  CCHK(h->d_rowmask.upload(rowmask)); CCHK(h->d_blkmask.upload(blkmask));
  CCHK(h->d_comm.alloc((size_t)2 * nb + 64));
  {
    // all-gather of w: every rank's owned rows (blocks below the cut, prediction blocks included), in device order; the
    // replicated top is sampled identically everywhere and does not travel
    std::vector<std::vector<int>> rows_of(h->world);
    for (int i = 0; i < nb; ++i) {
      const int o = h->blk_owner[i];
      if (o < 0) continue;
      const Blk &B = h->blks[i];
      for (int r2 = 0; r2 < B.m; ++r2) rows_of[o].push_back((int)(B.row0 + r2));
    }
    size_t mx = 0;
    for (auto &v : rows_of) mx = std::max(mx, v.size());
    h->gather_cnt = (int)mx + 1;   // last slot: the rank's failure word
    std::vector<int> gi((size_t)h->world * h->gather_cnt, -1);
    for (int r = 0; r < h->world; ++r)
      for (size_t i2 = 0; i2 < rows_of[r].size(); ++i2) gi[(size_t)r * h->gather_cnt + i2] = rows_of[r][i2];
    CCHK(h->d_gidx.upload(gi));
    CCHK(h->d_gather.alloc(gi.size()));
    CCHK(h->d_gerr.alloc(64));
  }
  CCHK(h->d_w.alloc(n)); CCHK(h->d_xb.alloc(n)); CCHK(h->d_z.alloc(n)); CCHK(h->d_tmp_n.alloc(n + 64));
  CCHK(hipMemset(h->d_w.p, 0, n * sizeof(double)));
  CCHK(hipMemset(h->d_xb.p, 0, n * sizeof(double)));
  CCHK(hipMemset(h->d_z.p, 0, n * sizeof(double)));
  CCHK(h->d_B.alloc((size_t)pb->p * pb->q));
  CCHK(h->d_tsq.alloc(QMAX));
  for (int s = 0; s < 2; ++s) {
    CCHK(h->d_panels[s].alloc(h->panel_total + 2));   // + 2: a 16-byte LDS-DMA piece may read one double past a row
    CCHK(hipMemset(h->d_panels[s].p, 0, (h->panel_total + 2) * sizeof(double)));
    CCHK(h->d_logdet[s].alloc(nb)); CCHK(h->d_loglik[s].alloc(nb));
    CCHK(hipMemset(h->d_logdet[s].p, 0, nb * sizeof(double)));
    CCHK(hipMemset(h->d_loglik[s].p, 0, nb * sizeof(double)));
  }
  CCHK(h->d_acc.alloc(std::max<size_t>(h->acc_total, 1)));
  CCHK(hipMemset(h->d_acc.p, 0, std::max<size_t>(h->acc_total, 1) * sizeof(double)));
  CCHK(h->d_scalars.alloc(8 + 2 * SUM2_WG));
  CCHK(h->d_err.alloc(2));
  CCHK(h->d_partial.alloc((size_t)STATS_WG * (pb->p * pb->q + pb->q)));
  CCHK(h->d_stats.alloc((size_t)pb->p * pb->q + pb->q));
  // scratch for the generic kernels: a bounded number of resident workgroups, each with its own slice
  {
    size_t need = 0;
    auto upd = [&](const LevelInfo &L) {
      if (L.big_factor || L.bigmfma) need = std::max(need, scratch_factor_doubles(L.maxP, L.maxM, L.maxMa));
      if (L.wide_count > 0) need = std::max(need, (size_t)2 * L.maxP * L.wide_maxN + (size_t)L.maxMa * L.wide_maxN);
      if (L.big_sample) need = std::max(need, (size_t)L.maxM * L.maxM);
    };
    for (auto &L : h->levels) upd(L);
    if (!h->pred_list.empty()) upd(h->pred_info);
    if (need > 0) {
      h->scratch_wgs = h->sm_count * 4;
      h->scratch_stride = (long long)((need + 15) & ~(size_t)15);
      CCHK(h->d_scratch.alloc((size_t)h->scratch_wgs * h->scratch_stride));
    }
  }
  // opt in to > 64 KiB dynamic LDS
  const int lim = (int)h->lds_limit;
  (void)hipFuncSetAttribute((const void *)k_factor<false, MODE_FACTOR>, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
  (void)hipFuncSetAttribute((const void *)k_factor<true, MODE_FACTOR>, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
  (void)hipFuncSetAttribute((const void *)k_factor<false, MODE_PREDICT>, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
  (void)hipFuncSetAttribute((const void *)k_factor<true, MODE_PREDICT>, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
  (void)hipFuncSetAttribute((const void *)k_sample<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
  (void)hipFuncSetAttribute((const void *)k_sample<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
  (void)hipFuncSetAttribute((const void *)k_sample<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
  (void)hipFuncSetAttribute((const void *)k_loglik, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
  (void)hipFuncSetAttribute((const void *)k_factor_mfma, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
  (void)hipFuncSetAttribute((const void *)k_factor_wide<WG_JT>, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
  {
    // k_factor_lchain: static + dynamic LDS must fit one CU's 160 KB, else the level stays on the older kernels
    hipFuncAttributes fa;
    size_t st96 = 16 * 1024, st136 = 16 * 1024;
    bool ok96 = false, ok136 = false;   // a build whose unrolling failed keeps K in scratch memory (localSizeBytes > 0): never use that
    if (hipFuncGetAttributes(&fa, (const void *)k_factor_lchain<96>) == hipSuccess) { st96 = fa.sharedSizeBytes; ok96 = fa.localSizeBytes == 0; }
    if (hipFuncGetAttributes(&fa, (const void *)k_factor_lchain<136>) == hipSuccess) { st136 = fa.sharedSizeBytes; ok136 = fa.localSizeBytes == 0; }
    for (auto &L : h->levels) {
      if (!L.lchain) continue;
      const size_t need = lc_dyn_doubles(L.lchain) * 8 + (L.lchain == 96 ? st96 : st136);
      if (need > 160 * 1024 || !(L.lchain == 96 ? ok96 : ok136)) L.lchain = 0;
    }
    (void)hipFuncSetAttribute((const void *)k_factor_lchain<96>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lc_dyn_doubles(96) * 8));
    (void)hipFuncSetAttribute((const void *)k_factor_lchain<136>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(lc_dyn_doubles(136) * 8));
  }
  (void)hipFuncSetAttribute((const void *)k_factor_bigmfma<5, 3, 24>, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
  (void)hipFuncSetAttribute((const void *)k_factor_bigmfma<3, 5, 34>, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
  (void)hipFuncSetAttribute((const void *)k_factor_bigmfma<4, 5, 34>, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
  (void)hipFuncSetAttribute((const void *)k_sample_mfma, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
  (void)hipFuncSetAttribute((const void *)k_sample_wave, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
  (void)hipFuncSetAttribute((const void *)k_marginal_invchol, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
  {
    // phase A kernel for the column-group levels: 3 (default) = k_factor_quad where a level is eligible (big enough,
    // chains <= 200 rows, LDS fits) and k_factor_mfma elsewhere; 1 = k_factor_mfma everywhere
    const char *e = getenv("SPAMTREE_FACTOR_KERNEL");
    h->factor_gen = (e && e[0] == '1') ? 1 : 3;
    { const char *e2 = getenv("SPAMTREE_SAMPLE_LEAN"); h->sample_lean = (e2 && e2[0] == '0') ? 0 : 1; }
    { const char *e2 = getenv("SPAMTREE_SPLIT_GRAM"); h->split_gram = (e2 && e2[0] == '0') ? 0 : ((e2 && e2[0] == '2') ? 2 : 1); }
    { const char *e2 = getenv("SPAMTREE_SAMPLE_WAVE"); h->sample_wave = (e2 && e2[0] == '0') ? 0 : ((e2 && e2[0] == '2') ? 2 : 1); }   // 2: every eligible level (tests)
  }
  {
    // k_factor_quad: static + dynamic LDS must fit; levels that do not fit (or are too small to fill the chip) keep k_factor_mfma
    const void *fq = (const void *)k_factor_quad<4, 50, 13, false, true>;
    hipFuncAttributes fa;
    size_t stat = 24 * 1024;
    if (hipFuncGetAttributes(&fa, fq) == hipSuccess) stat = fa.sharedSizeBytes;
    {
      const void *fr = (const void *)k_factor_quad<4, 50, 13, true, true>;
      if (hipFuncGetAttributes(&fa, fr) == hipSuccess) stat = std::max(stat, (size_t)fa.sharedSizeBytes);
      const void *ft = (const void *)k_factor_quad<4, 50, 13, true, false>;
      if (hipFuncGetAttributes(&fa, ft) == hipSuccess) stat = std::max(stat, (size_t)fa.sharedSizeBytes);
    }
    for (auto &L : h->levels) {
      if (L.q_nkx == 0) continue;
      if (L.lds_quad + stat > 160 * 1024) L.q_nkx = 0;
    }
#define QATTR(NU_, NKX_, NKT_)                                                                                                       \
  (void)hipFuncSetAttribute((const void *)k_factor_quad<NU_, NKX_, NKT_, true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - (int)stat); \
  (void)hipFuncSetAttribute((const void *)k_factor_quad<NU_, NKX_, NKT_, true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - (int)stat); \
  (void)hipFuncSetAttribute((const void *)k_factor_quad<NU_, NKX_, NKT_, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - (int)stat)
    QATTR(4, 50, 13); QATTR(4, 44, 11); QATTR(4, 38, 10); QATTR(4, 32, 8);
#undef QATTR
  }
  {
    // top levels that st_factor_begin may run ahead: the leading levels on k_factor_mfma (no global scratch), when every
    // level of the tree is on the column-group path (the generic kernels share one scratch arena between phases)
    h->g_top = 0;
    bool all_fast = !h->limited && !h->force_generic;
    for (int g = 0; g < n_actual; ++g) all_fast = all_fast && h->levels[g].fast;
    if (all_fast) {
      while (h->g_top < n_actual && !(h->factor_gen == 3 && h->levels[h->g_top].q_nkx > 0)) ++h->g_top;
      if (h->g_top >= n_actual) h->g_top = 0;   // nothing would be left for the main stream to hide it under
    }
    // the top levels are a fixed cost (0.19 ms at n = 1e6: a quarter of a rank's phase A on 8 GPUs, 40 % of phase A at
    // n = 1e5); at n = 1e6 on one GPU the sweep fills the chip and the gain is 1.5 % (SPAMTREE_ASYNC_TOP=0 turns it off)
    const char *e = getenv("SPAMTREE_ASYNC_TOP");
    h->async_top = h->g_top > 0 && !(e && e[0] == '0');
    std::vector<int> tl;
    for (int b : h->own_obs_list) if (h->blks[b].level < h->g_top) tl.push_back(b);
    h->n_toplist = (int)tl.size();
    if (tl.empty()) tl.push_back(0);
    CCHK(h->d_toplist.upload(tl));
    CCHK(h->d_err2.alloc(2));
    if (h->async_top) {
      CCHK(hipStreamCreateWithFlags(&h->stream2, hipStreamNonBlocking));
      CCHK(hipEventCreateWithFlags(&h->ev_top, hipEventDisableTiming));
      CCHK(hipEventCreateWithFlags(&h->ev_main, hipEventDisableTiming));
      CCHK(hipEventCreateWithFlags(&h->ev_stats, hipEventDisableTiming));
    }
  }
  (void)hipGetLastError();
#undef CCHK
  h->prof_level_ms.assign(2 * n_actual, 0.0);
  h->prof_level_n.assign(2 * n_actual, 0);
  *out = h;
  return ST_OK;
}

// ---- simple state accessors ---------------------------------------------------------------------------------
static int upload_rows(st_handle h, const double *src, double *dst) {
  std::vector<double> tmp(h->n_all);
  for (long long i = 0; i < h->n_all; ++i) tmp[i] = src[h->dev2model[i]];
  HCHK(h, hipMemcpyAsync(dst, tmp.data(), h->n_all * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HCHK(h, hipStreamSynchronize(h->stream));
  return ST_OK;
}
static int download_rows(st_handle h, const double *src, double *dst) {
  std::vector<double> tmp(h->n_all);
  HCHK(h, hipMemcpyAsync(tmp.data(), src, h->n_all * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HCHK(h, hipStreamSynchronize(h->stream));
  for (long long i = 0; i < h->n_all; ++i) dst[h->dev2model[i]] = tmp[i];
  return ST_OK;
}

extern "C" int st_set_w(st_handle h, const double *w) {
  if (!h || !w) return ST_ERR_USAGE;
  invalidate_stats(h);
  HCHK(h, hipSetDevice(h->device));
  return upload_rows(h, w, h->d_w.p);
}
extern "C" int st_get_w(st_handle h, double *w) {
  if (!h || !w) return ST_ERR_USAGE;
  HCHK(h, hipSetDevice(h->device));
  return download_rows(h, h->d_w.p, w);
}
extern "C" int st_get_xb(st_handle h, double *xb) {
  if (!h || !xb) return ST_ERR_USAGE;
  HCHK(h, hipSetDevice(h->device));
  return download_rows(h, h->d_xb.p, xb);
}
extern "C" int st_set_beta(st_handle h, const double *Bcoeff) {
  if (!h || !Bcoeff) return ST_ERR_USAGE;
  invalidate_stats(h);
  HCHK(h, hipSetDevice(h->device));
  HCHK(h, hipMemcpyAsync(h->d_B.p, Bcoeff, (size_t)h->p * h->q * sizeof(double), hipMemcpyHostToDevice, h->stream));
  {
    ProfScope ps(h, 4);
    const int grid = (int)((h->n_all + NT - 1) / NT);
    hipLaunchKernelGGL(k_xb, dim3(grid), dim3(NT), 0, h->stream, h->d_X.p, h->d_mv.p, h->d_B.p, h->n_all, h->p, h->d_xb.p);
  }
  HCHK(h, hipGetLastError());
  HCHK(h, hipStreamSynchronize(h->stream));
  return ST_OK;
}
extern "C" int st_set_tausq_inv(st_handle h, const double *t) {
  if (!h || !t) return ST_ERR_USAGE;
  for (int j = 0; j < h->q; ++j) h->tausq_inv[j] = t[j];
  HCHK(h, hipSetDevice(h->device));
  HCHK(h, hipMemcpyAsync(h->d_tsq.p, h->tausq_inv, QMAX * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HCHK(h, hipStreamSynchronize(h->stream));
  return ST_OK;
}
// readers of a slot's arena while st_factor_begin's launches may still be writing it on the second stream: order the
// main stream behind them (top_pending stays set: the next st_factor still picks the result up)
static int settle_top(st_handle h) {
  if (h->top_pending) { HCHK(h, hipSetDevice(h->device)); HCHK(h, hipStreamWaitEvent(h->stream, h->ev_top, 0)); }
  return ST_OK;
}
extern "C" int st_swap(st_handle h) {
  if (!h) return ST_ERR_USAGE;
  if (h->top_pending) {   // the proposal's top levels are being written into the arena that would become the accepted slot
    h->err = "st_swap between st_factor_begin and the st_factor / st_factor_local that picks its result up";
    return ST_ERR_USAGE;
  }
  std::swap(h->slot_map[0], h->slot_map[1]);
  std::swap(h->theta[0], h->theta[1]);
  h->gram_valid = false;
  return ST_OK;
}
extern "C" int st_synchronize(st_handle h) {
  if (!h) return ST_ERR_USAGE;
  HCHK(h, hipSetDevice(h->device));
  HCHK(h, hipStreamSynchronize(h->stream));
  return ST_OK;
}
extern "C" void *st_stream(st_handle h) { return h ? (void *)h->stream : nullptr; }
// run every kernel on the caller's stream (e.g. the stream RCCL collectives are enqueued on): no host synchronisation
// is then needed between the library's kernels and the exchanges
extern "C" int st_set_stream(st_handle h, void *stream) {
  if (!h) return ST_ERR_USAGE;
  HCHK(h, hipSetDevice(h->device));
  HCHK(h, hipStreamSynchronize(h->stream));
  prof_harvest(h);
  if (h->stream && !h->ext_stream) (void)hipStreamDestroy(h->stream);
  h->stream = (hipStream_t)stream;
  h->ext_stream = true;
  return ST_OK;
}
extern "C" int st_shard_info(st_handle h, int32_t *rank, int32_t *world, int32_t *cut_level, int64_t *owned_blocks, int64_t *owned_rows) {
  if (!h) return ST_ERR_USAGE;
  if (rank) *rank = h->rank;
  if (world) *world = h->world;
  if (cut_level) *cut_level = h->cut;
  long long ob = 0, orow = 0;
  for (size_t i = 0; i < h->blks.size(); ++i) if (h->blk_owner[i] == h->rank) { ++ob; orow += h->blks[i].m; }
  if (owned_blocks) *owned_blocks = ob;
  if (owned_rows) *owned_rows = orow;
  return ST_OK;
}

// CovarianceParams::transform (covariance_functions.cpp:34-75) + vec_to_symmat (:77-92)
static int make_covpar(st_handle h, const double *theta, int ntheta, CovPar *cp) {
  const int q = h->q, ncb = q > 2 ? 3 : 1, npars = 3 * q + ncb, k = q * (q - 1) / 2;
  if (ntheta != npars + k) { h->err = "theta has the wrong length"; return ST_ERR_USAGE; }
  std::memset(cp, 0, sizeof(*cp));
  cp->q = q; cp->ncb = ncb;
  for (int j = 0; j < q; ++j) { cp->ai1[j] = theta[j]; cp->ai2[j] = theta[q + j]; cp->phi[j] = theta[2 * q + j]; }
  for (int j = 0; j < ncb; ++j) cp->tmv[j] = theta[3 * q + j];
  int ix = 0;
  for (int j = 0; j < q; ++j)
    for (int i = j + 1; i < q; ++i) { cp->D[i * q + j] = theta[npars + ix]; cp->D[j * q + i] = theta[npars + ix]; ++ix; }
  finish_covpar(cp);
  return ST_OK;
}

template <bool BIG, int MODE>
static void launch_factor(st_handle h, const LevelInfo &L, FactorArgs &A, const CovPar &cp, hipStream_t st = nullptr) {
  int grid = A.nlist;
  if (BIG) {
    grid = std::min(grid, h->scratch_wgs);
    A.scratch = h->d_scratch.p; A.scratch_stride = h->scratch_stride;
  }
  hipLaunchKernelGGL((k_factor<BIG, MODE>), dim3(grid), dim3(NT), L.lds_factor, st ? st : h->stream, A, cp);
}

static int reduce_loglik(st_handle h, int phys, double *loglik) {
  {
    ProfScope ps(h, 3);
    launch_sum2(h->stream, h->d_logdet[phys].p, h->d_loglik[phys].p, (int)h->n_blocks, h->d_scalars.p + 8, h->d_scalars.p);
  }
  HCHK(h, hipGetLastError());
  double s[2];
  HCHK(h, hipMemcpyAsync(s, h->d_scalars.p, 2 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HCHK(h, hipStreamSynchronize(h->stream));
  *loglik = s[0] + s[1];   // loglik_w = logdetCi + sum(loglik_w_comps)  (:987-988, :815-816)
  return ST_OK;
}

static int read_err(st_handle h, int *code) {
  int e[2];
  HCHK(h, hipMemcpyAsync(e, h->d_err.p, 2 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HCHK(h, hipStreamSynchronize(h->stream));
  *code = (e[0] == INT_MAX) ? 0 : (e[0] & 15);
  return ST_OK;
}
static int reset_err(st_handle h) {
  const int init[2] = {INT_MAX, 0};
  HCHK(h, hipMemcpyAsync(h->d_err.p, init, 2 * sizeof(int), hipMemcpyHostToDevice, h->stream));
  return ST_OK;
}

// sum-with-zeros exchange buffers: every entry is contributed by exactly one rank (replicated blocks by rank 0), so
// an all-reduce(sum) reproduces the single-GPU arrays bit for bit, for any number of ranks and any reduction order
__global__ void k_pack_comps(const double *logdet, const double *loglik, const unsigned char *mask, int nb, const int *err, int rank,
                             int world, double *buf) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nb) {
    buf[i] = mask[i] ? logdet[i] : 0.0;
    buf[nb + i] = mask[i] ? loglik[i] : 0.0;
  }
  if (i < world) buf[2 * nb + i] = (i == rank && err[0] != INT_MAX) ? (double)err[0] : 0.0;
}
// all-gather form of the exchange of w: a rank's slice of the gather buffer = its owned rows + its failure word
__global__ void k_gather_pack(const double *w, const int *idx, int cnt, const int *err, double *out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < cnt - 1) { const int r = idx[i]; out[i] = r >= 0 ? w[r] : 0.0; }
  else if (i == cnt - 1) out[i] = err[0] != INT_MAX ? (double)err[0] : 0.0;
}
__global__ void k_gather_unpack(const double *recv, const int *idx, int cnt, long long total, double *w, double *errs) {
  const long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (j >= total) return;
  const int i = (int)(j % cnt);
  if (i == cnt - 1) { errs[j / cnt] = recv[j]; return; }
  const int r = idx[j];
  if (r >= 0) w[r] = recv[j];
}
__global__ void k_pack_w(const double *w, const unsigned char *mask, long long n, const int *err, int rank, int world, double *buf) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) buf[i] = mask[i] ? w[i] : 0.0;
  if (i < world) buf[n + i] = (i == rank && err[0] != INT_MAX) ? (double)err[0] : 0.0;
}

// levels [g_lo, g_hi); `st` / `errflag`: the launch stream and failure word (st_factor_begin: the second stream, d_err2)
static int factor_launch(st_handle h, int phys, const CovPar &cp, int g_lo = 0, int g_hi = INT_MAX, hipStream_t st = nullptr, int *errflag = nullptr) {
  g_hi = std::min(g_hi, h->n_actual_groups);
  if (!st) st = h->stream;
  if (!errflag) errflag = h->d_err.p;
  int n_launch = 0;
  for (int g = g_lo; g < g_hi; ++g) n_launch += ((h->levels[g].fast ? h->levels[g].gown_n : h->levels[g].own_n) != 0);
  ProfScope phase(h, 0, -2, n_launch, st);   // profile mode 2: the phase's launches between ONE pair of events (mean launch = total / launches)
  if (g_lo == 0 && h->limited && !h->twin_list.empty()) {
    MarginalArgs M;
    M.blks = h->d_blks.p; M.list = h->d_twin.p; M.nlist = (int)h->twin_list.size(); M.cx = h->d_cx.p; M.cy = h->d_cy.p; M.mv = h->d_mv.p;
    M.panels = h->d_panels[phys].p; M.errflag = errflag; M.maxM = h->twin_maxM;
    const size_t lds = (size_t)2 * h->twin_maxM * h->twin_maxM * sizeof(double);
    hipLaunchKernelGGL(k_marginal_invchol, dim3(std::min(M.nlist, 8 * h->sm_count)), dim3(NT), lds, st, M, cp);
  }
  for (int g = g_lo; g < g_hi; ++g) {
    const LevelInfo &L = h->levels[g];
    if ((L.fast ? L.gown_n : L.own_n) == 0) continue;
    FactorArgs A;
    std::memset(&A, 0, sizeof(A));
    A.blks = h->d_blks.p; A.anc_idx = h->d_anc.p; A.list = h->d_lvl.p + L.first + L.own_lo; A.nlist = L.own_n;
    A.cx = h->d_cx.p; A.cy = h->d_cy.p; A.mv = h->d_mv.p; A.w_in = h->d_w.p; A.w_out = nullptr; A.z = nullptr;
    A.panels = h->d_panels[phys].p; A.logdet_c = h->d_logdet[phys].p; A.loglik_c = h->d_loglik[phys].p;
    A.errflag = errflag; A.maxP = L.maxP; A.maxM = L.maxM; A.maxMa = L.maxMa; A.SR = L.big_factor ? 4 : 8;
    {
      ProfScope ps(h, 0, g, 1, st);
      if (L.fast && h->factor_gen == 3 && L.q_nkx > 0) {
        QuadArgs F;
        std::memset(&F, 0, sizeof(F));
        F.blks = h->d_blks.p; F.anc_idx = h->d_anc.p; F.grps = h->d_grps.p + L.grp_first;
        F.quads = h->d_quads.p + L.quad_first + L.qown_lo; F.nquad = L.qown_n;
        F.cx = h->d_cx.p; F.cy = h->d_cy.p; F.mv = h->d_mv.p; F.w = h->d_w.p; F.panels = h->d_panels[phys].p;
        F.logdet_c = h->d_logdet[phys].p; F.loglik_c = h->d_loglik[phys].p; F.errflag = errflag; F.ldS = L.q_ldS;
        F.gdesc = h->d_gdesc.p + (size_t)L.grp_first * h->gd_stride; F.gd_stride = h->gd_stride;
        F.wave_chol = L.maxM <= 27 ? 1 : 0;
#define QLAUNCH(NU_, NKX_, NKT_)                                                                                               \
  do {                                                                                                                         \
    if (L.isref && F.wave_chol) hipLaunchKernelGGL((k_factor_quad<NU_, NKX_, NKT_, true, true>), dim3(L.qown_n), dim3(128 * NU_), L.lds_quad, st, F, cp); \
    else if (L.isref) hipLaunchKernelGGL((k_factor_quad<NU_, NKX_, NKT_, true, false>), dim3(L.qown_n), dim3(128 * NU_), L.lds_quad, st, F, cp); \
    else hipLaunchKernelGGL((k_factor_quad<NU_, NKX_, NKT_, false, true>), dim3(L.qown_n), dim3(128 * NU_), L.lds_quad, st, F, cp);         \
  } while (0)
        if (L.q_nkx == 32) QLAUNCH(4, 32, 8); else if (L.q_nkx == 38) QLAUNCH(4, 38, 10); else if (L.q_nkx == 44) QLAUNCH(4, 44, 11); else QLAUNCH(4, 50, 13);
#undef QLAUNCH
      } else if (L.fast) {
        FastArgs F;
        std::memset(&F, 0, sizeof(F));
        F.blks = h->d_blks.p; F.anc_idx = h->d_anc.p; F.grps = h->d_grps.p + L.grp_first + L.gown_lo; F.ngrp = L.gown_n;
        F.cx = h->d_cx.p; F.cy = h->d_cy.p; F.mv = h->d_mv.p; F.w = h->d_w.p; F.panels = h->d_panels[phys].p;
        F.logdet_c = h->d_logdet[phys].p; F.loglik_c = h->d_loglik[phys].p; F.errflag = errflag;
        F.Pm4 = L.Pm4; F.ldKV = L.ldKV; F.ldS = L.ldS; F.SRm = L.SRm; F.stage_dbl = L.stage_dbl;
        F.gdesc = h->d_gdesc.p + (size_t)(L.grp_first + L.gown_lo) * h->gd_stride; F.gd_stride = h->gd_stride;
        hipLaunchKernelGGL(k_factor_mfma, dim3(L.gown_n), dim3(NT), L.lds_fast, st, F, cp);
      } else if (L.bigmfma && h->factor_gen == 3 && L.lchain) {
        LcArgs C;
        std::memset(&C, 0, sizeof(C));
        C.blks = h->d_blks.p; C.anc_idx = h->d_anc.p; C.slabs = h->d_lcslabs.p + L.lc_first; C.nslab = L.lc_count;
        C.cx = h->d_cx.p; C.cy = h->d_cy.p; C.mv = h->d_mv.p; C.w_in = h->d_w.p; C.panels = h->d_panels[phys].p;
        C.rowtmp = h->d_lcrow.p; C.n_rows = h->n_all; C.errflag = errflag;
        if (L.lchain == 96) hipLaunchKernelGGL((k_factor_lchain<96>), dim3(L.lc_count), dim3(LC_NT), lc_dyn_doubles(96) * 8, st, C, cp);
        else hipLaunchKernelGGL((k_factor_lchain<136>), dim3(L.lc_count), dim3(LC_NT), lc_dyn_doubles(136) * 8, st, C, cp);
        hipLaunchKernelGGL(k_lchain_scalars, dim3((A.nlist + 255) / 256), dim3(256), 0, st, h->d_blks.p, A.list, A.nlist, h->d_lcrow.p, h->n_all,
                           h->d_logdet[phys].p, h->d_loglik[phys].p);
      } else if (L.bigmfma && h->factor_gen == 3 && L.wide_count > 0) {
        WideArgs W;
        std::memset(&W, 0, sizeof(W));
        W.blks = h->d_blks.p; W.anc_idx = h->d_anc.p; W.list = h->d_lvl.p + L.first + L.own_lo; W.groups = h->d_wgrps.p + L.wide_first;
        W.ngroups = L.wide_count; W.cx = h->d_cx.p; W.cy = h->d_cy.p; W.mv = h->d_mv.p; W.w_in = h->d_w.p; W.panels = h->d_panels[phys].p;
        W.logdet_c = h->d_logdet[phys].p; W.loglik_c = h->d_loglik[phys].p; W.errflag = errflag; W.scratch = h->d_scratch.p;
        W.scratch_stride = h->scratch_stride; W.maxP = L.maxP; W.maxN = L.wide_maxN; W.maxM = L.maxM; W.maxMa = L.maxMa; W.ldS = L.bm_ldS;
        hipLaunchKernelGGL((k_factor_wide<WG_JT>), dim3(std::min(L.wide_count, h->sm_count)), dim3(WG_NT), L.lds_wide, st, W, cp);
      } else if (L.bigmfma && h->factor_gen == 3) {
        A.scratch = h->d_scratch.p; A.scratch_stride = h->scratch_stride; A.SR = L.bm_ldS;
        if (L.maxM <= 48) hipLaunchKernelGGL((k_factor_bigmfma<3, 5, 34>), dim3(std::min(A.nlist, h->sm_count)), dim3(BM_NT), L.lds_bigmfma, st, A, cp);
        else if (L.maxM <= 64) hipLaunchKernelGGL((k_factor_bigmfma<4, 5, 34>), dim3(std::min(A.nlist, h->sm_count)), dim3(BM_NT), L.lds_bigmfma, st, A, cp);
        else hipLaunchKernelGGL((k_factor_bigmfma<5, 3, 24>), dim3(std::min(A.nlist, h->sm_count)), dim3(BM_NT), L.lds_bigmfma, st, A, cp);
      } else if (L.big_factor) launch_factor<true, MODE_FACTOR>(h, L, A, cp, st);
      else launch_factor<false, MODE_FACTOR>(h, L, A, cp, st);
    }
    HCHK(h, hipGetLastError());
  }
  return ST_OK;
}

__global__ void k_merge_err(int *err, const int *err2) {
  if (threadIdx.x == 0 && blockIdx.x == 0 && err2[0] < err[0]) err[0] = err2[0];
}

// Phase A of the top levels ahead of time, on the second stream: call before the sweep with the theta st_factor /
// st_factor_local will be given next for the same slot.  A no-op when the tree does not qualify (or SPAMTREE_ASYNC_TOP=0).
extern "C" int st_factor_ahead_levels(st_handle h) { return (h && h->async_top) ? h->g_top : 0; }
extern "C" int st_factor_begin(st_handle h, int slot, const double *theta, int ntheta) {
  if (!h || !theta || slot < 0 || slot > 1) return ST_ERR_USAGE;
  if (!h->async_top) return ST_OK;
  HCHK(h, hipSetDevice(h->device));
  CovPar cp;
  int rc = make_covpar(h, theta, ntheta, &cp);
  if (rc) return rc;
  if (h->top_pending) HCHK(h, hipStreamWaitEvent(h->stream2, h->ev_top, 0));
  HCHK(h, hipEventRecord(h->ev_main, h->stream));          // everything issued so far (the previous iteration) comes first
  HCHK(h, hipStreamWaitEvent(h->stream2, h->ev_main, 0));
  const int init[2] = {INT_MAX, 0};
  HCHK(h, hipMemcpyAsync(h->d_err2.p, init, 2 * sizeof(int), hipMemcpyHostToDevice, h->stream2));
  const int phys = h->slot_map[slot];
  h->prof_suspend = true;   // not timed: the launches overlap the sweep on another stream
  rc = factor_launch(h, phys, cp, 0, h->g_top, h->stream2, h->d_err2.p);
  h->prof_suspend = false;
  if (rc) return rc;
  HCHK(h, hipEventRecord(h->ev_top, h->stream2));
  h->top_pending = true; h->top_phys = phys; h->top_theta.assign(theta, theta + ntheta);
  return ST_OK;
}
static int fix_top_comps(st_handle h, int phys);
static int run_stats(st_handle h, hipStream_t st);
// The beta / tausq statistics of the iteration need the sweep's w and the current XB only: when a proposal is about to be
// factorised they start on the second stream and run under phase A (the driver asks for them after the Metropolis step).
static int stats_begin(st_handle h) {
  if (!h->stream2 || h->stats_valid) return ST_OK;
  HCHK(h, hipEventRecord(h->ev_main, h->stream));           // w of the sweep is final at this point of the main stream
  HCHK(h, hipStreamWaitEvent(h->stream2, h->ev_main, 0));
  int rc = run_stats(h, h->stream2);
  if (rc) return rc;
  // the results travel to pinned host memory on the same stream: when the driver asks (after the Metropolis step) they are
  // there, and fetching them costs an event query instead of a copy + a synchronisation of the main stream
  const size_t nq = (size_t)h->p * h->q + h->q;
  h->stats_prefetched = nq <= 40;
  if (h->stats_prefetched) HCHK(h, hipMemcpyAsync(h->pin + 20, h->d_stats.p, nq * sizeof(double), hipMemcpyDeviceToHost, h->stream2));
  HCHK(h, hipEventRecord(h->ev_stats, h->stream2));
  h->stats_on_stream2 = true;
  return ST_OK;
}

extern "C" int st_factor_local(st_handle h, int slot, const double *theta, int ntheta) {
  if (!h || !theta || slot < 0 || slot > 1) return ST_ERR_USAGE;
  HCHK(h, hipSetDevice(h->device));
  CovPar cp;
  int rc = make_covpar(h, theta, ntheta, &cp);
  if (rc) return rc;
  h->theta[slot].assign(theta, theta + ntheta);
  if (slot == 0) h->gram_valid = false;
  rc = reset_err(h);
  if (rc) return rc;
  const int phys = h->slot_map[slot];
  if (slot == 1) { rc = stats_begin(h); if (rc) return rc; }
  bool reuse = false;
  if (h->top_pending) {
    HCHK(h, hipStreamWaitEvent(h->stream, h->ev_top, 0));   // the top levels are done (or at least out of the way) before anything below
    reuse = h->top_phys == phys && (int)h->top_theta.size() == ntheta && std::equal(theta, theta + ntheta, h->top_theta.begin());
    h->top_pending = false;
  }
  if (!reuse) return factor_launch(h, phys, cp);
  rc = factor_launch(h, phys, cp, h->g_top);
  if (rc) return rc;
  rc = fix_top_comps(h, phys);
  if (rc) return rc;
  hipLaunchKernelGGL(k_merge_err, dim3(1), dim3(64), 0, h->stream, h->d_err.p, h->d_err2.p);
  HCHK(h, hipGetLastError());
  return ST_OK;
}

extern "C" int st_mg_pack_comps(st_handle h, int slot, void **dev_ptr, int64_t *len) {
  if (!h || slot < 0 || slot > 1) return ST_ERR_USAGE;
  HCHK(h, hipSetDevice(h->device));
  { const int rc0 = settle_top(h); if (rc0) return rc0; }
  const int phys = h->slot_map[slot], nb = (int)h->n_blocks;
  {
    ProfScope ps(h, 3);
    hipLaunchKernelGGL(k_pack_comps, dim3((std::max(nb, h->world) + NT - 1) / NT), dim3(NT), 0, h->stream, h->d_logdet[phys].p,
                       h->d_loglik[phys].p, h->d_blkmask.p, nb, h->d_err.p, h->rank, h->world, h->d_comm.p);
  }
  HCHK(h, hipGetLastError());
  if (dev_ptr) *dev_ptr = h->d_comm.p;
  if (len) *len = 2 * (int64_t)nb + h->world;
  return ST_OK;
}

// after the exchange of the packed components: deterministic sum + failure code agreed by all ranks
extern "C" int st_mg_finish(st_handle h, double *loglik) {
  if (!h) return ST_ERR_USAGE;
  HCHK(h, hipSetDevice(h->device));
  const int nb = (int)h->n_blocks;
  {
    ProfScope ps(h, 3);
    launch_sum2(h->stream, h->d_comm.p, h->d_comm.p + nb, nb, h->d_scalars.p + 8, h->d_scalars.p);
  }
  HCHK(h, hipGetLastError());
  double s2[2], errw[64];
  HCHK(h, hipMemcpyAsync(s2, h->d_scalars.p, 2 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HCHK(h, hipMemcpyAsync(errw, h->d_comm.p + 2 * (size_t)nb, h->world * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HCHK(h, hipStreamSynchronize(h->stream));
  int best = INT_MAX;
  for (int r = 0; r < h->world; ++r) if (errw[r] > 0.5) best = std::min(best, (int)errw[r]);
  if (best != INT_MAX) return best & 15;
  if (loglik) *loglik = s2[0] + s2[1];   // loglik_w = logdetCi + sum(loglik_w_comps)  (:987-988, :815-816)
  return ST_OK;
}

#define NCHK(h, call)                                                                                        \
  do {                                                                                                        \
    ncclResult_t r_ = (call);                                                                                 \
    if (r_ != ncclSuccess) { (h)->err = std::string(#call) + ": " + ncclGetErrorString(r_); return ST_ERR_HIP; } \
  } while (0)

// pack -> RCCL all-reduce(sum) on the launch stream -> deterministic finish (native path of the multi-GPU protocol)
static int exchange_comps_and_finish(st_handle h, int slot, double *loglik) {
  void *ptr = nullptr;
  int64_t len = 0;
  int rc = st_mg_pack_comps(h, slot, &ptr, &len);
  if (rc) return rc;
  {
    ProfScope ps(h, 7);
    NCHK(h, ncclAllReduce(ptr, ptr, (size_t)len, ncclDouble, ncclSum, h->comm, h->stream));
  }
  return st_mg_finish(h, loglik);
}

extern "C" int st_comm_unique_id(void *out, int32_t cap) {
  if (!out || cap < (int32_t)sizeof(ncclUniqueId)) return ST_ERR_USAGE;
  ncclUniqueId id;
  if (ncclGetUniqueId(&id) != ncclSuccess) return ST_ERR_HIP;
  std::memcpy(out, &id, sizeof(id));
  return (int)sizeof(id);
}
extern "C" int st_comm_init(st_handle h, const void *unique_id) {
  if (!h || !unique_id) return ST_ERR_USAGE;
  HCHK(h, hipSetDevice(h->device));
  ncclUniqueId id;
  std::memcpy(&id, unique_id, sizeof(id));
  NCHK(h, ncclCommInitRank(&h->comm, h->world, id, h->rank));
  return ST_OK;
}

extern "C" int st_factor(st_handle h, int slot, const double *theta, int ntheta, double *loglik) {
  if (!h) return ST_ERR_USAGE;
  if (h->world > 1 || h->comm) {   // an attached communicator selects the exchange protocol even with one rank (tests)
    if (!h->comm) { h->err = "world > 1: call st_comm_init first, or use st_factor_local / st_mg_pack_comps / (all-reduce) / st_mg_finish"; return ST_ERR_USAGE; }
    int rc = st_factor_local(h, slot, theta, ntheta);
    if (rc) return rc;
    return exchange_comps_and_finish(h, slot, loglik);
  }
  int rc = st_factor_local(h, slot, theta, ntheta);
  if (rc) return rc;
  // the failure word and the two sums come back in ONE synchronisation (the sums are meaningless after a failure)
  {
    ProfScope ps(h, 3);
    const int phys = h->slot_map[slot];
    launch_sum2(h->stream, h->d_logdet[phys].p, h->d_loglik[phys].p, (int)h->n_blocks, h->d_scalars.p + 8, h->d_scalars.p);
  }
  HCHK(h, hipGetLastError());
  HCHK(h, hipMemcpyAsync(h->pin, h->d_scalars.p, 2 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HCHK(h, hipMemcpyAsync(h->pin + 2, h->d_err.p, 2 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  HCHK(h, hipStreamSynchronize(h->stream));
  const int e0 = ((const int *)(h->pin + 2))[0];
  if (e0 != INT_MAX) return e0 & 15;  // the reference's `return false` (:971-982); deeper levels hold unspecified values (Q5)
  if (loglik) *loglik = h->pin[0] + h->pin[1];   // loglik_w = logdetCi + sum(loglik_w_comps)  (:987-988)
  return ST_OK;
}

// st_sample_w followed by st_loglik_w(slot) with ONE synchronisation (the sweep's failure word travels with the sums).
static int gather_w_scatter(st_handle h);
extern "C" int st_sample_w_loglik(st_handle h, const double *z, uint64_t seed, uint32_t iter, int slot, double *loglik) {
  if (!h || slot < 0 || slot > 1) return ST_ERR_USAGE;
  if (h->world > 1 || h->comm) {   // an attached communicator selects the exchange protocol even with one rank (tests)
    if (!h->comm) {
      int rc = st_sample_w(h, z, seed, iter);   // reports the missing communicator
      if (rc) return rc;
      return st_loglik_w(h, slot, loglik);
    }
    // Fused exchange: the log-density of a rank's own blocks needs w of its own subtrees and of the replicated top only,
    // both current BEFORE the other ranks' rows arrive -- so phase C runs ahead of the exchange of w, and w and the
    // log-density components travel in ONE grouped RCCL call, followed by ONE host synchronisation.
    int rc = st_sample_w_local(h, z, seed, iter);
    if (rc) return rc;
    if (h->top_len > 0) {
      ProfScope ps(h, 7);
      NCHK(h, ncclAllReduce(h->d_acc.p + h->top_off, h->d_acc.p + h->top_off, (size_t)h->top_len, ncclDouble, ncclSum, h->comm, h->stream));
    }
    rc = st_sample_w_top(h);
    if (rc) return rc;
    void *pw = nullptr, *pr = nullptr, *pc = nullptr;
    int64_t lw = 0, lc = 0;
    rc = st_mg_gather_w_pack(h, &pw, &pr, &lw);   // own rows of w + this rank's failure word of the sweep
    if (rc) return rc;
    rc = st_loglik_local(h, slot);          // resets the failure word after the pack above (stream order)
    if (rc) return rc;
    rc = st_mg_pack_comps(h, slot, &pc, &lc);
    if (rc) return rc;
    {
      ProfScope ps(h, 7);
      NCHK(h, ncclGroupStart());
      NCHK(h, ncclAllGather(pw, pr, (size_t)lw, ncclDouble, h->comm, h->stream));
      NCHK(h, ncclAllReduce(pc, pc, (size_t)lc, ncclDouble, ncclSum, h->comm, h->stream));
      NCHK(h, ncclGroupEnd());
    }
    rc = gather_w_scatter(h);
    if (rc) return rc;
    invalidate_stats(h);
    const int nb = (int)h->n_blocks;
    {
      ProfScope ps(h, 3);
      launch_sum2(h->stream, h->d_comm.p, h->d_comm.p + nb, nb, h->d_scalars.p + 8, h->d_scalars.p);
    }
    HCHK(h, hipGetLastError());
    double s2[2], errw[64], errc[64];
    HCHK(h, hipMemcpyAsync(errw, h->d_gerr.p, h->world * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HCHK(h, hipMemcpyAsync(s2, h->d_scalars.p, 2 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HCHK(h, hipMemcpyAsync(errc, h->d_comm.p + 2 * (size_t)nb, h->world * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    HCHK(h, hipStreamSynchronize(h->stream));
    int best = INT_MAX;
    for (int r = 0; r < h->world; ++r) if (errw[r] > 0.5) best = std::min(best, (int)errw[r]);
    if (best != INT_MAX) return best & 15;   // 10 / 11: the reference stops with "Error at gibbs_sample_w" (:1215-1217)
    for (int r = 0; r < h->world; ++r) if (errc[r] > 0.5) best = std::min(best, (int)errc[r]);
    if (best != INT_MAX) return best & 15;
    if (loglik) *loglik = s2[0] + s2[1];
    return ST_OK;
  }
  int rc = st_sample_w_local(h, z, seed, iter);
  if (rc) return rc;
  rc = st_sample_w_top(h);
  if (rc) return rc;
  HCHK(h, hipMemcpyAsync(h->pin + 2, h->d_err.p, 2 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  rc = st_loglik_local(h, slot);   // resets the failure word after the copy above (stream order)
  if (rc) return rc;
  {
    ProfScope ps(h, 3);
    const int phys = h->slot_map[slot];
    launch_sum2(h->stream, h->d_logdet[phys].p, h->d_loglik[phys].p, (int)h->n_blocks, h->d_scalars.p + 8, h->d_scalars.p);
  }
  HCHK(h, hipGetLastError());
  HCHK(h, hipMemcpyAsync(h->pin, h->d_scalars.p, 2 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HCHK(h, hipStreamSynchronize(h->stream));
  const int e0 = ((const int *)(h->pin + 2))[0];
  if (e0 != INT_MAX) return e0 & 15;  // 10 / 11: the reference stops with "Error at gibbs_sample_w" (:1215-1217)
  if (loglik) *loglik = h->pin[0] + h->pin[1];
  return ST_OK;
}

// The same pair WITHOUT the host synchronisation: the log-density of the sweep's w is not needed on the host before the
// Metropolis step, i.e. after the proposal's factorisation -- whose own synchronisation then brings it along.  One round trip
// and one idle gap of the GPU less per iteration (the driver enqueues phase A right behind phase C).  _end returns what the
// synchronous call would have returned (failure code of the sweep / the log-density); it synchronises only if nobody has yet.
// With a communicator attached (multi-GPU) _begin simply runs the synchronous protocol and _end hands its result over.
extern "C" int st_sample_w_loglik_begin(st_handle h, const double *z, uint64_t seed, uint32_t iter, int slot) {
  if (!h || slot < 0 || slot > 1) return ST_ERR_USAGE;
  if (h->c_pending) return ST_ERR_USAGE;
  if (h->world > 1 || h->comm) {
    h->c_ll = 0.0;
    h->c_rc = st_sample_w_loglik(h, z, seed, iter, slot, &h->c_ll);
    h->c_pending = true;
    return h->c_rc < 0 ? h->c_rc : ST_OK;
  }
  int rc = st_sample_w_local(h, z, seed, iter);
  if (rc) return rc;
  rc = st_sample_w_top(h);
  if (rc) return rc;
  HCHK(h, hipMemcpyAsync(h->pin + 10, h->d_err.p, 2 * sizeof(int), hipMemcpyDeviceToHost, h->stream));
  rc = st_loglik_local(h, slot);   // resets the failure word after the copy above (stream order)
  if (rc) return rc;
  {
    ProfScope ps(h, 3);
    const int phys = h->slot_map[slot];
    launch_sum2(h->stream, h->d_logdet[phys].p, h->d_loglik[phys].p, (int)h->n_blocks, h->d_scalars.p + 8, h->d_scalars.p);
  }
  HCHK(h, hipGetLastError());
  HCHK(h, hipMemcpyAsync(h->pin + 8, h->d_scalars.p, 2 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  h->c_rc = INT_MIN;   // not known yet
  h->c_pending = true;
  return ST_OK;
}
extern "C" int st_sample_w_loglik_end(st_handle h, double *loglik) {
  if (!h || !h->c_pending) return ST_ERR_USAGE;
  h->c_pending = false;
  if (h->c_rc != INT_MIN) {   // the synchronous protocol ran in _begin
    if (h->c_rc == ST_OK && loglik) *loglik = h->c_ll;
    return h->c_rc;
  }
  HCHK(h, hipStreamSynchronize(h->stream));   // returns at once when a later call (st_factor) has synchronised already
  const int e0 = ((const int *)(h->pin + 10))[0];
  if (e0 != INT_MAX) return e0 & 15;   // 10 / 11: "Error at gibbs_sample_w"
  if (loglik) *loglik = h->pin[8] + h->pin[9];
  return ST_OK;
}

static int gen_or_upload_z(st_handle h, const double *z, uint64_t seed, uint32_t iter, unsigned stream_id, double *dst) {
  if (z) return upload_rows(h, z, dst);
  {
    ProfScope ps(h, 5);
    const int grid = (int)((h->n_all + NT - 1) / NT);
    hipLaunchKernelGGL(k_normals, dim3(grid), dim3(NT), 0, h->stream, dst, h->d_dev2model.p, h->n_all, iter, stream_id,
                       (unsigned long long)seed);
  }
  HCHK(h, hipGetLastError());
  return ST_OK;
}

static int sample_launch(st_handle h, int g_hi, int g_lo) {   // levels g_hi-1 ... g_lo
  const int phys = h->slot_map[0];
  for (int g = g_hi - 1; g >= g_lo; --g) {
    const LevelInfo &L = h->levels[g];
    if ((L.fast ? L.gown_n : L.own_n) == 0) continue;
    SampleArgs A;
    std::memset(&A, 0, sizeof(A));
    A.blks = h->d_blks.p; A.anc_idx = h->d_anc.p; A.dch_idx = h->d_dch.p; A.list = h->d_lvl.p + L.first + L.own_lo; A.nlist = L.own_n;
    A.panels = h->d_panels[phys].p; A.w = h->d_w.p; A.y = h->d_y.p; A.xb = h->d_xb.p; A.z = h->d_z.p; A.mv = h->d_mv.p;
    A.obs = h->d_obs.p; A.acc = h->d_acc.p; A.errflag = h->d_err.p; A.maxP = L.maxP; A.maxM = L.maxM; A.maxLd = L.maxLd;
    for (int j = 0; j < QMAX; ++j) A.tausq_inv[j] = h->tausq_inv[j];
    A.do_gram = (h->gram_valid && h->cache_gram) ? 0 : 1;
    A.no_fwd = h->limited ? 1 : 0;
    A.lds_sq = (L.big_sample && L.sample_sq) ? 2 : 0;
    A.s0 = h->d_s0.p; A.s0off = h->d_s0off.p;
    {
      ProfScope ps(h, 1, h->n_actual_groups + g);   // per-level slots of phase B follow those of phase A
      if (L.fast) {
        SampleFastArgs F;
        std::memset(&F, 0, sizeof(F));
        F.do_gram = A.do_gram; F.no_fwd = A.no_fwd;
        F.blks = h->d_blks.p; F.anc_idx = h->d_anc.p; F.dch_idx = h->d_dch.p; F.grps = h->d_grps.p + L.grp_first + L.gown_lo; F.ngrp = L.gown_n;
        F.panels = h->d_panels[phys].p; F.w = h->d_w.p; F.y = h->d_y.p; F.xb = h->d_xb.p; F.z = h->d_z.p; F.mv = h->d_mv.p;
        F.acc = h->d_acc.p; F.errflag = h->d_err.p; F.ldN = L.ldN; F.Mr4 = L.Mr4; F.Mrows = L.Mrows; F.maxP = L.maxP; F.av_dbl = L.av_dbl;
        F.gdesc = h->d_gdesc.p + (size_t)(L.grp_first + L.gown_lo) * h->gd_stride; F.gd_stride = h->gd_stride;
        for (int j = 0; j < QMAX; ++j) F.tausq_inv[j] = h->tausq_inv[j];
        const bool lean_ok = h->sample_lean != 0 && !(!L.isref && L.maxP > 255);
        // the theta-only Gram parts on their own (k_gram), then the lean sweep kernels: pays on big reference levels (n = 1e6,
        // level 7: 0.84 -> 0.72 ms averaged over a run's sweeps), loses on leaf levels and on smaller reference levels, where
        // the staged Gram of k_sample_mfma is cheaper (SPAMTREE_SPLIT_GRAM=2: every level; records identical either way)
        if (F.do_gram && lean_ok && h->split_gram && (h->split_gram == 2 || (L.isref && L.gown_n >= 32 * h->sm_count))) {
          hipLaunchKernelGGL(k_gram, dim3(L.gown_n), dim3(NT), 0, h->stream, F);
          F.do_gram = 0;
        }
        if (F.do_gram || !lean_ok) hipLaunchKernelGGL(k_sample_mfma, dim3(L.gown_n), dim3(NT), L.lds_sfast, h->stream, F);
        else if (!L.isref) hipLaunchKernelGGL(k_sample_leaf, dim3(L.gown_n), dim3(NT), ((size_t)L.maxP + 32 + 4 * 256 + 3 * 32) * 8, h->stream, F);
        else if (h->sample_wave && L.maxM <= 27 && (h->sample_wave == 2 || L.gown_n >= 32 * h->sm_count)) {   // one block per wave: 10 % faster on a level
          // that keeps every CU busy for many rounds (n = 1e6, level 7: 0.48 -> 0.43 ms), slower on latency-bound small levels: gd | wv | seg | tv, ev | Ri, per wave
          const size_t per = (((size_t)h->gd_stride + L.maxP + 32 + L.av_dbl + 64 + (size_t)std::max(L.maxM, 1) * CH_LD + 1) & ~(size_t)1);
          F.ldN = (int)per; F.Mrows = L.maxM;
          hipLaunchKernelGGL(k_sample_wave, dim3((L.gown_n + NT / 64 - 1) / (NT / 64)), dim3(NT), per * 8 * (NT / 64), h->stream, F);
        } else { F.av_dbl = L.av_dbl + 224; hipLaunchKernelGGL(k_sample_lean, dim3(L.gown_n), dim3(NT), L.lds_slean, h->stream, F); }
      } else {
        if (A.do_gram && L.big_sample && h->gram_big) {
          // the theta-only parts first, on the matrix cores (k_gram_big); the sweep kernel then takes its cached branch
          GramBigArgs Gb;
          std::memset(&Gb, 0, sizeof(Gb));
          Gb.blks = h->d_blks.p; Gb.anc_idx = h->d_anc.p; Gb.dch_idx = h->d_dch.p; Gb.list = A.list; Gb.nlist = A.nlist;
          Gb.panels = A.panels; Gb.acc = A.acc; Gb.s0 = A.s0; Gb.s0off = A.s0off; Gb.no_fwd = A.no_fwd;
          hipLaunchKernelGGL(k_gram_big, dim3(A.nlist), dim3(NT), 0, h->stream, Gb);
          A.do_gram = 0;
        }
        if (L.big_sample) {
          A.scratch = h->d_scratch.p; A.scratch_stride = h->scratch_stride;
          if (L.isref) hipLaunchKernelGGL((k_sample<true, false>), dim3(std::min(L.own_n, h->scratch_wgs)), dim3(NT), L.lds_sample, h->stream, A);
          else hipLaunchKernelGGL((k_sample<true, true>), dim3(std::min(L.own_n, h->scratch_wgs)), dim3(NT), L.lds_sample, h->stream, A);
        } else {
          hipLaunchKernelGGL((k_sample<false>), dim3(L.own_n), dim3(NT), L.lds_sample, h->stream, A);
        }
      }
    }
    HCHK(h, hipGetLastError());
  }
  return ST_OK;
}

// levels below the cut (this rank's subtrees); then the cut level's message records of the other ranks are zeroed so
// that an all-reduce(sum) over st_mg_top_region() completes them
extern "C" int st_sample_w_local(st_handle h, const double *z, uint64_t seed, uint32_t iter) {
  if (!h) return ST_ERR_USAGE;
  invalidate_stats(h);
  HCHK(h, hipSetDevice(h->device));
  int rc = gen_or_upload_z(h, z, seed, iter, 0u, h->d_z.p);
  if (rc) return rc;
  h->z_valid = true;
  rc = reset_err(h);
  if (rc) return rc;
  rc = sample_launch(h, h->n_actual_groups, std::min(h->cut, h->n_actual_groups));
  if (rc) return rc;
  for (auto &zr : h->top_zero) HCHK(h, hipMemsetAsync(h->d_acc.p + zr.first, 0, (size_t)zr.second * sizeof(double), h->stream));
  return ST_OK;
}
extern "C" int st_mg_top_region(st_handle h, void **dev_ptr, int64_t *len) {
  if (!h) return ST_ERR_USAGE;
  if (dev_ptr) *dev_ptr = h->d_acc.p + h->top_off;
  if (len) *len = h->top_len;
  return ST_OK;
}
extern "C" int st_sample_w_top(st_handle h) {   // the replicated levels above the cut
  if (!h) return ST_ERR_USAGE;
  invalidate_stats(h);
  HCHK(h, hipSetDevice(h->device));
  const int rc = sample_launch(h, std::min(h->cut, h->n_actual_groups), 0);
  if (rc == ST_OK) h->gram_valid = true;   // every record now carries the Gram sums of the accepted theta
  return rc;
}
extern "C" int st_mg_pack_w(st_handle h, void **dev_ptr, int64_t *len) {
  if (!h) return ST_ERR_USAGE;
  HCHK(h, hipSetDevice(h->device));
  const long long nn = std::max<long long>(h->n_all, h->world);
  hipLaunchKernelGGL(k_pack_w, dim3((unsigned)((nn + NT - 1) / NT)), dim3(NT), 0, h->stream, h->d_w.p, h->d_rowmask.p, h->n_all, h->d_err.p,
                     h->rank, h->world, h->d_tmp_n.p);
  HCHK(h, hipGetLastError());
  if (dev_ptr) *dev_ptr = h->d_tmp_n.p;
  if (len) *len = h->n_all + h->world;
  return ST_OK;
}
extern "C" int st_mg_unpack_w(st_handle h) {
  if (!h) return ST_ERR_USAGE;
  invalidate_stats(h);
  HCHK(h, hipSetDevice(h->device));
  double errw[64];
  HCHK(h, hipMemcpyAsync(h->d_w.p, h->d_tmp_n.p, (size_t)h->n_all * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  HCHK(h, hipMemcpyAsync(errw, h->d_tmp_n.p + h->n_all, h->world * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HCHK(h, hipStreamSynchronize(h->stream));
  int best = INT_MAX;
  for (int r = 0; r < h->world; ++r) if (errw[r] > 0.5) best = std::min(best, (int)errw[r]);
  return best == INT_MAX ? ST_OK : (best & 15);
}

// all-gather form (half the traffic of the all-reduce of n doubles): pack -> all-gather(recv, count per rank) -> unpack
extern "C" int st_mg_gather_w_pack(st_handle h, void **send_ptr, void **recv_ptr, int64_t *count_per_rank) {
  if (!h) return ST_ERR_USAGE;
  HCHK(h, hipSetDevice(h->device));
  const int cnt = h->gather_cnt;
  double *mine = h->d_gather.p + (size_t)h->rank * cnt;
  hipLaunchKernelGGL(k_gather_pack, dim3((cnt + NT - 1) / NT), dim3(NT), 0, h->stream, h->d_w.p, h->d_gidx.p + (size_t)h->rank * cnt, cnt,
                     h->d_err.p, mine);
  HCHK(h, hipGetLastError());
  if (send_ptr) *send_ptr = mine;
  if (recv_ptr) *recv_ptr = h->d_gather.p;
  if (count_per_rank) *count_per_rank = cnt;
  return ST_OK;
}
static int gather_w_scatter(st_handle h) {   // launches only: rows into w, failure words into d_gerr
  const long long total = (long long)h->world * h->gather_cnt;
  hipLaunchKernelGGL(k_gather_unpack, dim3((unsigned)((total + NT - 1) / NT)), dim3(NT), 0, h->stream, h->d_gather.p, h->d_gidx.p, h->gather_cnt,
                     total, h->d_w.p, h->d_gerr.p);
  HCHK(h, hipGetLastError());
  return ST_OK;
}
extern "C" int st_mg_gather_w_unpack(st_handle h) {
  if (!h) return ST_ERR_USAGE;
  invalidate_stats(h);
  HCHK(h, hipSetDevice(h->device));
  int rc = gather_w_scatter(h);
  if (rc) return rc;
  double errw[64];
  HCHK(h, hipMemcpyAsync(errw, h->d_gerr.p, h->world * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HCHK(h, hipStreamSynchronize(h->stream));
  int best = INT_MAX;
  for (int r = 0; r < h->world; ++r) if (errw[r] > 0.5) best = std::min(best, (int)errw[r]);
  return best == INT_MAX ? ST_OK : (best & 15);
}

extern "C" int st_sample_w(st_handle h, const double *z, uint64_t seed, uint32_t iter) {
  if (!h) return ST_ERR_USAGE;
  if (h->world > 1 || h->comm) {   // an attached communicator selects the exchange protocol even with one rank (tests)
    if (!h->comm) { h->err = "world > 1: call st_comm_init first, or use st_sample_w_local / st_mg_top_region / st_sample_w_top / st_mg_pack_w / st_mg_unpack_w"; return ST_ERR_USAGE; }
    int rc = st_sample_w_local(h, z, seed, iter);
    if (rc) return rc;
    if (h->top_len > 0) {
      ProfScope ps(h, 7);
      NCHK(h, ncclAllReduce(h->d_acc.p + h->top_off, h->d_acc.p + h->top_off, (size_t)h->top_len, ncclDouble, ncclSum, h->comm, h->stream));
    }
    rc = st_sample_w_top(h);
    if (rc) return rc;
    void *snd = nullptr, *rcv = nullptr;
    int64_t cnt = 0;
    rc = st_mg_gather_w_pack(h, &snd, &rcv, &cnt);
    if (rc) return rc;
    {
      ProfScope ps(h, 7);
      NCHK(h, ncclAllGather(snd, rcv, (size_t)cnt, ncclDouble, h->comm, h->stream));
    }
    return st_mg_gather_w_unpack(h);
  }
  int rc = st_sample_w_local(h, z, seed, iter);
  if (rc) return rc;
  rc = st_sample_w_top(h);
  if (rc) return rc;
  int code = 0;
  rc = read_err(h, &code);
  if (rc) return rc;
  return code;  // 10 / 11: the reference stops with "Error at gibbs_sample_w" (:1215-1217)
}

extern "C" int st_loglik_local(st_handle h, int slot) {
  if (!h || slot < 0 || slot > 1) return ST_ERR_USAGE;
  HCHK(h, hipSetDevice(h->device));
  if (slot == 1) { const int rc0 = settle_top(h); if (rc0) return rc0; }
  const int phys = h->slot_map[slot];
  int maxP = 0, maxM = 0;
  for (auto &L : h->levels) { maxP = std::max(maxP, L.maxP); maxM = std::max(maxM, L.maxM); }
  const size_t lds = lds_loglik_bytes(maxP, maxM);
  LoglikArgs A;
  A.blks = h->d_blks.p; A.anc_idx = h->d_anc.p; A.list = h->d_ownslow.p; A.nlist = (int)h->own_obs_slow.size();
  A.panels = h->d_panels[phys].p; A.w = h->d_w.p; A.loglik_c = h->d_loglik[phys].p; A.maxP = maxP; A.maxM = maxM;
  {
    ProfScope ps(h, 2);
    if (A.nlist > 0) hipLaunchKernelGGL(k_loglik, dim3(A.nlist), dim3(NT), lds, h->stream, A);
    if (!h->own_grp_list.empty()) {
      LoglikGrpArgs Gr;
      Gr.blks = h->d_blks.p; Gr.anc_idx = h->d_anc.p; Gr.grps = h->d_grps.p; Gr.list = h->d_owngrp.p; Gr.nlist = (int)h->own_grp_list.size();
      Gr.panels = h->d_panels[phys].p; Gr.w = h->d_w.p; Gr.loglik_c = h->d_loglik[phys].p; Gr.maxP = maxP;
      Gr.gdesc = h->d_gdesc.p; Gr.gd_stride = h->gd_stride;
      hipLaunchKernelGGL(k_loglik_grp, dim3(Gr.nlist), dim3(NT), (size_t)(maxP + 32) * sizeof(double), h->stream, Gr);
    }
  }
  HCHK(h, hipGetLastError());
  HCHK(h, reset_err(h) == ST_OK ? hipSuccess : hipErrorUnknown);
  return ST_OK;
}
// the quadratic forms of the top blocks with the CURRENT w (st_factor_begin ran them with the w of the sweep's start)
static int fix_top_comps(st_handle h, int phys) {
  if (h->n_toplist == 0) return ST_OK;
  int maxP = 0, maxM = 0;
  for (int g = 0; g < h->g_top; ++g) { maxP = std::max(maxP, h->levels[g].maxP); maxM = std::max(maxM, h->levels[g].maxM); }
  LoglikArgs A;
  A.blks = h->d_blks.p; A.anc_idx = h->d_anc.p; A.list = h->d_toplist.p; A.nlist = h->n_toplist;
  A.panels = h->d_panels[phys].p; A.w = h->d_w.p; A.loglik_c = h->d_loglik[phys].p; A.maxP = maxP; A.maxM = maxM;
  {
    ProfScope ps(h, 0, h->g_top > 0 ? h->g_top - 1 : 0);
    hipLaunchKernelGGL(k_loglik, dim3(A.nlist), dim3(NT), lds_loglik_bytes(maxP, maxM), h->stream, A);
  }
  HCHK(h, hipGetLastError());
  return ST_OK;
}
extern "C" int st_loglik_w(st_handle h, int slot, double *loglik) {
  if (!h) return ST_ERR_USAGE;
  if (h->world > 1 || h->comm) {   // an attached communicator selects the exchange protocol even with one rank (tests)
    if (!h->comm) { h->err = "world > 1: call st_comm_init first, or use st_loglik_local / st_mg_pack_comps / (all-reduce) / st_mg_finish"; return ST_ERR_USAGE; }
    int rc = st_loglik_local(h, slot);
    if (rc) return rc;
    return exchange_comps_and_finish(h, slot, loglik);
  }
  int rc = st_loglik_local(h, slot);
  if (rc) return rc;
  double ll = 0.0;
  rc = reduce_loglik(h, h->slot_map[slot], &ll);
  if (rc) return rc;
  if (loglik) *loglik = ll;
  return ST_OK;
}

extern "C" int st_predict(st_handle h, int theta_changed) {
  (void)theta_changed;  // H of a prediction block is rebuilt from the ancestor chain every call: same values as the cache
  if (!h) return ST_ERR_USAGE;
  if (h->pred_list.empty()) return ST_OK;
  invalidate_stats(h);
  if (!h->z_valid) { h->err = "st_predict needs the normals of a preceding st_sample_w (spamtree_model.cpp:1325)"; return ST_ERR_USAGE; }
  if (h->theta[0].empty()) { h->err = "st_predict before st_factor(slot 0)"; return ST_ERR_USAGE; }
  HCHK(h, hipSetDevice(h->device));
  CovPar cp;
  int rc = make_covpar(h, h->theta[0].data(), (int)h->theta[0].size(), &cp);
  if (rc) return rc;
  const LevelInfo &L = h->pred_info;
  FactorArgs A;
  std::memset(&A, 0, sizeof(A));
  A.blks = h->d_blks.p; A.anc_idx = h->d_anc.p; A.list = h->d_pred.p; A.nlist = (int)h->pred_list.size();
  A.cx = h->d_cx.p; A.cy = h->d_cy.p; A.mv = h->d_mv.p; A.w_in = h->d_w.p; A.w_out = h->d_w.p; A.z = h->d_z.p;
  A.panels = h->d_panels[h->slot_map[0]].p; A.logdet_c = nullptr; A.loglik_c = nullptr; A.errflag = h->d_err.p;
  A.maxP = L.maxP; A.maxM = L.maxM; A.maxMa = L.maxMa; A.SR = L.big_factor ? 4 : 8;
  {
    ProfScope ps(h, 6);
    if (L.big_factor) launch_factor<true, MODE_PREDICT>(h, L, A, cp);
    else launch_factor<false, MODE_PREDICT>(h, L, A, cp);
  }
  HCHK(h, hipGetLastError());
  HCHK(h, hipStreamSynchronize(h->stream));
  return ST_OK;
}

static int run_stats(st_handle h, hipStream_t st = nullptr) {
  const int nq = h->p * h->q + h->q;
  if (h->stats_valid) return ST_OK;   // w and XB unchanged since the last reduction: both statistics are still current
  if (!st) st = h->stream;
  {
    ProfScope ps(h, 4, -1, 1, st);
    hipLaunchKernelGGL(k_stats, dim3(STATS_WG), dim3(NT), 0, st, h->d_X.p, h->d_y.p, h->d_w.p, h->d_xb.p, h->d_mv.p, h->d_obs.p,
                       h->d_partner.p, h->n_all, h->p, h->q, h->d_partial.p);
    hipLaunchKernelGGL(k_stats_final, dim3(nq), dim3(NT), 0, st, h->d_partial.p, STATS_WG, nq, h->d_stats.p);
  }
  HCHK(h, hipGetLastError());
  h->stats_valid = true;
  return ST_OK;
}
// both statistics travel to the host together; a second request for the same (w, XB) is served from the host copy
static int fetch_stats(st_handle h) {
  if (h->stats_valid && h->host_stats_valid) return ST_OK;
  int rc = run_stats(h);
  if (rc) return rc;
  const size_t nq = (size_t)h->p * h->q + h->q;
  h->host_stats.resize(nq);
  if (h->stats_on_stream2) {   // started under phase A (stats_begin)
    if (h->stats_prefetched) {   // ... and already copied to pinned memory behind them on the second stream
      HCHK(h, hipEventSynchronize(h->ev_stats));
      HCHK(h, hipStreamWaitEvent(h->stream, h->ev_stats, 0));   // later work on the main stream stays ordered behind the reduction
      h->stats_on_stream2 = false;
      for (size_t i = 0; i < nq; ++i) h->host_stats[i] = h->pin[20 + i];
      h->host_stats_valid = true;
      return ST_OK;
    }
    HCHK(h, hipStreamWaitEvent(h->stream, h->ev_stats, 0));   // its results before the copy
    h->stats_on_stream2 = false;
  }
  HCHK(h, hipMemcpyAsync(h->host_stats.data(), h->d_stats.p, nq * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HCHK(h, hipStreamSynchronize(h->stream));
  h->host_stats_valid = true;
  return ST_OK;
}
extern "C" int st_beta_stats(st_handle h, double *xty) {
  if (!h || !xty) return ST_ERR_USAGE;
  HCHK(h, hipSetDevice(h->device));
  int rc = fetch_stats(h);
  if (rc) return rc;
  for (int i = 0; i < h->p * h->q; ++i) xty[i] = h->host_stats[i];
  return ST_OK;
}
extern "C" int st_tausq_stats(st_handle h, double *ssq, int64_t *n_obs_by_q) {
  if (!h || !ssq) return ST_ERR_USAGE;
  HCHK(h, hipSetDevice(h->device));
  int rc = fetch_stats(h);
  if (rc) return rc;
  for (int j = 0; j < h->q; ++j) ssq[j] = h->host_stats[(size_t)h->p * h->q + j];
  if (n_obs_by_q)
    for (int j = 0; j < h->q; ++j) n_obs_by_q[j] = h->n_obs_q[j];
  return ST_OK;
}
extern "C" int st_xtx(st_handle h, double *xtx) {
  if (!h || !xtx) return ST_ERR_USAGE;
  std::memcpy(xtx, h->xtx.data(), h->xtx.size() * sizeof(double));
  return ST_OK;
}

extern "C" int st_yhat(st_handle h, const double *noise, uint64_t seed, uint32_t iter, double *yhat) {
  if (!h || !yhat) return ST_ERR_USAGE;
  HCHK(h, hipSetDevice(h->device));
  int rc = gen_or_upload_z(h, noise, seed, iter, 5u, h->d_tmp_n.p);
  if (rc) return rc;
  const int grid = (int)((h->n_all + NT - 1) / NT);
  hipLaunchKernelGGL(k_yhat, dim3(grid), dim3(NT), 0, h->stream, h->d_xb.p, h->d_w.p, h->d_tmp_n.p, h->d_mv.p, h->n_all, h->d_tsq.p,
                     h->d_tmp_n.p);
  HCHK(h, hipGetLastError());
  return download_rows(h, h->d_tmp_n.p, yhat);
}

// ---- inspection ---------------------------------------------------------------------------------------------
extern "C" int st_block_dims(st_handle h, int64_t u, int64_t *m, int64_t *P, int32_t *is_ref, int32_t *n_obs) {
  if (!h || u < 0 || u >= h->n_blocks) return ST_ERR_USAGE;
  const Blk &B = h->blks[h->blk_model2dev[u]];
  if (m) *m = B.m;
  if (P) *P = B.P;
  if (is_ref) *is_ref = B.isref;
  if (n_obs) *n_obs = B.nobs;
  return ST_OK;
}
extern "C" int st_get_block(st_handle h, int slot, int64_t u, double *negRiH, double *Ri) {
  if (!h || u < 0 || u >= h->n_blocks || slot < 0 || slot > 1) return ST_ERR_USAGE;
  const Blk &B = h->blks[h->blk_model2dev[u]];
  if (B.panel_off < 0) { h->err = "block has no observations, hence no cache"; return ST_ERR_USAGE; }
  HCHK(h, hipSetDevice(h->device));
  { const int rc0 = settle_top(h); if (rc0) return rc0; }
  std::vector<double> pan((size_t)B.m * B.ld);
  HCHK(h, hipMemcpyAsync(pan.data(), h->d_panels[h->slot_map[slot]].p + B.panel_off, pan.size() * sizeof(double), hipMemcpyDeviceToHost,
                         h->stream));
  HCHK(h, hipStreamSynchronize(h->stream));
  if (negRiH)
    for (int i = 0; i < B.m; ++i)
      for (int k = 0; k < B.P; ++k) negRiH[(size_t)k * B.m + i] = pan[(size_t)i * B.ld + k];
  if (Ri) {
    if (B.isref) {
      for (int i = 0; i < B.m; ++i)
        for (int j = 0; j < B.m; ++j) Ri[(size_t)j * B.m + i] = pan[(size_t)i * B.ld + B.P + j];
    } else {
      for (int i = 0; i < B.m; ++i) Ri[i] = pan[(size_t)i * B.ld + B.P];
    }
  }
  return ST_OK;
}
extern "C" int st_get_comps(st_handle h, int slot, double *logdet_c, double *loglik_c) {
  if (!h || slot < 0 || slot > 1) return ST_ERR_USAGE;
  HCHK(h, hipSetDevice(h->device));
  { const int rc0 = settle_top(h); if (rc0) return rc0; }
  const int phys = h->slot_map[slot];
  std::vector<double> a(h->n_blocks), b(h->n_blocks);
  HCHK(h, hipMemcpyAsync(a.data(), h->d_logdet[phys].p, h->n_blocks * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HCHK(h, hipMemcpyAsync(b.data(), h->d_loglik[phys].p, h->n_blocks * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HCHK(h, hipStreamSynchronize(h->stream));
  for (long long i = 0; i < h->n_blocks; ++i) {
    const int u = h->blks[i].model_id;
    if (logdet_c) logdet_c[u] = a[i];
    if (loglik_c) loglik_c[u] = b[i];
  }
  return ST_OK;
}

// ---- measurement --------------------------------------------------------------------------------------------
extern "C" int st_algorithmic_bytes(st_handle h, double *out5, double *flops3) {
  if (!h || !out5) return ST_ERR_USAGE;
  double a = 0, b = 0, c = 0, msg = 0, fa = 0, fb = 0, fc = 0;
  for (auto &L : h->levels) { a += L.alg_bytes_A; b += L.alg_bytes_B; c += L.alg_bytes_C; msg += L.alg_bytes_msg; fa += L.flops_A; fb += L.flops_B; fc += L.flops_C; }
  out5[0] = a; out5[1] = b; out5[2] = c; out5[3] = msg; out5[4] = 40.0 * (double)h->n_obs;
  if (flops3) { flops3[0] = fa; flops3[1] = fb; flops3[2] = fc; }
  return ST_OK;
}
extern "C" int st_profile_enable(st_handle h, int enable) {
  if (!h) return ST_ERR_USAGE;
  h->prof = enable == 2 ? 2 : (enable != 0 ? 1 : 0);
  return ST_OK;
}
extern "C" int st_profile_get(st_handle h, double *ms_total, int64_t *launches) {
  if (!h) return ST_ERR_USAGE;
  HCHK(h, hipSetDevice(h->device));
  prof_harvest(h);
  for (int f = 0; f < ST_N_KERNEL_FAMILIES; ++f) {
    if (ms_total) ms_total[f] = h->prof_ms[f];
    if (launches) launches[f] = h->prof_n[f];
    h->prof_ms[f] = 0; h->prof_n[f] = 0;
  }
  return ST_OK;
}
extern "C" int st_profile_levels(st_handle h, int32_t *n_levels, double *ms_by_level, double *bytes_by_level, int32_t cap) {
  if (!h || !n_levels) return ST_ERR_USAGE;
  HCHK(h, hipSetDevice(h->device));
  prof_harvest(h);
  *n_levels = h->n_actual_groups;
  // entries [0, n_levels): phase A; when cap >= 2 n_levels, entries [n_levels, 2 n_levels): phase B (k_sample*)
  const int ng = h->n_actual_groups;
  for (int g = 0; g < 2 * ng && g < cap; ++g) {
    if (ms_by_level) ms_by_level[g] = h->prof_level_n[g] ? h->prof_level_ms[g] / (double)h->prof_level_n[g] : 0.0;  // mean per launch
    if (bytes_by_level) bytes_by_level[g] = g < ng ? h->levels[g].alg_bytes_A : h->levels[g - ng].alg_bytes_B + h->levels[g - ng].alg_bytes_msg;
    h->prof_level_ms[g] = 0.0; h->prof_level_n[g] = 0;
  }
  return ST_OK;
}

// which phase-A kernel each observed level takes and the sizes that decide it (tests prove the branch they mean to reach)
extern "C" int st_level_info(st_handle h, int32_t *n_levels, int32_t *kernel, int32_t *max_m, int32_t *max_P, int32_t *n_blocks, int32_t cap) {
  if (!h || !n_levels) return ST_ERR_USAGE;
  *n_levels = h->n_actual_groups;
  for (int g = 0; g < h->n_actual_groups && g < cap; ++g) {
    const LevelInfo &L = h->levels[g];
    int k = L.big_factor ? ST_KERNEL_GENERIC_SCRATCH : ST_KERNEL_GENERIC_LDS;
    if (L.fast && h->factor_gen == 3 && L.q_nkx > 0) k = ST_KERNEL_QUAD;
    else if (L.fast) k = ST_KERNEL_MFMA;
    else if (L.bigmfma && h->factor_gen == 3) k = L.lchain ? ST_KERNEL_LCHAIN : (L.wide_count > 0 ? ST_KERNEL_WIDE : ST_KERNEL_BIGMFMA);
    if (kernel) kernel[g] = k;
    if (max_m) max_m[g] = L.maxM;
    if (max_P) max_P[g] = L.maxP;
    if (n_blocks) n_blocks[g] = L.count;
  }
  return ST_OK;
}

// ---- CrossCovarianceAG10 (covariance_functions.cpp:301-355): dense n1 x n2 cross-covariance, column-major output.
// mv1 / mv2 are 1-based (as in R).  Like the reference it refuses a 1 x 1 Dmat ("Invalid Dmat for multivariate data").
extern "C" int st_cross_covariance_ag10(const double *coords1, const int64_t *mv1, int64_t n1, const double *coords2, const int64_t *mv2,
                                        int64_t n2, const double *ai1, const double *ai2, const double *phi_i, const double *thetamv,
                                        const double *Dmat, int32_t q, int32_t device, double *out) {
  if (!coords1 || !coords2 || !mv1 || !mv2 || !out || !Dmat || q < 2 || q > QMAX) {
    g_create_error = q < 2 ? "Invalid Dmat for multivariate data" : "st_cross_covariance_ag10: bad argument";
    return ST_ERR_USAGE;
  }
  if (hipSetDevice(device) != hipSuccess) { g_create_error = "no usable HIP device"; return ST_ERR_HIP; }
  CovPar cp;
  std::memset(&cp, 0, sizeof(cp));
  cp.q = q; cp.ncb = q > 2 ? 3 : 1;
  for (int j = 0; j < q; ++j) { cp.ai1[j] = ai1[j]; cp.ai2[j] = ai2[j]; cp.phi[j] = phi_i[j]; }
  for (int j = 0; j < cp.ncb; ++j) cp.tmv[j] = thetamv[j];
  for (int i = 0; i < q * q; ++i) cp.D[i] = Dmat[i];   // symmetric: layout irrelevant
  finish_covpar(&cp);
  std::vector<int> m1(n1), m2(n2);
  for (int64_t i = 0; i < n1; ++i) { m1[i] = (int)mv1[i] - 1; if (m1[i] < 0 || m1[i] >= q) { g_create_error = "mv1 out of range"; return ST_ERR_USAGE; } }
  for (int64_t i = 0; i < n2; ++i) { m2[i] = (int)mv2[i] - 1; if (m2[i] < 0 || m2[i] >= q) { g_create_error = "mv2 out of range"; return ST_ERR_USAGE; } }
  DevBuf<double> d1, d2, dout;
  DevBuf<int> dm1, dm2;
  int rc = ST_OK;
  auto bad = [&](hipError_t e) { if (e != hipSuccess) { g_create_error = hipGetErrorString(e); rc = ST_ERR_HIP; } return e != hipSuccess; };
  if (!bad(d1.alloc(2 * n1)) && !bad(d2.alloc(2 * n2)) && !bad(dout.alloc((size_t)n1 * n2)) && !bad(dm1.upload(m1)) && !bad(dm2.upload(m2)) &&
      !bad(hipMemcpy(d1.p, coords1, 2 * n1 * sizeof(double), hipMemcpyHostToDevice)) &&
      !bad(hipMemcpy(d2.p, coords2, 2 * n2 * sizeof(double), hipMemcpyHostToDevice))) {
    hipLaunchKernelGGL(k_cross_cov, dim3((unsigned)((n1 + NT - 1) / NT), (unsigned)n2), dim3(NT), 0, 0, d1.p, dm1.p, (long long)n1, d2.p, dm2.p,
                       (long long)n2, cp, dout.p);
    if (!bad(hipGetLastError())) bad(hipMemcpy(out, dout.p, (size_t)n1 * n2 * sizeof(double), hipMemcpyDeviceToHost));
  }
  d1.free(); d2.free(); dout.free(); dm1.free(); dm2.free();
  return rc;
}

// ---- running posterior means on device: call st_summary_accumulate on every saved iteration
extern "C" int st_summary_reset(st_handle h) {
  if (!h) return ST_ERR_USAGE;
  HCHK(h, hipSetDevice(h->device));
  if (!h->d_sum_w.p) { HCHK(h, h->d_sum_w.alloc(h->n_all)); HCHK(h, h->d_sum_yhat.alloc(h->n_all)); }
  HCHK(h, hipMemsetAsync(h->d_sum_w.p, 0, h->n_all * sizeof(double), h->stream));
  HCHK(h, hipMemsetAsync(h->d_sum_yhat.p, 0, h->n_all * sizeof(double), h->stream));
  h->n_summary = 0; h->n_draws = 0;
  return ST_OK;
}
extern "C" int st_summary_accumulate(st_handle h, uint64_t seed, uint32_t iter) {   // yhat noise: device stream 5 (spamtree_fit.cpp:384)
  if (!h) return ST_ERR_USAGE;
  if (!h->d_sum_w.p) { const int rc0 = st_summary_reset(h); if (rc0) return rc0; }
  HCHK(h, hipSetDevice(h->device));
  int rc = gen_or_upload_z(h, nullptr, seed, iter, 5u, h->d_tmp_n.p);
  if (rc) return rc;
  const int grid = (int)((h->n_all + NT - 1) / NT);
  hipLaunchKernelGGL(k_yhat, dim3(grid), dim3(NT), 0, h->stream, h->d_xb.p, h->d_w.p, h->d_tmp_n.p, h->d_mv.p, h->n_all, h->d_tsq.p, h->d_tmp_n.p);
  hipLaunchKernelGGL(k_axpy_sum, dim3(grid), dim3(NT), 0, h->stream, h->d_sum_yhat.p, h->d_tmp_n.p, h->n_all);
  hipLaunchKernelGGL(k_axpy_sum, dim3(grid), dim3(NT), 0, h->stream, h->d_sum_w.p, h->d_w.p, h->n_all);
  HCHK(h, hipGetLastError());
  if (h->n_draws < h->draws_cap) {   // keep the draw itself for the quantiles (device order, one contiguous row per draw)
    HCHK(h, hipMemcpyAsync(h->d_draws_w.p + (size_t)h->n_draws * h->n_all, h->d_w.p, h->n_all * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    HCHK(h, hipMemcpyAsync(h->d_draws_yhat.p + (size_t)h->n_draws * h->n_all, h->d_tmp_n.p, h->n_all * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    h->n_draws += 1;
  }
  h->n_summary += 1;
  return ST_OK;
}
extern "C" int st_summary_reserve(st_handle h, int64_t keep) {
  if (!h || keep < 0) return ST_ERR_USAGE;
  HCHK(h, hipSetDevice(h->device));
  HCHK(h, hipStreamSynchronize(h->stream));
  h->d_draws_w.free(); h->d_draws_yhat.free();
  h->draws_cap = 0; h->n_draws = 0;
  if (keep == 0) return ST_OK;
  if (keep > 16384) { h->err = "st_summary_reserve: at most 16384 saved draws (one row's draws are sorted in one workgroup's LDS)"; return ST_ERR_UNSUPPORTED; }
  HCHK(h, h->d_draws_w.alloc((size_t)keep * h->n_all));
  HCHK(h, h->d_draws_yhat.alloc((size_t)keep * h->n_all));
  h->draws_cap = keep;
  return ST_OK;
}
extern "C" int st_summary_quantile(st_handle h, double q, double *w_q, double *yhat_q) {
  if (!h || !(q >= 0.0 && q <= 1.0)) return ST_ERR_USAGE;
  if (h->n_draws == 0) { h->err = "no draw stored: call st_summary_reserve before the saved iterations"; return ST_ERR_USAGE; }
  HCHK(h, hipSetDevice(h->device));
  int Kpad = 2;
  while (Kpad < h->n_draws) Kpad <<= 1;
  const int R = std::max(1, std::min(8, (int)(128 * 1024 / ((size_t)Kpad * 8))));
  const size_t lds = (size_t)R * Kpad * sizeof(double);
  (void)hipFuncSetAttribute((const void *)k_qtile, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->lds_limit);
  for (int which = 0; which < 2; ++which) {
    double *dst = which == 0 ? w_q : yhat_q;
    if (!dst) continue;
    QtArgs A;
    A.draws = which == 0 ? h->d_draws_w.p : h->d_draws_yhat.p; A.n = h->n_all; A.keep = (int)h->n_draws; A.Kpad = Kpad; A.R = R; A.q = q;
    A.out = h->d_tmp_n.p;
    hipLaunchKernelGGL(k_qtile, dim3((unsigned)((h->n_all + R - 1) / R)), dim3(NT), lds, h->stream, A);
    HCHK(h, hipGetLastError());
    const int rc = download_rows(h, h->d_tmp_n.p, dst);
    if (rc) return rc;
  }
  return ST_OK;
}
extern "C" int st_summary_get(st_handle h, double *w_mean, double *yhat_mean, int64_t *n_accumulated) {
  if (!h) return ST_ERR_USAGE;
  if (n_accumulated) *n_accumulated = h->n_summary;
  if (h->n_summary == 0 || !h->d_sum_w.p) { h->err = "no iteration accumulated"; return ST_ERR_USAGE; }
  HCHK(h, hipSetDevice(h->device));
  const double inv = 1.0 / (double)h->n_summary;
  if (w_mean) { int rc = download_rows(h, h->d_sum_w.p, w_mean); if (rc) return rc; for (long long i = 0; i < h->n_all; ++i) w_mean[i] *= inv; }
  if (yhat_mean) { int rc = download_rows(h, h->d_sum_yhat.p, yhat_mean); if (rc) return rc; for (long long i = 0; i < h->n_all; ++i) yhat_mean[i] *= inv; }
  return ST_OK;
}
