// Phase A / P, generic any-size kernels (k_factor<BIG, MODE>) and the limited-tree marginal factor (k_marginal_invchol).
#pragma once
#include "st_device.hpp"

// k_factor<BIG, MODE>, k_factor_bigmfma: one block per workgroup
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
  const double *vscr;     // k_factor_ref_finish: V = Linv_pa K_pa,u as k_factor_lchain left it (row-major per block, row stride RF_LDB) ...
  const long long *voff;  // ... per block of the list: where its matrix starts ...
  const double *hvrow;    // ... per device row: (T w_pa)_j as k_factor_lchain summed it
};

#define MODE_FACTOR 0
#define MODE_PREDICT 1
// k_marginal_invchol (limited_tree): the marginal inverse Cholesky factors the children read as their chain
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

#ifdef ST_DEFS_FACTOR_GENERIC
// ---------------------------------------------------------------------------------------------------------------
// Phase A / P : per block, one workgroup.  BIG=false keeps K/V, T, the row stage and the m x m factors in LDS;
// BIG=true keeps them in a per-workgroup slice of a global scratch arena (any m, P).
// ---------------------------------------------------------------------------------------------------------------


// limited_tree (/root/reference/src/spamtree_model.cpp:901-903, 1275-1278: Kxx_inv(u) = inv_sympd(K_uu), every block has
// ONE parent, /root/reference/src/tree_dep.cpp:133-186): the chain factor the children of u work with is the block's MARGINAL
// inverse Cholesky chol(K_uu)^{-1} (m x m, row-major, Blk::chain_off), not its conditional panel.  One workgroup per block.
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

// The same for blocks of at most 27 rows, ONE BLOCK PER WAVE (round 3): the workgroup version above factorises with a
// barrier per pivot and inverts with one thread per column -- 0.7 ms for the 21 845 reference blocks of n = 1e6, as much as
// all the level kernels of a limited tree together; here a wave builds K_uu in its own LDS slice and runs the blocked
// 16 + 11 elimination of chol_blocked.hpp (9.5 k cycles, no barrier anywhere).
__global__ __launch_bounds__(NT) void k_marginal_invchol_wave(MarginalArgs A, CovPar cp) {
  extern __shared__ double lds[];   // per wave: R (32 x CH_LD) | L^{-1} (32 x CH_LD)
  __shared__ int s_failw[NT / 64];
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  double *R = lds + (size_t)wid * 64 * CH_LD, *Li = R + 32 * CH_LD;
  for (int li = blockIdx.x * (NT / 64) + wid; li < A.nlist; li += gridDim.x * (NT / 64)) {
    const Blk B = A.blks[A.list[li]];
    const int m = B.m;   // <= 27 (host)
    if (lane == 0) s_failw[wid] = 0;
    for (int e = lane; e < m * (m + 1) / 2; e += 64) {
      int i = (int)((sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
      while (i * (i + 1) / 2 > e) --i;
      while ((i + 1) * (i + 2) / 2 <= e) ++i;
      const int j = e - i * (i + 1) / 2;
      const long long ri = B.row0 + i, rj = B.row0 + j;
      R[i * CH_LD + j] = cov_entry(cp, A.cx[ri], A.cy[ri], A.mv[ri], A.cx[rj], A.cy[rj], A.mv[rj]);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    wave_chol_eliminate_blocked<11>(R, Li, m, &s_failw[wid], lane);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    double *out = A.panels + B.chain_off;
    for (int idx = lane; idx < m * m; idx += 64) {
      const int i = idx / m, j = idx - i * m;
      out[idx] = (j <= i) ? Li[i * CH_LD + j] : 0.0;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0 && s_failw[wid]) atomicMin(A.errflag, B.level * 16 + 2);   // errtype 2 (:919), reported at the block's level
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


template __global__ void k_factor<false, MODE_FACTOR>(FactorArgs, CovPar);
template __global__ void k_factor<true, MODE_FACTOR>(FactorArgs, CovPar);
template __global__ void k_factor<false, MODE_PREDICT>(FactorArgs, CovPar);
template __global__ void k_factor<true, MODE_PREDICT>(FactorArgs, CovPar);
#else   // host side: prototypes only (the kernels are compiled in their own translation unit)
__global__ void k_marginal_invchol(MarginalArgs A, CovPar cp);
__global__ void k_marginal_invchol_wave(MarginalArgs A, CovPar cp);
template <bool BIG, int MODE> __global__ void k_factor(FactorArgs A, CovPar cp);
#endif
