// Phase C, the O(n) statistics, reductions, RNG, summaries and the small pack / gather kernels of the multi-GPU protocol.
#pragma once
#include "st_device.hpp"

// k_loglik: phase C, one block per workgroup (any size)
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

// k_loglik_grp: phase C, one column group per workgroup
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

#define SUM2_WG 64
#define STATS_WG 1024
// k_qtile: posterior quantiles of the saved draws (list_mean.cpp:62-137)
struct QtArgs {
  const double *draws;
  long long n;
  int keep, Kpad, R;
  double q;
  double *out;
};

#ifdef ST_DEFS_MISC
__global__ void k_normals(double *z, const long long *dev2model, long long n, unsigned iter, unsigned stream, unsigned long long seed) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) z[i] = philox_normal((unsigned long long)dev2model[i], iter, stream, seed);
}

// ---------------------------------------------------------------------------------------------------------------
// Phase C: residual + quadratic form per block (spamtree_model.cpp:781-826)
// ---------------------------------------------------------------------------------------------------------------

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

__global__ void k_pack_comps(const double *logdet, const double *loglik, const unsigned char *mask, int nb, const int *err, int rank,
                             int world, double *buf) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < nb) {
    buf[i] = mask[i] ? logdet[i] : 0.0;
    buf[nb + i] = mask[i] ? loglik[i] : 0.0;
  }
  if (i < world) buf[2 * nb + i] = (i == rank && err[0] != INT_MAX) ? (double)err[0] : 0.0;
}

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

__global__ void k_merge_err(int *err, const int *err2) {
  if (threadIdx.x == 0 && blockIdx.x == 0 && err2[0] < err[0]) err[0] = err2[0];
}

#else   // host side: prototypes only (the kernels are compiled in their own translation unit)
__global__ void k_normals(double *z, const long long *dev2model, long long n, unsigned iter, unsigned stream, unsigned long long seed);
__global__ void k_loglik(LoglikArgs A);
__global__ void k_loglik_grp(LoglikGrpArgs A);
__global__ void k_sum2_partial(const double *a, const double *b, int n, double *partial);
__global__ void k_sum2_final(const double *partial, double *out);
__global__ void k_xb(const double *X, const int *mv, const double *B, long long n, int p, double *xb);
__global__ void k_stats(const double *X, const double *y, const double *w, const double *xb, const int *mv, const unsigned char *obs, const long long *partner, long long n, int p, int q, double *partial);
__global__ void k_stats_final(const double *partial, int nwg, int nq, double *out);
__global__ void k_yhat(const double *xb, const double *w, const double *noise, const int *mv, long long n, const double *tsq_inv_q, double *yhat);
__global__ void k_cross_cov(const double *c1, const int *mv1, long long n1, const double *c2, const int *mv2, long long n2, CovPar cp, double *out);
__global__ void k_qtile(QtArgs A);
__global__ void k_axpy_sum(double *acc, const double *x, long long n);
__global__ void k_pack_comps(const double *logdet, const double *loglik, const unsigned char *mask, int nb, const int *err, int rank, int world, double *buf);
__global__ void k_gather_pack(const double *w, const int *idx, int cnt, const int *err, double *out);
__global__ void k_gather_unpack(const double *recv, const int *idx, int cnt, long long total, double *w, double *errs);
__global__ void k_pack_w(const double *w, const unsigned char *mask, long long n, const int *err, int rank, int world, double *buf);
__global__ void k_merge_err(int *err, const int *err2);
#endif

static void launch_sum2(hipStream_t st, const double *a, const double *b, int n, double *partial, double *out) {
  hipLaunchKernelGGL(k_sum2_partial, dim3(SUM2_WG), dim3(NT), 0, st, a, b, n, partial);
  hipLaunchKernelGGL(k_sum2_final, dim3(1), dim3(64), 0, st, partial, out);
}
