// Phase B kernels (block-Gibbs sweep of w) and the Gram-part builders.
#pragma once
#include "st_device.hpp"

// k_sample<BIG, NOREF>: the any-size kernels (one block per workgroup; BIG: intermediates in a global scratch slice)
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

// k_gram_big: theta-only parts of the generic path's records + Ri' Ri on the matrix cores
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
  int mirror;     // 1: off-diagonal tiles are written to both triangles (a reader on another kernel family takes full squares); 0: the
                  // reader is k_gram_big / the generic sweep's cached branch, which take tiles it >= jt / the lower triangle only
};

// the column-group kernels: k_sample_mfma, k_gram, k_gram_direct, k_sample_lean, k_sample_wave, k_sample_leaf, k_sample_leaf_seg
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
  const long long *gdesc_all;   // k_gram_direct: descriptors of ALL groups (the children's are looked up by group id)
  // theta-only part of a reference block's posterior precision, S0 = Ri' Ri + the children's Gram parts for the block
  // (spamtree_model.cpp:912, 1044-1051 without the tausq term): cached per accepted theta like the records' Gram parts (Q4).
  // s0_mode 0: compute (as every sweep did before round 3); 1: compute and store; 2: load.  Stored and loaded values are the
  // same bits, so the draws do not depend on the mode.
  double *s0;
  const long long *s0off;       // per block: offset into s0 (row stride m) or -1
  int s0_mode;
};
#define GRAM_DIRECT_MAXCH 4     // k_gram_direct: child groups per block

#ifdef ST_DEFS_SAMPLE

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
    // Segment sums seg[t][r] = sum_j N[r][oa_t + j] w_a[j] (ancestor t's part of row r of N w_pa), ONE coalesced pass over the
    // chain part of the panel: a 32-lane half-wave per (row, ancestor), its lanes on consecutive entries, a fixed-order butterfly.
    // N w_pa is their sum over the ancestors, and the messages below take ev - seg from here -- round 2 walked the panel once for
    // N w_pa (a wave per four rows) and once more for the segments (a THREAD per (row, ancestor): 64 rows per load instruction).
    {
      const int half = lane >> 5, l32 = lane & 31, ntask = m * J;
      constexpr int SU = 4;   // (row, ancestor) pairs per half-wave and trip: their loads (up to three each) travel together
      for (int base = 2 * SU * wid; base < ntask; base += 2 * SU * (NT / 64)) {
        double a[SU];
        const double *row[SU], *wa[SU];
        int ma[SU];
#pragma unroll
        for (int u = 0; u < SU; ++u) {
          const int task = min(base + 2 * u + half, ntask - 1), r = task / J, t = task - r * J;
          ma[u] = s_am[t];
          row[u] = N + (size_t)r * ld + s_ao[t];
          wa[u] = wv + s_ao[t];
        }
        double x[SU][3];
#pragma unroll
        for (int u = 0; u < SU; ++u)
#pragma unroll
          for (int c = 0; c < 3; ++c) x[u][c] = (l32 + 32 * c < ma[u]) ? row[u][l32 + 32 * c] : 0.0;
#pragma unroll
        for (int u = 0; u < SU; ++u) {
          a[u] = 0.0;
#pragma unroll
          for (int c = 0; c < 3; ++c) a[u] += x[u][c] * ((l32 + 32 * c < ma[u]) ? wa[u][l32 + 32 * c] : 0.0);
          for (int j = l32 + 96; j < ma[u]; j += 32) a[u] += row[u][j] * wa[u][j];   // (ancestors wider than 96 rows)
        }
#pragma unroll
        for (int u = 0; u < SU; ++u) {
#pragma unroll
          for (int o = 16; o >= 1; o >>= 1) a[u] += __shfl_xor(a[u], o, 64);
          const int task = base + 2 * u + half;
          if (l32 == 0 && task < ntask) { const int r = task / J, t = task - r * J; seg[t * maxM + r] = a[u]; }
        }
      }
    }
    __syncthreads();
    for (int i = tid; i < m; i += NT) {
      double a = 0.0;
      for (int t = 0; t < J; ++t) a += seg[t * maxM + i];
      tv[i] = a;
    }
    __syncthreads();

    STAMP(0);
    if (!NOREF && B.isref) {
      const double *Ri = N + P;  // Ri[i][j] = N[i*ld + P + j]
      // Sigi_tot = Ri'Ri + sum_children Sigi_children + diag(tausq_inv)      (:1044-1051)
      const int ms = (BIG && A.lds_sq) ? ((m + 7) | 1) : m;   // row stride of S
      const long long so = (BIG && A.s0off) ? A.s0off[b] : -1;
      // The lower triangle only, entry e = i (i + 1) / 2 + j, six entries per thread and trip with every load of the trip in flight
      // (s0 or -- first sweep after an accepted theta on this path -- Ri' Ri formed here, the children's parts); the upper
      // triangle is zero, the diagonal's tausq_inv comes from `ev` (filled below, free until the draw).  Round 2 walked the full
      // square behind an `if (j <= i)`: no two entries' loads travelled together (22 dependent round trips per thread at m = 75).
      double *Sl = (BIG && A.lds_sq) ? Np : S;
      for (int idx = tid; idx < m * ms; idx += NT) Sl[idx] = 0.0;
      for (int i = tid; i < m; i += NT) ev[i] = A.tausq_inv[A.mv[B.row0 + i]];
      __syncthreads();
      {
        const int ntri = m * (m + 1) / 2;
        const bool cached = so >= 0 && !A.do_gram;
        const int nch = min(B.ndch, 16);
        for (int e0 = tid; e0 < ntri; e0 += 6 * NT) {
          double acc[6];
          int ii[6], jj[6];
#pragma unroll
          for (int u = 0; u < 6; ++u) {
            const int e = min(e0 + u * NT, ntri - 1);
            int i = (int)((sqrtf(8.0f * (float)e + 1.0f) - 1.0f) * 0.5f);
            while (i * (i + 1) / 2 > e) --i;
            while ((i + 1) * (i + 2) / 2 <= e) ++i;
            ii[u] = i; jj[u] = e - i * (i + 1) / 2;
          }
          if (cached) {   // Ri' Ri is a function of theta only
#pragma unroll
            for (int u = 0; u < 6; ++u) acc[u] = A.s0[so + ii[u] * m + jj[u]];
          } else {
#pragma unroll
            for (int u = 0; u < 6; ++u) {
              double a = 0.0;
              for (int k = ii[u]; k < m; ++k) a += Ri[(size_t)k * ld + ii[u]] * Ri[(size_t)k * ld + jj[u]];
              if (so >= 0 && e0 + u * NT < ntri) A.s0[so + ii[u] * m + jj[u]] = a;
              acc[u] = a;
            }
          }
          for (int c = 0; c < nch; ++c) {
            const double *rc = A.acc + s_choff[c] + B.acc_len;
            double x[6];
#pragma unroll
            for (int u = 0; u < 6; ++u) x[u] = rc[ii[u] * m + jj[u]];
#pragma unroll
            for (int u = 0; u < 6; ++u) acc[u] += x[u];
          }
          for (int c = 16; c < B.ndch; ++c) {
            const long long co = A.blks[A.dch_idx[B.dch_ptr + c]].acc_off;
#pragma unroll
            for (int u = 0; u < 6; ++u) acc[u] += A.acc[co + B.acc_len + ii[u] * m + jj[u]];
          }
#pragma unroll
          for (int u = 0; u < 6; ++u)
            if (e0 + u * NT < ntri) Sl[(size_t)ii[u] * ms + jj[u]] = acc[u] + (ii[u] == jj[u] ? ev[ii[u]] : 0.0);
        }
      }
      // Smu_tot = A_u' w_pa + sum_children Smu_children + tausq_inv*(y - XB)   (:1062-1077)
      // - Ri' (N w_pa), walking Ri by ROWS (thread i reads entry i of row k: coalesced).  Blocks of up to NT / 2 rows: two threads
      // per row (rows k of equal parity from the diagonal), sixteen entries in flight each, the two halves added in a fixed order
      // through `av` (a latency chain of ten batches of eight before)
      const int np = 2 * m <= NT ? 2 : 1;
      for (int it = tid; it < np * m; it += NT) {
        const int i = it % m, part = it / m;
        double acc = 0.0;
        for (int k0 = i + part; k0 < m; k0 += 16 * np) {
          double x[16];
#pragma unroll
          for (int q = 0; q < 16; ++q) x[q] = Ri[(size_t)min(k0 + q * np, m - 1) * ld + i];
#pragma unroll
          for (int q = 0; q < 16; ++q)
            if (k0 + q * np < m) acc -= x[q] * tv[k0 + q * np];
        }
        if (part == 1) av[i] = acc; else bv[i] = acc;
      }
      __syncthreads();
      for (int i = tid; i < m; i += NT) {
        double acc = bv[i] + (np == 2 ? av[i] : 0.0);
        for (int c = 0; c < B.ndch; ++c) {
          const long long co = c < 16 ? s_choff[c] : A.blks[A.dch_idx[B.dch_ptr + c]].acc_off;
          acc += A.acc[co + B.acc_len + m * m + i];
        }
        const long long r = B.row0 + i;
        acc += ev[i] * (A.y[r] - A.xb[r]);
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
      for (int i0 = 8 * wid; i0 < m; i0 += 8 * (NT / 64)) {   // ev = Ri w_u + N w_pa: a wave per row (coalesced), eight rows per trip
        double a8[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        for (int j0 = 0; j0 < m; j0 += 128) {
          double x[8][2];
#pragma unroll
          for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
              const int j = j0 + lane + 64 * c;
              x[q][c] = (i0 + q < m && j <= i0 + q) ? Ri[(size_t)(i0 + q) * ld + j] : 0.0;
            }
#pragma unroll
          for (int c = 0; c < 2; ++c) {
            const int j = j0 + lane + 64 * c;
            const double wj = j < m ? wv[P + j] : 0.0;
#pragma unroll
            for (int q = 0; q < 8; ++q) a8[q] += x[q][c] * wj;
          }
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          const double r = wave_sum(a8[q]);
          if (lane == 0 && i0 + q < m) ev[i0 + q] = tv[i0 + q] + r;
        }
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
        seg[t * maxM + r] = ev[r] - seg[t * maxM + r];
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
        for (int r0 = 0; r0 < m; r0 += 16) {
          double x[16];
#pragma unroll
          for (int rr = 0; rr < 16; ++rr) x[rr] = N[(size_t)min(r0 + rr, m - 1) * ld + k];
#pragma unroll
          for (int rr = 0; rr < 16; ++rr)
            if (r0 + rr < m) a -= x[rr] * avt[r0 + rr];
        }
        for (int c = 0; c < (A.no_fwd ? 0 : B.ndch); ++c) {
          const long long co = c < 16 ? s_choff[c] : A.blks[A.dch_idx[B.dch_ptr + c]].acc_off;   // (read once per block above)
          a += A.acc[co + s_aoff[t] + ma * ma + i];
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
      for (int r = tid; r < m; r += NT) av[r] = ev[r] - seg[t * maxM + r];
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

__global__ __launch_bounds__(NT) void k_gram_big(GramBigArgs A) {
  __shared__ int s_am[MAXJ + 1], s_ao[MAXJ + 1];
  __shared__ long long s_aoff[MAXJ + 1];
  __shared__ long long s_choff[16];
  __shared__ double s_tr[NT / 64][16 * 17];   // per wave: a tile on its way to the mirrored position
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
    int o = 0;
    long long ao = 0;
    for (int t = 0; t < J; ++t) { s_ao[t] = o; s_aoff[t] = ao; o += s_am[t]; ao += (long long)s_am[t] * s_am[t] + s_am[t]; }
    if (so >= 0) { s_am[J] = m; s_ao[J] = P; s_aoff[J] = 0; }
  }
  __syncthreads();
  const double *N = A.panels + B.panel_off;
  double *rec = A.acc + B.acc_off;
  const int ns = (m + 3) >> 2;
  const int nch = A.no_fwd ? 0 : B.ndch;
  const int JJ = J + (so >= 0 ? 1 : 0);
  // Round 3: the children's forwarded parts requested four children at a time (one round trip per tile instead of one per child),
  // the mirrored tile -- where a reader wants both triangles at all: A.mirror -- through LDS, so that its store runs along rows like
  // the tile's own (lane = column made every mirrored store 64 scattered sectors: 1.4 of the 6.8 ms this kernel took at config #4).
  // Same operands, same K order, same order of the children: bit-identical records.  (Staging an ancestor's slice of the panel
  // in LDS, so that the tiles stop re-reading it from L2, measured SLOWER: 48 KB per workgroup, two workgroups per CU.)
  int tbase = 0;   // tiles of the ancestors before t: the block's tiles are dealt over the waves as ONE sequence
  for (int t = 0; t < JJ; ++t) {
    const int ma = s_am[t], oa = s_ao[t];
    const bool isS0 = t == J;
    const int nt = (ma + 15) >> 4, ntask = nt * (nt + 1) / 2;
    double *out = isS0 ? A.s0 + so : rec + s_aoff[t];
    const int e0 = (wid - tbase % (NT / 64) + (NT / 64)) % (NT / 64);
    tbase += ntask;
    for (int e = e0; e < ntask; e += NT / 64) {
      int it = 0;
      while ((it + 1) * (it + 2) / 2 <= e) ++it;
      const int jt = e - it * (it + 1) / 2;
      const int ci = 16 * it + l15, cj = 16 * jt + l15;
      d4 c = (d4){0.0, 0.0, 0.0, 0.0};
      {
        const double *ap = N + (size_t)l4 * ld + oa + min(ci, ma - 1), *bp = N + (size_t)l4 * ld + oa + min(cj, ma - 1);
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
      }
      // C layout: entry (i = 16 it + 4 q + l4, j = 16 jt + l15); the children's parts, four children at a time, in child order
      double v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = c[q];
      if (!isS0) {
        for (int c0 = 0; c0 < nch; c0 += 4) {
          double x[4][4];
#pragma unroll
          for (int cc = 0; cc < 4; ++cc) {
            const int ch = min(c0 + cc, nch - 1);
            const long long co = ch < 16 ? s_choff[ch] : A.blks[A.dch_idx[B.dch_ptr + ch]].acc_off;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int i = 16 * it + 4 * q + l4;
              x[cc][q] = (c0 + cc < nch && i < ma && cj < ma) ? A.acc[co + s_aoff[t] + (size_t)i * ma + cj] : 0.0;
            }
          }
#pragma unroll
          for (int cc = 0; cc < 4; ++cc)
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (c0 + cc < nch) v[q] += x[cc][q];
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int i = 16 * it + 4 * q + l4;
        if (i < ma && cj < ma) out[(size_t)i * ma + cj] = v[q];
      }
      if (A.mirror && it != jt) {   // the mirrored tile (jt, it): transposed through LDS, stored along its rows
        double *tr = s_tr[wid];
#pragma unroll
        for (int q = 0; q < 4; ++q) tr[(4 * q + l4) * 17 + l15] = v[q];   // tr[i'][j'] = tile entry (i', j')
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (one wave: its LDS operations execute in order)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int jr = 16 * jt + 4 * q + l4, ic = 16 * it + l15;   // mirrored entry (jr, ic) = tile entry (ic', jr') = tr[l15][4 q + l4]
          if (jr < ma && ic < ma) out[(size_t)jr * ma + ic] = tr[l15 * 17 + 4 * q + l4];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
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
// k_gram for the LAST reference level when its children (the leaf groups) keep no Gram records at all: what a leaf group's
// record would have carried for ancestor t -- N_{c,t}' N_{c,t}, the Gram matrix of the child's own panel columns -- is formed
// HERE from the child's panel rows instead of being written by the leaf level (655 MB at n = 1e6), read back (655 MB) and
// summed.  The leaf level of a rebuild sweep is then the ordinary leaf sweep (k_sample_leaf, vectors only: 0.40 instead of
// 0.80 ms).  Same arithmetic as the two-kernel route, so the records are bit-identical: per ancestor tile one MFMA chain over the
// block's own rows, one chain per child over that child's rows (exactly what the child's k_gram did: zero-padded K-steps),
// the children's results summed in child order from 0.0 and added to the own part.  The part FOR this block itself
// (the children's N_{c,u}' N_{c,u}, which the sweep kernels read from the children's records) goes to the children's
// record slots as before.  spamtree_model.cpp:1162, 1190-1192.
__global__ __launch_bounds__(NT, 4) void k_gram_direct(SampleFastArgs A) {
  __shared__ int s_am[MAXJ], s_ao[MAXJ + 1];
  __shared__ long long s_arow[MAXJ], s_aoff[MAXJ + 1];
  __shared__ long long s_bpan[32], s_brow[32];
  __shared__ int s_bld[32];
  __shared__ long long s_coff[64];
  __shared__ long long s_gd[GD_MAXW];
  __shared__ long long s_rowoff[32];                         // panel offset of the block's own row r
  __shared__ long long s_crow[GRAM_DIRECT_MAXCH][32];        // ... of child c's row r
  __shared__ int s_cM[GRAM_DIRECT_MAXCH];

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
  const int M = B0.M, J = B0.nanc, P = B0.P, nch = B0.ndch;   // nch <= GRAM_DIRECT_MAXCH (host)
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
  if (tid >= 64 && tid < 64 + 32 * GRAM_DIRECT_MAXCH) {   // the children's rows: their descriptors sit behind the record offsets
    const int c = (tid - 64) >> 5, r = (tid - 64) & 31;
    long long off = 0;
    int Mc = 0;
    if (c < nch) {
      const long long cg = s_gd[8 + 4 * J + 3 * B0.nblk + B0.ndch + c];
      const long long *g = A.gdesc_all + (size_t)cg * A.gd_stride;
      Mc = (int)(g[2] & 0xffffffffLL);
      const int Jc = (int)(g[3] & 0xffffffffLL), nb = (int)(g[3] >> 32);
      if (r < Mc) {
        const long long ra = g[0] + r;
        const long long *q = g + 8 + 4 * Jc;
        int bi = 0;
        while (bi + 1 < nb && ra >= q[3 * (bi + 1) + 1]) ++bi;
        off = q[3 * bi] + (ra - q[3 * bi + 1]) * q[3 * bi + 2];
      }
    }
    s_crow[c][r] = off;
    if (r == 0) s_cM[c] = Mc;
  }
  __syncthreads();
  const int nsteps = (M + 3) >> 2;          // <= 8
  double *rec = A.acc + B0.acc_off;
  // a wave owns one 16-row HALF of one ancestor's Gram matrix (tiles (it, 0) and (it, 1)): the two 16-column halves of a row
  // set are loaded once per item and feed both chains -- half the operand loads of k_gram's one-tile-per-wave mapping, which
  // at 53 rows per ancestor (own + children) is what bound this kernel, and few enough registers for four workgroups per CU
  for (int u = wid; u < (J + 1) * 2; u += NT / 64) {
    const int t = u >> 1, it = u & 1;
    const bool self = t == J;                // the part for this block itself: children only, their columns [P, P + M)
    const int ma = self ? M : s_am[t], oa = self ? P : s_ao[t];
    if (it * 16 >= ma) continue;
    const bool wide = ma > 16;               // otherwise only tile (0, 0) exists
    // one row set (the block's own rows, or one child's) through the two chains
    auto chains = [&](const long long *rowoff, int Mr, d4 (&cq)[2]) __attribute__((always_inline)) {
      const int ns = (Mr + 3) >> 2;
      double b0[8], b1[8];
#pragma unroll
      for (int st = 0; st < 8; ++st) {
        const int r = 4 * st + l4;
        const bool rok = st < ns && r < Mr;
        const double *row = A.panels + rowoff[min(r, 31)] + oa + l15;
        b0[st] = rok ? row[0] : 0.0;
        b1[st] = (rok && wide) ? row[16] : 0.0;
      }
      cq[0] = (d4){0.0, 0.0, 0.0, 0.0}; cq[1] = cq[0];
#pragma unroll
      for (int st = 0; st < 8; ++st)
        if (st < ns) {
          const double ai = it ? b1[st] : b0[st];
          cq[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, b0[st], cq[0], 0, 0, 0);
          if (wide) cq[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(ai, b1[st], cq[1], 0, 0, 0);
        }
    };
    auto store = [&](double *out, const d4 (&v)[2]) __attribute__((always_inline)) {
#pragma unroll
      for (int jt = 0; jt < 2; ++jt) {
        if (jt > 0 && !wide) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int i = it * 16 + l4 + 4 * r, j = jt * 16 + l15;
          if (i < ma && j < ma) out[i * ma + j] = v[jt][r];
        }
      }
    };
    // the children first: chv = ((0 + v_0) + v_1) + ... in child order, v_c = the child's chain (+ its own empty children's sum),
    // exactly the sum k_gram forms from the children's records (entries outside the ancestor's m are never stored: no masks)
    d4 chv[2];
    chv[0] = (d4){0.0, 0.0, 0.0, 0.0}; chv[1] = chv[0];
    for (int cc = 0; cc < nch; ++cc) {
      d4 cch[2];
      chains(s_crow[cc], s_cM[cc], cch);
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) { cch[q][r] += 0.0; chv[q][r] += cch[q][r]; }
      if (self) store(A.acc + s_coff[cc] + B0.acc_len, cch);   // the child's own record slot for its parent
    }
    if (!self) {
      d4 c[2];
      chains(s_rowoff, M, c);
#pragma unroll
      for (int q = 0; q < 2; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) c[q][r] += chv[q][r];
      store(rec + s_aoff[t], c);
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
// LAT (round 3, levels that do not fill the chip: a block there is a CHAIN of dependent round trips -- descriptor, rows' data, w + Ri,
// two batches of pass 1, the cached precision, the children's vectors, four batches of pass 2: about thirteen -- and the level takes
// as long as one block): every global load that needs the descriptor only is requested at once and parked in registers -- pass 2's
// rows by the threads of waves 1-3, over the draw --, so that three round trips are left.  Same arithmetic in the same order
// (identical draws); 256 registers per thread instead of 102, which costs occupancy the small levels do not use.
template <bool LAT>
__global__ __launch_bounds__(NT, LAT ? 2 : 5) void k_sample_lean(SampleFastArgs A) {
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
  const long long so = (refgrp && A.s0off && A.s0_mode) ? A.s0off[B0.blk0] : -1;
  const bool s0_load = so >= 0 && A.s0_mode == 2;
  double xq[28], x2[28], ch2[4], s0v[3], smv[4];   // LAT: pass 1's segment, pass 2's column, its children's vectors, the cached precision
  const int k2 = tid - 64;                           // LAT: pass 2's chain column of this thread (waves 1-3; columns beyond 191: late)
  const int nch2 = A.no_fwd ? 0 : B0.ndch;
  if constexpr (LAT) {
    double t_ = 0.0, y_ = 0.0, xb_ = 0.0, z_ = 0.0, r_ = 0.0;
    int mv_ = 0;
    const bool rowthr = tid >= 32 && tid < 64 && tid - 32 < M;
    if (tid >= 32 && tid < 64) {
      const int j = tid - 32;
      int bi = 0;
      long long ro = 0;
      if (j < M) {
        const long long r = G.row0 + j;
        mv_ = A.mv[r]; y_ = A.y[r]; xb_ = A.xb[r]; z_ = A.z[r];
        const long long *gb = s_gd + 8 + 4 * J;   // per block: panel offset, first row, ld
        while (bi + 1 < G.nblk && r >= gb[3 * (bi + 1) + 1]) ++bi;
        ro = gb[3 * bi] + (r - gb[3 * bi + 1]) * gb[3 * bi + 2];
        if (!refgrp) r_ = A.panels[ro + P];
      }
      s_cb[j] = bi; s_rowoff[j] = ro;
    }
    lds_barrier();   // (LDS only: the loads above stay in flight)
    // everything else that needs the descriptor only
    double wreg = 0.0;
    int wk = -1;
    if (tid < P) {
      int t = 0;
      while (t + 1 < J && tid >= s_ao[t + 1]) ++t;
      wk = tid; wreg = A.w[s_arow[t] + (tid - s_ao[t])];
    }
    double rireg[3];
#pragma unroll
    for (int e = 0; e < 3; ++e) {
      const int idx = tid + NT * e, i = idx / M, j = idx - i * M;
      rireg[e] = (refgrp && idx < M * M && j <= i) ? A.panels[s_rowoff[i] + P + j] : 0.0;
    }
    {
      const int idx = tid, mj = M * J;   // pass 1: thread (row r, ancestor t)
      const int ii = mj > 0 ? min(idx, mj - 1) : 0, r = J ? ii / J : 0, t = J ? ii - r * J : 0;
      const int ma = J ? s_am[t] : 0;
      const double *row = A.panels + s_rowoff[r] + (J ? s_ao[t] : 0);
#pragma unroll
      for (int jj = 0; jj < 28; ++jj) xq[jj] = (idx < mj && jj < ma) ? row[jj] : 0.0;
    }
#pragma unroll
    for (int cc = 0; cc < 4; ++cc)
      smv[cc] = (refgrp && tid < M && cc < B0.ndch) ? A.acc[s_coff[min(cc, max(B0.ndch - 1, 0))] + B0.acc_len + M * M + tid] : 0.0;
    if (k2 >= 0 && k2 < P) {   // pass 2: every row of the group at column k2, and the children's vectors for it
      int t = 0;
      while (t + 1 < J && k2 >= s_ao[t + 1]) ++t;
      const int ma = s_am[t], i = k2 - s_ao[t];
#pragma unroll
      for (int cc = 0; cc < 4; ++cc) ch2[cc] = (cc < nch2) ? A.acc[s_coff[min(cc, max(nch2 - 1, 0))] + s_aoff[t] + ma * ma + i] : 0.0;
#pragma unroll
      for (int rr = 0; rr < 28; ++rr) x2[rr] = (rr < M) ? A.panels[s_rowoff[min(rr, M - 1)] + k2] : 0.0;
    }
#pragma unroll
    for (int e = 0; e < 3; ++e) {   // (last: they wait for `so`, itself a load behind the descriptor)
      const int idx = tid + NT * e, i = idx / M, j = idx - i * M;
      s0v[e] = (s0_load && idx < M * M && j <= i) ? A.s0[so + idx] : 0.0;
    }
    if (rowthr) t_ = A.tausq_inv[mv_];
    if (tid >= 32 && tid < 64) { const int j = tid - 32; tsq[j] = t_; yx[j] = y_ - xb_; zc[j] = z_; rjv[j] = r_; }
    if (wk >= 0) wv[wk] = wreg;
    for (int k = NT + tid; k < P; k += NT) {   // (chains of more than 256 rows)
      int t = 0;
      while (t + 1 < J && k >= s_ao[t + 1]) ++t;
      wv[k] = A.w[s_arow[t] + (k - s_ao[t])];
    }
    if (refgrp) {
#pragma unroll
      for (int e = 0; e < 3; ++e) {
        const int idx = tid + NT * e, i = idx / M, j = idx - i * M;
        if (idx < M * M) Rc[i * CH_LD + j] = rireg[e];
      }
      for (int idx = 3 * NT + tid; idx < M * M; idx += NT) {   // (groups of more than 27 rows)
        const int i = idx / M, j = idx - i * M;
        Rc[i * CH_LD + j] = (j <= i) ? A.panels[s_rowoff[i] + P + j] : 0.0;
      }
    }
    __syncthreads();
  } else {
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
  }
  // ---- pass 1: segment sums seg[t][r] = sum_j N[r][oa_t + j] w_a[j]
  for (int idx = tid; idx < M * J; idx += NT) {
    const int r = idx / J, t = idx - r * J;
    const int ma = s_am[t], oa = s_ao[t];
    const double *row = A.panels + s_rowoff[r] + oa;
    const double *wa = wv + oa;
    double a = 0.0;
    if (LAT && idx == tid && ma <= 28) {   // the segment is in registers (same products, same order; the zero tail adds nothing)
#pragma unroll
      for (int jj = 0; jj < 28; ++jj) a += xq[jj] * ((jj < ma) ? wa[jj] : 0.0);
    } else
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
        if (s0_load) a = (LAT && idx < 3 * NT) ? (idx < NT ? s0v[0] : (idx < 2 * NT ? s0v[1] : s0v[2])) : A.s0[so + idx];
        else {
          double ch[4];   // the children's records: four loads in flight, fixed summation order
          for (int c0 = 0; c0 < s_nch; c0 += 4) {
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) ch[cc] = (c0 + cc < s_nch) ? A.acc[s_coff[min(c0 + cc, s_nch - 1)] + B0.acc_len + idx] : 0.0;
            if (c0 == 0) for (int k = i; k < M; ++k) a += Rc[k * CH_LD + i] * Rc[k * CH_LD + j];
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) a += ch[cc];
          }
          if (s_nch == 0) for (int k = i; k < M; ++k) a += Rc[k * CH_LD + i] * Rc[k * CH_LD + j];
          if (so >= 0) A.s0[so + idx] = a;   // s0_mode 1
        }
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
        for (int cc = 0; cc < 4; ++cc)
          ch[cc] = (LAT && c0 == 0) ? smv[cc] : ((c0 + cc < s_nch) ? A.acc[s_coff[min(c0 + cc, s_nch - 1)] + B0.acc_len + M * M + i] : 0.0);
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
  const int kfirst = LAT ? (tid >= 64 ? k2 : NT - 64 + tid) : tid;   // LAT: waves 1-3 hold columns 0 .. 191 in registers; beyond: late
  for (int k = kfirst; k < P; k += NT) {
    int t = 0;
    while (t + 1 < J && k >= s_ao[t + 1]) ++t;
    const int ma = s_am[t], i = k - s_ao[t];
    const double *avt = seg + t * 32;
    const bool held = LAT && tid >= 64 && k == k2 && M <= 28;
    double ch[4];   // the children's vectors: requested first, added after the dot product in a fixed order
    const int nch = A.no_fwd ? 0 : s_nch;
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) ch[cc] = held ? ch2[cc] : ((cc < nch) ? A.acc[s_coff[min(cc, max(nch - 1, 0))] + s_aoff[t] + ma * ma + i] : 0.0);
    double a = 0.0;
    if (held) {
#pragma unroll
      for (int rr = 0; rr < 28; ++rr) a -= x2[rr] * avt[min(rr, 31)];
      // (rows beyond M: x2 = 0 and avt = 0 -- the same no-ops the batches of eight below append)
    } else
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
template __global__ void k_sample_lean<false>(SampleFastArgs);
template __global__ void k_sample_lean<true>(SampleFastArgs);

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
  // ---- pass 1: segment sums seg[t][r] = sum_j N[r][oa_t + j] w_a[j].
  // Round 3: the chain part of the panel is read by ROWS (lanes on consecutive entries: whole 512-byte runs), four rows per trip
  // through an LDS staging area, and lane (row, ancestor) then walks its segment there -- in the order and with the arithmetic of
  // the direct form below, whose 64 lanes read 64 different 25-entry segments per load instruction (63 sectors per request
  // instead of 8: the kernel was bound by those requests).  The staging area is Ri's (filled afterwards from registers
  // requested up front); blocks too short for it to hold a chain row keep the direct form.
  const int PS = P | 1;
  const int RG = min(4, (A.Mrows * CH_LD) / PS);
  if (RG >= 1) {   // wave-uniform
    constexpr int NRI = (27 * 27 + 63) / 64;
    double rireg[NRI];
#pragma unroll
    for (int q = 0; q < NRI; ++q) {
      const int idx = lane + 64 * q, i = idx / M, j = idx - i * M;
      rireg[q] = (idx < M * M && j <= i) ? A.panels[bpan + (size_t)i * bld + P + j] : 0.0;
    }
    double *stg = Rc;
    for (int r0 = 0; r0 < M; r0 += RG) {
      for (int c0 = 0; 64 * c0 < P; c0 += 4) {
        double x[4][4];
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int k = lane + 64 * (c0 + c);
            x[rr][c] = (rr < RG && r0 + rr < M && k < P) ? A.panels[bpan + (size_t)(r0 + rr) * bld + k] : 0.0;
          }
#pragma unroll
        for (int rr = 0; rr < 4; ++rr)
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            const int k = lane + 64 * (c0 + c);
            if (rr < RG && k < P) stg[rr * PS + k] = x[rr][c];
          }
      }
      WSYNC();
      if (lane < RG * J) {
        const int rr = lane / J, t = lane - rr * J, r = r0 + rr;
        if (r < M) {
          const int ma = am_of(t), oa = ao_of(t);
          const double *srow = stg + rr * PS + oa;
          const double *wa = wv + oa;
          double a = 0.0;
          for (int j = 0; j < ma; ++j) a += srow[j] * wa[j];
          seg[t * 32 + r] = a;
        }
      }
      // (the next trip's stores follow these reads in program order: one wave's LDS operations execute in order)
    }
    WSYNC();
#pragma unroll
    for (int q = 0; q < NRI; ++q) {   // Ri -> LDS (the panel's last M columns)
      const int idx = lane + 64 * q, i = idx / M, j = idx - i * M;
      if (idx < M * M) Rc[i * CH_LD + j] = rireg[q];
    }
    WSYNC();
  } else {
  for (int idx = lane; idx < M * M; idx += 64) {     // Ri -> LDS (the panel's last M columns)
    const int i = idx / M, j = idx - i * M;
    Rc[i * CH_LD + j] = (j <= i) ? A.panels[bpan + (size_t)i * bld + P + j] : 0.0;
  }
  WSYNC();
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
  }
  double tvi = 0.0;
  if (row) { for (int t = 0; t < J; ++t) tvi += seg[t * 32 + lane]; tv[lane] = tvi; }
  WSYNC();
  // ---- row `lane` of the posterior precision Ri'Ri + the children's Gram parts + tausq_inv, straight into registers
  double a[27];
#pragma unroll
  for (int j = 0; j < 27; ++j) a[j] = 0.0;
  double bv = 0.0;
  const long long so = (A.s0off && A.s0_mode) ? A.s0off[slo(s_gd[6])] : -1;
  if (so >= 0 && A.s0_mode == 2) {                   // (wave-uniform) the theta-only part of the row comes from the cache
    const double *sp = A.s0 + so + (size_t)li * M;
#pragma unroll
    for (int j = 0; j < 27; ++j) a[j] = (row && j <= lane) ? sp[j] : 0.0;
    for (int k = 0; k < M; ++k) bv -= Rc[k * CH_LD + li] * tv[k];
    for (int c = 0; c < nch; ++c) bv += row ? A.acc[coff[c] + acc_len + M * M + li] : 0.0;
  } else {
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
    if (so >= 0 && row) {                              // s0_mode 1
      double *sp = A.s0 + so + (size_t)li * M;
#pragma unroll
      for (int j = 0; j < 27; ++j) if (j <= lane) sp[j] = a[j];
    }
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
// k_sample_leaf with SEGMENT-ALIGNED lanes (round 3).  k_sample_leaf maps lanes to consecutive chain columns, so the
// per-ancestor segment sums of a row -- eight of them at n = 1e6 -- are eight masked 64-lane reductions (about 25
// instructions each: 200 of a row's 260).  Here ancestor t owns a 32-lane HALF of register chunk t >> 1 (lane & 31 = its row,
// blocks have at most 32 rows on this path), so one rotate-and-add butterfly inside the 16-lane rows (dpp_ror_add) plus the four
// row sums through v_readlane give TWO ancestors' sums: 22 instructions per chunk, 88 per row for J = 8, no masks, no
// ancestor search, and a lane's own segment sum stays in the lane for the vector records.  Same data, same result up to the
// summation order inside a segment.  NCH = chunks = ceil(J / 2) <= NCH; spamtree_model.cpp:1091-1155, 1190-1203.
template <int NCH>
__global__ __launch_bounds__(NT, NCH <= 4 ? 4 : 3) void k_sample_leaf_seg(SampleFastArgs A) {
  extern __shared__ double lds[];
  __shared__ int s_am[MAXJ], s_ao[MAXJ + 1];
  __shared__ long long s_arow[MAXJ], s_aoff[MAXJ + 1];
  __shared__ long long s_bpan[32], s_brow[32], s_rowoff[32];
  __shared__ int s_bld[32];
  __shared__ int s_fail;
  __shared__ long long s_coff[64];
  __shared__ int s_nch;
  __shared__ long long s_gd[GD_MAXW];

  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  double *red = lds;                        // 4 x (64 NCH): per-wave column sums, slot = chunk * 64 + lane
  double *tsq = red + 4 * 64 * NCH, *yx = tsq + 32, *zc = yx + 32;

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
  // The rows' own data (tid 32..63: outcome id -> tausq_inv is a dependent pair of loads) travels WITH the ancestors' w and the first
  // batch of panel rows instead of ahead of them: requested here, parked in registers over an LDS-only barrier, stored after the
  // other requests have been issued (a group is a chain of dependent round trips -- 7 rows per wave's worth of arithmetic --, so
  // each one taken off the chain counts: round 3, leaf sweep 0.28 ms at n = 1e6)
  double t_ = 0.0, y_ = 0.0, z_ = 0.0, xb_ = 0.0;
  int mv_ = 0;
  const bool rowthr = tid >= 32 && tid < 64 && tid - 32 < M;
  if (tid >= 32 && tid < 64) {
    const int j = tid - 32;
    long long ro = 0;
    if (j < M) {
      const long long r = G.row0 + j;
      mv_ = A.mv[r]; y_ = A.y[r]; xb_ = A.xb[r]; z_ = A.z[r];
      int bi = 0;
      const long long *gb = s_gd + 8 + 4 * J;   // per block: panel offset, first row, ld
      while (bi + 1 < G.nblk && r >= gb[3 * (bi + 1) + 1]) ++bi;
      ro = gb[3 * bi] + (r - gb[3 * bi + 1]) * gb[3 * bi + 2];
    }
    s_rowoff[j] = ro;
  }
  lds_barrier();   // (LDS only: the loads above stay in flight)
  // this lane's columns: chunk c -> ancestor t = 2 c + (lane >> 5), its row i = lane & 31
  const int hi = lane >> 5, li = lane & 31;
  int kc[NCH];       // chain column, or -1
  double wk[NCH], acc[NCH];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    const int t = 2 * c + hi;
    const bool ok = t < J && li < s_am[min(t, MAXJ - 1)];
    kc[c] = ok ? s_ao[t] + li : -1;
    wk[c] = ok ? A.w[s_arow[t] + li] : 0.0;
    acc[c] = 0.0;
  }
  double v[4][NCH], rjv[4];   // a batch of panel rows: rows wid + 4 (4 b + rr)
  auto fetch = [&](int b) {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int j = wid + 4 * (4 * b + rr);
      const double *src = A.panels + s_rowoff[min(j, M - 1)];
#pragma unroll
      for (int c = 0; c < NCH; ++c) v[rr][c] = (j < M && kc[c] >= 0) ? src[kc[c]] : 0.0;
      rjv[rr] = j < M ? src[P] : 0.0;          // the row's r_j (same address for every lane)
    }
  };
  fetch(0);
  if (rowthr) t_ = A.tausq_inv[mv_];
  if (tid >= 32 && tid < 64) { const int j = tid - 32; tsq[j] = t_; yx[j] = y_ - xb_; zc[j] = z_; }
  __syncthreads();
#pragma unroll 1
  for (int b = 0; b < 2; ++b) {
    if (b == 1) {
      if (M <= 16) break;   // (workgroup-uniform) no second batch
      fetch(1);
    }
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) {
      const int j = wid + 4 * (4 * b + rr);
      if (j < M) {   // wave-uniform
        double own[NCH];     // this lane's ancestor's segment sum, per chunk
        double tvj = 0.0;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
          if (2 * c < J) {   // wave-uniform
            double x = v[rr][c] * wk[c];
            x = dpp_ror_add(x, 8); x = dpp_ror_add(x, 4); x = dpp_ror_add(x, 2); x = dpp_ror_add(x, 1);
            const double r0 = readlane_f64(x, 0), r1 = readlane_f64(x, 16), r2 = readlane_f64(x, 32), r3 = readlane_f64(x, 48);
            const double se = r0 + r1, so = r2 + r3;     // ancestors 2 c and 2 c + 1 (zero beyond J)
            tvj += se; tvj += so;                        // tv = the segment sums in ancestor order
            own[c] = hi ? so : se;
          } else own[c] = 0.0;
        }
        const double rj = rjv[rr];
        const double sig = rj * rj + tsq[j];
        if (!(sig > 0.0) && lane == 0) s_fail = 1;
        const double mu = -rj * tvj + tsq[j] * yx[j];
        const double cc = 1.0 / sqrt(sig);
        const double wj = cc * cc * mu + cc * zc[j];
        if (lane == 0) A.w[G.row0 + j] = wj;
        const double evj = rj * wj + tvj;
        // this row's share of the vector records: -N[j][k] (ev_j - seg_t(k)[j])
#pragma unroll
        for (int c = 0; c < NCH; ++c) acc[c] -= v[rr][c] * (evj - own[c]);
      }
    }
  }
#pragma unroll
  for (int c = 0; c < NCH; ++c) red[(wid * NCH + c) * 64 + lane] = acc[c];
  __syncthreads();
  double *rec = A.acc + B0.acc_off;
  for (int e = tid; e < 64 * NCH; e += NT) {
    const int c = e >> 6, l = e & 63, t = 2 * c + (l >> 5), i = l & 31;
    if (t < J && i < s_am[t]) {
      const int ma = s_am[t];
      double a = ((red[e] + red[64 * NCH + e]) + red[2 * 64 * NCH + e]) + red[3 * 64 * NCH + e];
      for (int cc = 0; cc < s_nch; ++cc) a += A.acc[s_coff[cc] + s_aoff[t] + ma * ma + i];
      rec[s_aoff[t] + ma * ma + i] = a;
    }
  }
  if (tid == 0 && s_fail) atomicMin(A.errflag, B0.level * 16 + 11);
}
template __global__ void k_sample_leaf_seg<4>(SampleFastArgs);
template __global__ void k_sample_leaf_seg<6>(SampleFastArgs);

// ---------------------------------------------------------------------------------------------------------------
// Phase B, non-reference blocks of WIDE-block trees (the default multivariate tree: 36-row leaves behind seven 75-row
// ancestors, config #4), which are not column groups and took the generic kernel's three passes over the panel (2.2 ms at #4).
// The segment-aligned scheme of k_sample_leaf_seg with THREE 32-lane halves per ancestor (ancestors of at most 96 rows, at
// most 8 of them: 24 halves = 12 register chunks): a wave owns whole rows, one coalesced pass over the panel, two rows at a
// time.  One block per workgroup (a block's record is its own; non-reference blocks have no children).
// spamtree_model.cpp:1091-1155, 1190-1203.
__global__ __launch_bounds__(NT, 2) void k_sample_leaf_wide(SampleArgs A) {
  constexpr int NCH = 12, NHA = 3;   // chunks; halves per ancestor
  extern __shared__ double lds[];
  __shared__ int s_am[MAXJ], s_ao[MAXJ + 1];
  __shared__ long long s_arow[MAXJ], s_aoff[MAXJ + 1];
  __shared__ int s_fail;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  double *red = lds;                         // 4 x (64 NCH): per-wave column sums
  double *tsq = red + 4 * 64 * NCH, *yx = tsq + 64, *zc = yx + 64;
  const int hi = lane >> 5, li = lane & 31;
  for (int bi = blockIdx.x; bi < A.nlist; bi += gridDim.x) {
    const int b = A.list[bi];
    const Blk B = A.blks[b];
    const int M = B.m, P = B.P, J = B.nanc, ld = B.ld;   // M <= 64, J <= 8, ancestors <= 96 rows (host)
    __syncthreads();
    if (tid < J) {
      const int a = A.anc_idx[B.anc_ptr + tid];
      s_am[tid] = A.blks[a].m;
      s_arow[tid] = A.blks[a].row0;
    }
    if (tid == 0) s_fail = 0;
    if (tid >= 64 && tid < 64 + M) {
      const int j = tid - 64;
      const long long r = B.row0 + j;
      tsq[j] = A.tausq_inv[A.mv[r]]; yx[j] = A.y[r] - A.xb[r]; zc[j] = A.z[r];
    }
    __syncthreads();
    if (tid == 0) {
      int o = 0;
      long long ao = 0;
      for (int t = 0; t < J; ++t) { s_ao[t] = o; s_aoff[t] = ao; o += s_am[t]; ao += (long long)s_am[t] * s_am[t] + s_am[t]; }
      s_ao[J] = o; s_aoff[J] = ao;
    }
    __syncthreads();
    // this lane's columns: chunk c, half hx = 2 c + hi -> ancestor t = hx / 3, its rows 32 (hx % 3) + li
    int kc[NCH];
    double wk[NCH], acc[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int hx = 2 * c + hi, t = hx / NHA, i = 32 * (hx - NHA * t) + li;
      const bool ok = t < J && i < s_am[min(t, MAXJ - 1)];
      kc[c] = ok ? s_ao[t] + i : -1;
      wk[c] = ok ? A.w[s_arow[t] + i] : 0.0;
      acc[c] = 0.0;
    }
    const double *pg = A.panels + B.panel_off;
#pragma unroll 1
    for (int j0 = 2 * wid; j0 < M; j0 += 2 * (NT / 64)) {
      double v[2][NCH], rjv[2];
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        const int j = j0 + rr;
        const double *src = pg + (size_t)min(j, M - 1) * ld;
#pragma unroll
        for (int c = 0; c < NCH; ++c) v[rr][c] = (j < M && kc[c] >= 0) ? src[kc[c]] : 0.0;
        rjv[rr] = j < M ? src[P] : 0.0;
      }
#pragma unroll
      for (int rr = 0; rr < 2; ++rr) {
        const int j = j0 + rr;
        if (j < M) {   // wave-uniform
          double hs[2 * NCH];       // the sums of the 24 halves (wave-uniform)
#pragma unroll
          for (int c = 0; c < NCH; ++c) {
            if (2 * c < NHA * J) {   // wave-uniform
              double x = v[rr][c] * wk[c];
              x = dpp_ror_add(x, 8); x = dpp_ror_add(x, 4); x = dpp_ror_add(x, 2); x = dpp_ror_add(x, 1);
              hs[2 * c] = readlane_f64(x, 0) + readlane_f64(x, 16);
              hs[2 * c + 1] = readlane_f64(x, 32) + readlane_f64(x, 48);
            } else { hs[2 * c] = 0.0; hs[2 * c + 1] = 0.0; }
          }
          double sg[2 * NCH / NHA];   // per ancestor: its three halves in order
          double tvj = 0.0;
#pragma unroll
          for (int t = 0; t < 2 * NCH / NHA; ++t) { sg[t] = (hs[NHA * t] + hs[NHA * t + 1]) + hs[NHA * t + 2]; tvj += sg[t]; }
          const double rj = rjv[rr];
          const double sig = rj * rj + tsq[j];
          if (!(sig > 0.0) && lane == 0) s_fail = 1;
          const double mu = -rj * tvj + tsq[j] * yx[j];
          const double cc = 1.0 / sqrt(sig);
          const double wj = cc * cc * mu + cc * zc[j];
          if (lane == 0) A.w[B.row0 + j] = wj;
          const double evj = rj * wj + tvj;
#pragma unroll
          for (int c = 0; c < NCH; ++c) {
            const double own = hi ? sg[(2 * c + 1) / NHA] : sg[(2 * c) / NHA];
            acc[c] -= v[rr][c] * (evj - own);
          }
        }
      }
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c) red[(wid * NCH + c) * 64 + lane] = acc[c];
    __syncthreads();
    double *rec = A.acc + B.acc_off;
    for (int e = tid; e < 64 * NCH; e += NT) {
      const int c = e >> 6, l = e & 63, hx = 2 * c + (l >> 5), t = hx / NHA, i = 32 * (hx - NHA * t) + (l & 31);
      if (t < J && i < s_am[t]) {
        const int ma = s_am[t];
        rec[s_aoff[t] + (long long)ma * ma + i] = ((red[e] + red[64 * NCH + e]) + red[2 * 64 * NCH + e]) + red[3 * 64 * NCH + e];
      }
    }
    if (tid == 0 && s_fail) atomicMin(A.errflag, B.level * 16 + 11);
  }
}

template __global__ void k_sample<false, false>(SampleArgs);
template __global__ void k_sample<true, false>(SampleArgs);
template __global__ void k_sample<true, true>(SampleArgs);
#else   // host side: prototypes only (the kernels are compiled in their own translation unit)
template <bool BIG, bool NOREF = false> __global__ void k_sample(SampleArgs A);
__global__ void k_gram_big(GramBigArgs A);
__global__ void k_sample_leaf_wide(SampleArgs A);
__global__ void k_sample_mfma(SampleFastArgs A);
__global__ void k_gram(SampleFastArgs A);
__global__ void k_gram_direct(SampleFastArgs A);
template <bool LAT> __global__ void k_sample_lean(SampleFastArgs A);
__global__ void k_sample_wave(SampleFastArgs A);
__global__ void k_sample_leaf(SampleFastArgs A);
template <int NCH> __global__ void k_sample_leaf_seg(SampleFastArgs A);
#endif
