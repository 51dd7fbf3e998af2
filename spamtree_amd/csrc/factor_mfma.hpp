// Phase A, column-group kernel of the small / ineligible levels (k_factor_mfma).
#pragma once
#include "st_device.hpp"

// k_factor_mfma: one column group per workgroup
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
#ifdef ST_DEFS_FACTOR_MFMA


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


#else   // host side: prototypes only (the kernels are compiled in their own translation unit)
__global__ void k_factor_mfma(FastArgs A, CovPar cp);
#endif
