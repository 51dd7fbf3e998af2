// Phase A, third generation ("quads").  Included by spamtree_hip.hip (needs Blk, Grp, CovPar, cov_entry, d4, CH_LD).
//
// A workgroup factorises up to NU "units" at once.  A unit is what k_factor_mfma calls a column group: one reference
// block, or up to 32 columns of sibling non-reference blocks.  The units of a quad share their ancestor chain, except
// possibly for the last ancestor ("private" ancestor: leaf groups whose parents are siblings).  The shared chain's
// inverse-Cholesky panels are staged through LDS ONCE for all units: 32 consecutive ROWS OF THE CONCATENATED CHAIN at a
// time (the chain's inverse Cholesky factor is one lower-triangular Pc x Pc matrix kept as one row panel per ancestor, so a
// step may straddle two ancestors: its rows then have different lengths and the shorter ones are zero-filled -- 175 chain
// rows are 11 row tiles of 16 instead of the 14 that 7 separately padded 25-row panels make) by LDS-DMA into two
// buffers, the next step in flight while the matrix cores work on the current one, one LDS-only barrier per step;
// per panel every wave has 4x the matrix work of k_factor_mfma between barriers and nothing is exchanged between the
// waves of the main loop (the private ancestors' panels: all units side by side, 16 rows at a time, before it):
//   * wave (u, jt) owns 16 columns of unit u over the WHOLE chain: its K_{pa,u} B operands live in registers (kx),
//     evaluated once straight from the coordinates; so do its T = H_u accumulators (tacc, tiles [column][chain]);
//   * V = Linv_panel K (A from LDS, B = kx; two row tiles per panel);  T += V' Linv_panel (the V tiles in C layout are
//     the A operand, B from LDS);  the Schur complement K_uu - V'V is accumulated on the fly (reference units: MFMA on the V tiles, the
//     off-diagonal tile uses the partner wave's V tiles of the previous panel through LDS; leaf units: diagonal only);
//   * epilogue: NU Cholesky eliminations side by side (team = the unit's two waves, one barrier per pivot for all),
//     N = -Ri T on the matrix cores with the T tiles as B operands (half of them swapped between the two waves so that
//     each wave holds all columns for its chain tiles), outputs straight from registers.
// Chain length P <= 4 * NKX, NKT = ceil(P / 16) T tiles.  Results equal k_factor_mfma's up to rounding.
#pragma once

struct Quad {
  int g0, nu;   // units = column groups g0 .. g0+nu-1 of the launch's group list
  int Jc, Pc;   // shared chain: the first Jc ancestors of every unit (Pc rows); a unit has Jc or Jc + 1 ancestors
};

struct QuadArgs {
  const Blk *blks;
  const int *anc_idx;
  const Grp *grps;
  const Quad *quads;
  int nquad;
  const double *cx, *cy;
  const int *mv;
  const double *w;
  double *panels;
  double *logdet_c, *loglik_c;
  int *errflag;
  const long long *gdesc;   // group descriptors of the level's first group onwards (Quad::g0 is relative to it)
  int gd_stride;
  int ldS;   // staged row stride: >= longest row + 24 zero-filled columns
  int wave_chol;   // the level's blocks have <= 27 rows: one-wave register elimination (a property of the LEVEL, so that a
                   // unit's arithmetic does not depend on which units share its workgroup)
  int predict;     // leaf instantiations: phase P (spamtree_model.cpp:1296-1326) -- draw w_j = H_j w_pa + sqrt(max(K_jj - H_j K_pa,j, 0)) z_j
  const double *z; // ... from these normals (device order), into w_out; nothing else is written
  double *w_out;
};

#define RFL(x) __builtin_amdgcn_readfirstlane(x)

// Row stride (doubles) of the staged chain rows, a function of the instantiation only, so that every LDS operand address
// of the main loop is ONE per-lane base register + an immediate offset: >= the longest chain (4 NKX) + 24 zero-filled
// columns, >= 186 (the epilogue's per-unit overlay of 16 rows: Ri [0, 1056), then one 192-double hand-over slot per T tile --
// ceil(4 NKX / 16) of them -- over R and the elimination scratch), and 2 x odd (conflict-free A-operand reads).
constexpr int QUAD_LEAF_KH = 5;   // leaf levels: K-steps per pass of the covariance scratch (8 waves x 5 x 64 doubles behind the arena)
__host__ __device__ constexpr int quad_lds_stride(int nkx) {
  int s = 4 * nkx + 24 > 186 ? 4 * nkx + 24 : 186;
  while ((s & 1) || ((s >> 1) & 1) == 0) ++s;
  return s;
}

#ifdef ST_DEFS_FACTOR_QUAD
// One row of Kb doubles, global -> LDS, by LDS-DMA (16 bytes per lane, 1 KiB per wave-instruction, no registers): lane l
// moves doubles 2l, 2l+1 of each 128-double piece.  The source needs 8-byte alignment only.  When Kb is odd the last
// active lane also drops the row's successor into column Kb: the caller zero-fills [Kb, Kb+24) after the data has landed.
// `two`: issue the second piece (wave-uniform; callers that count instructions pass Kb > 128).
__device__ __forceinline__ void dma_row(const double *src, double *dst, int Kb, int lane, bool two) {
  if (2 * lane < Kb) __builtin_amdgcn_global_load_lds((q_glb_void *)(src + 2 * lane), (q_lds_void *)dst, 16, 0, 0);
  if (two) {
    if (128 + 2 * lane < Kb) __builtin_amdgcn_global_load_lds((q_glb_void *)(src + 128 + 2 * lane), (q_lds_void *)(dst + 128), 16, 0, 0);
  }
}

// WCH (reference levels): the level's blocks have <= 27 rows -> one-wave register elimination; else the team elimination.
// A template parameter, not a run-time branch: with both bodies in one kernel the register allocator budgets for the larger
// one and parks a T accumulator in scratch for the whole main loop.
template <int NU, int NKX, int NKT, bool ISREF, bool WCH = true>
__global__ __launch_bounds__(128 * NU, 2) void k_factor_quad(QuadArgs A, CovPar cp) {
  constexpr int NTQ = 128 * NU, NW = 2 * NU;
  // covariance scratch ([KH][64] doubles per wave, lane-private slots): reference levels keep it in the SECOND staging buffer
  // (4 passes), leaf levels behind the arena (KH = 5: 20 KB), so that the first panel rows can travel by LDS-DMA UNDER the
  // covariance pass instead of after it
  constexpr int PMAX = 4 * NKX, KH = ISREF ? (NKX + 3) / 4 : QUAD_LEAF_KH, NPASS = (NKX + KH - 1) / KH;
  static_assert(NKT * 16 >= PMAX, "T tiles must cover the chain");
  extern __shared__ double lds[];
  __shared__ int s_am[MAXJ], s_ao[MAXJ + 1];
  __shared__ long long s_arow[MAXJ], s_apan[MAXJ];
  __shared__ int s_uM[NU], s_uP[NU], s_ublk0[NU], s_unblk[NU], s_uref[NU], s_uJ[NU], s_pm[NU], s_fail[NU], s_level;
  __shared__ long long s_urow0[NU], s_prow[NU], s_ppan[NU];
  constexpr int NB = ISREF ? 1 : 32, NUL = ISREF ? 1 : NU;   // leaf-only arrays
  __shared__ long long s_bpan[NU][NB], s_brow[NU][NB];
  __shared__ int s_bld[NU][NB];
  __shared__ double s_colx[NU][32], s_coly[NU][32], s_colw[NU][32], s_hv[NU][32];
  __shared__ double s_px[NUL][32], s_py[NUL][32], s_pw[NUL][32];
  __shared__ double s_e2[NU][32], s_lg[NU][32];
  __shared__ int s_colmv[NU][32], s_colblk[NUL][32], s_pmv[NUL][32];
  __shared__ int s_nit;   // steps of 32 chain rows over the shared chain
  __shared__ double s_sx[PMAX], s_sy[PMAX], s_wpa[PMAX];
  __shared__ int s_smv[PMAX];
  __shared__ double s_exp2[64];         // 2^(j/64): cov_exp_tab
  // multivariate covariance: the per-outcome-pair constants (rate, amp, amp2 of every pair, phi of every outcome).  Indexed per
  // lane out of the kernel arguments they are GLOBAL loads -- three or four dependent round trips per entry; the covariance
  // pass was 21-24 % of a quad's life at config #5 (stamps) -- so they live in LDS
  __shared__ double s_cvr[QMAX * QMAX], s_cva[QMAX * QMAX], s_cva2[QMAX * QMAX], s_cvp[QMAX];
  __shared__ int s_rlen[PMAX];          // chain row c: its length (entries up to and including its own ancestor's rows) ...
  __shared__ long long s_rsrc[PMAX];    // ... and where it starts in the panel arena

  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, l4 = lane >> 4, ttid = tid & 127;
  const int wid = RFL(tid >> 6), u = wid >> 1, jt = (wid & 1) ^ ((wid >> 2) & 1);   // waves w, w + 4 share a SIMD: one jt = 0 (it also has the off-diagonal Schur tile) and one jt = 1 each
  constexpr int ldS = quad_lds_stride(NKX);   // == A.ldS (host)
  double *arena = lds;
  double *zrow = arena + (size_t)NU * 16 * ldS;   // a row of zeros
  double *xch = zrow + ldS;                       // reference quads: V tiles of the jt = 1 waves, [NU][2 tiles][256]; leaf quads: the covariance scratch

  STAMP_DECL
  int qidx = blockIdx.x;
  {
    const int per = A.nquad >> 3;   // one contiguous run of quads per XCD (private L2): neighbours share chain panels
    if (qidx < per * 8) qidx = (qidx & 7) * per + (qidx >> 3);
  }
  const Quad Q = A.quads[qidx];
  const int Jc = Q.Jc, Pc = Q.Pc, nu = Q.nu;

  // ---- topology of the quad: the units' group descriptors (layout: GdHead) land in the arena, which is free until the
  // covariance pass -- one round trip instead of the grps -> blks -> anc_idx -> blks chain
  long long *gdl = (long long *)arena;   // [nu][gd_stride]
  const int gds = A.gd_stride;
  for (int e = tid; e < nu * gds; e += NTQ) {
    const int uu = e / gds, i = e - uu * gds;
    gdl[e] = A.gdesc[(size_t)(Q.g0 + uu) * gds + i];
  }
  for (int k = tid; k < ldS; k += NTQ) zrow[k] = 0.0;
  if (tid < 64) s_exp2[tid] = EXP2_64[tid];
  if (cp.q > 1 && tid >= 64 && tid < 64 + QMAX * QMAX) {
    const int e = tid - 64;
    s_cvr[e] = cp.rate[e]; s_cva[e] = cp.amp[e]; s_cva2[e] = cp.amp2[e];
    if (e < QMAX) s_cvp[e] = cp.phi[e];
  }
  __syncthreads();
  if (tid < NU) {
    int M = 0, P = 0, blk0 = 0, nblk = 0, isref = 0, J = 0, pm = 0;
    long long row0 = 0, prow = 0, ppan = 0;
    if (tid < nu) {
      const long long *g = gdl + (size_t)tid * gds;
      row0 = g[0];
      M = (int)(g[2] & 0xffffffffLL); P = (int)(g[2] >> 32);
      J = (int)(g[3] & 0xffffffffLL); nblk = (int)(g[3] >> 32);
      isref = (int)(g[4] & 0xffffffffLL);
      blk0 = (int)(g[6] & 0xffffffffLL);
      if (tid == 0) s_level = (int)(g[4] >> 32);
      if (J > Jc) {   // the private (last) ancestor
        const long long *a = g + 8 + 4 * Jc;
        pm = (int)(a[0] & 0xffffffffLL); prow = a[1]; ppan = a[2];
      }
    }
    s_uM[tid] = M; s_uP[tid] = P; s_ublk0[tid] = blk0; s_unblk[tid] = nblk; s_uref[tid] = isref; s_uJ[tid] = J;
    s_pm[tid] = pm; s_urow0[tid] = row0; s_prow[tid] = prow; s_ppan[tid] = ppan; s_fail[tid] = 0;
  }
  if (tid >= 64 && tid < 64 + Jc) {   // the shared chain: the first Jc ancestors of unit 0
    const int t = tid - 64;
    const long long *a = gdl + 8 + 4 * t;
    s_am[t] = (int)(a[0] & 0xffffffffLL); s_ao[t] = (int)(a[0] >> 32); s_arow[t] = a[1]; s_apan[t] = a[2];
  }
  if (tid == 32) {
    s_ao[Jc] = Pc;
    s_nit = (Pc + 31) >> 5;
  }
  for (int e = tid; e < NU * NB; e += NTQ) {
    const int uu = e / NB, b = e - uu * NB;
    if (uu < nu) {
      const long long *g = gdl + (size_t)uu * gds;
      const int J = (int)(g[3] & 0xffffffffLL), nblk = (int)(g[3] >> 32);
      if (b < nblk) {
        const long long *q = g + 8 + 4 * J + 3 * b;
        s_bpan[uu][b] = q[0]; s_brow[uu][b] = q[1]; s_bld[uu][b] = (int)q[2];
      }
    }
  }
  __syncthreads();
  const int Mu = RFL(s_uM[u]), Pu = RFL(s_uP[u]), pmu = RFL(s_pm[u]);
  constexpr bool isref = ISREF;   // a level is all reference blocks or all leaf groups (host)
  const bool wact = jt * 16 < Mu;   // this wave owns at least one column
  const bool two = Mu > 16;
  int Mmax = 0, pmmax = 0;
#pragma unroll
  for (int i = 0; i < NU; ++i) { Mmax = max(Mmax, s_uM[i]); pmmax = max(pmmax, s_pm[i]); }
  Mmax = RFL(Mmax); pmmax = RFL(pmmax);
  constexpr bool anyref = ISREF;

  // ---- coordinates and w of the shared chain, of the private ancestors, of the units' columns
  for (int k = tid; k < Pc; k += NTQ) {
    int t = 0;   // the ancestor of chain row k: independent compares (a search loop is a chain of dependent LDS reads)
#pragma unroll
    for (int j = 1; j < 8; ++j) t += (j < Jc && k >= s_ao[j]) ? 1 : 0;
    for (int j = 8; j < Jc; ++j) t += (k >= s_ao[j]) ? 1 : 0;
    const long long r = s_arow[t] + (k - s_ao[t]);
    s_sx[k] = A.cx[r]; s_sy[k] = A.cy[r]; s_smv[k] = A.mv[r]; s_wpa[k] = A.w[r];
    const int len = s_ao[t + 1];
    s_rlen[k] = len; s_rsrc[k] = s_apan[t] + (long long)(k - s_ao[t]) * len;
  }
  for (int e = tid; e < NU * 32; e += NTQ) {
    const int uu = e >> 5, i = e & 31;
    double x = 0.0, y = 0.0, ww = 0.0; int v = 0;
    if constexpr (!ISREF) {
      if (i < s_pm[uu]) { const long long r = s_prow[uu] + i; x = A.cx[r]; y = A.cy[r]; ww = A.w[r]; v = A.mv[r]; }
      s_px[uu][i] = x; s_py[uu][i] = y; s_pw[uu][i] = ww; s_pmv[uu][i] = v;
      x = 0.0; y = 0.0; ww = 0.0; v = 0;
    }
    int bi = 0;
    if (i < s_uM[uu]) {
      const long long r = s_urow0[uu] + i;
      x = A.cx[r]; y = A.cy[r]; ww = A.w[r]; v = A.mv[r];
      if constexpr (!ISREF) { while (bi + 1 < s_unblk[uu] && r >= s_brow[uu][bi + 1]) ++bi; }
    }
    s_colx[uu][i] = x; s_coly[uu][i] = y; s_colw[uu][i] = ww; s_colmv[uu][i] = v;
    if constexpr (!ISREF) s_colblk[uu][i] = bi;
  }
  __syncthreads();

  STAMP(0);
  const int p_sr0 = pmu > 16 ? (pmu + 1) >> 1 : pmu;   // rows of the first private sub-panel
  const int p_Kb = Pc + pmu;
  constexpr int RP = 32 / NW;   // staged rows per wave and step
  const int nit = RFL(s_nit);
  // step i of the shared chain covers chain rows [32 (nit-1-i), ...): its rows travel from global memory straight into LDS
  // (the rows' table entries are read TOGETHER -- one LDS latency per step instead of one dependent read per row -- and the
  // lengths stay in registers for the padding pass of the same rows one step later: rlen_nx)
  int rlen_cur[RP], rlen_nx[RP];
#pragma unroll
  for (int rr = 0; rr < RP; ++rr) { rlen_cur[rr] = 0; rlen_nx[rr] = 0; }
  auto issue = [&](int i, double *buf) {
    const int c0 = 32 * (nit - 1 - i), sr = min(32, Pc - c0);
    int lv[RP];
    long long sv[RP];
#pragma unroll
    for (int rr = 0; rr < RP; ++rr) {
      const int cc = c0 + min(wid + NW * rr, sr - 1);
      lv[rr] = s_rlen[cc]; sv[rr] = s_rsrc[cc];
    }
#pragma unroll
    for (int rr = 0; rr < RP; ++rr) {
      const int row = wid + NW * rr;
      const int len = RFL(lv[rr]);
      rlen_nx[rr] = len;
      if (row < sr) dma_row(A.panels + sv[rr], buf + (size_t)row * ldS, len, lane, len > 128);
    }
  };
  // private (last) ancestors, leaf quads: sub-panel sp of every unit side by side, unit u's rows at buf + (u * stride + row) * ldS
  auto priv_rows = [&](int sp) { return sp == 0 ? p_sr0 : (pmu > 16 ? pmu - p_sr0 : 0); };
  auto priv_issue = [&](int sp, int stride) {
    const int r0 = sp == 0 ? 0 : p_sr0, sr = priv_rows(sp);
    const double *src = A.panels + s_ppan[u] + (size_t)r0 * p_Kb;
    double *buf = arena + (size_t)u * stride * ldS;
#pragma unroll
    for (int rr = 0; rr < 8; ++rr) {
      const int row = jt + 2 * rr;
      if (row < sr) dma_row(src + (size_t)row * p_Kb, buf + (size_t)row * ldS, p_Kb, lane, true);
    }
  };
  // the first rows are requested NOW and land under the covariance pass (whose scratch lies elsewhere)
  if constexpr (ISREF) { if (nit > 0) issue(0, arena); }
  else { if (pmmax > 0) priv_issue(0, 16); }

  // ---- K_{pa,u}: every lane evaluates the B operands of its own K-steps, kx[st] = K[4 st + l4][column jt*16 + l15 of
  // unit u], in a rolled loop (one covariance body per pass) through lane-private LDS slots, then picks them up with
  // static register indices.  No barrier: nobody else touches the slots.
  double kx[NKX];
  {
    const int jc = jt * 16 + l15;
    const bool cok = jc < Mu;
    const double mx = s_colx[u][jc], my = s_coly[u][jc];
    const int mvj = s_colmv[u][jc];
    double *kb = (ISREF ? arena + (size_t)32 * ldS : xch) + (size_t)wid * (KH * 64) + lane;   // [KH][64] per wave (sizes: host)
#pragma unroll
    for (int hpass = 0; hpass < NPASS; ++hpass) {
      const int st0 = hpass * KH;
      if (4 * st0 < Pu) {
        if (cp.q == 1) {   // cexpcov, constants hoisted
          const double s2 = cp.ai1[0], nphi = -cp.tmv[0];
#pragma unroll 1
          for (int i = 0; i < KH; ++i) {
            const int k = 4 * (st0 + i) + l4;
            double ax = s_sx[min(k, PMAX - 1)], ay = s_sy[min(k, PMAX - 1)];
            if constexpr (!ISREF) {
              if (k >= Pc) { ax = s_px[u][(k - Pc) & 31]; ay = s_py[u][(k - Pc) & 31]; }
            }
            const double dx = ax - mx, dy = ay - my;
            const double v = s2 * cov_exp_tab(nphi * cov_sqrt(fma(dx, dx, dy * dy)), s_exp2);
            kb[i * 64] = (cok && k < Pu) ? v : 0.0;
          }
        } else {
          const int qn = cp.q;
#pragma unroll 1
          for (int i = 0; i < KH; ++i) {
            const int k = 4 * (st0 + i) + l4;
            // branch-free: every lane evaluates both exponentials (amp2 is zero for pairs of different outcomes; within a wave the
            // outcomes are mixed, so a branch on it made every wave walk both sides anyway), selects on the addresses
            const double *pxs = &s_sx[min(k, PMAX - 1)], *pys = &s_sy[min(k, PMAX - 1)];
            const int *pvs = &s_smv[min(k, PMAX - 1)];
            if constexpr (!ISREF) {
              if (k >= Pc) { pxs = &s_px[u][(k - Pc) & 31]; pys = &s_py[u][(k - Pc) & 31]; pvs = &s_pmv[u][(k - Pc) & 31]; }
            }
            const int av = *pvs;
            const double dx = *pxs - mx, dy = *pys - my;
            const double hd = cov_sqrt(dx * dx + dy * dy);
            const int ij = av * qn + mvj;
            double v = s_cva[ij] * cov_exp_tab(-s_cvr[ij] * hd, s_exp2);
            v = fma(s_cva2[ij], cov_exp_tab(-s_cvp[av] * hd, s_exp2), v);
            kb[i * 64] = (cok && k < Pu) ? v : 0.0;
          }
        }
#pragma unroll
        for (int i = 0; i < KH; ++i)
          if (st0 + i < NKX) kx[st0 + i] = kb[i * 64];
      } else {
#pragma unroll
        for (int i = 0; i < KH; ++i)
          if (st0 + i < NKX) kx[st0 + i] = 0.0;
      }
    }
  }
  // no barrier here: the scratch slots are lane-private, and the first DMA into their region (reference quads: step 1 into
  // the second buffer) is requested after the first barrier below
  STAMP(1);
  d4 tacc[NKT];
#pragma unroll
  for (int n = 0; n < NKT; ++n) tacc[n] = (d4){0.0, 0.0, 0.0, 0.0};
  d4 rown = (d4){0.0, 0.0, 0.0, 0.0}, rcross = rown;   // Schur tiles (jt, jt), (1, 0)
  double dacc = 0.0;                                   // leaf units: sum_k V[k][column l15]^2 (this lane's rows)

#define QMFMA(a_, b_, c_) c_ = __builtin_amdgcn_mfma_f64_16x16x4f64(a_, b_, c_, 0, 0, 0)
  // One 16-row tile of the step (rows at tg, nk = ceil(rows / 4) K-steps of the T update, row length KbT): V = Linv K for
  // the wave's columns, the off-diagonal Schur tile, then T += V' Linv and the diagonal Schur tile, with ONE V accumulator
  // live at a time (two tiles side by side made the register allocator keep a T accumulator in scratch: its reloads wait on
  // vmcnt, i.e. on the LDS-DMA of the next step too).  Reference units: the jt = 1 wave hands its V tile to its partner
  // through LDS slot `slot` of the unit (one workgroup barrier -- every wave takes it, with or without columns), and the
  // jt = 0 wave forms the Schur tile (1, 0) = V_1' V_0 from it and its own tile, still in registers.
  // rmask >= 0 (private sub-panels packed without padding rows): V rows >= rmask are forced to zero (what the tile holds there
  // belongs to the next unit: finite, but not ours).
  auto tile = [&](const double *tg, int nk, int KbT, int slot, int rmask) __attribute__((always_inline)) {
    d4 p = (d4){0.0, 0.0, 0.0, 0.0};
    if (wact) {
      const int ns = (KbT + 3) >> 2;
      const double *ap = tg + l15 * ldS + l4;   // rows beyond the step's last are zero (staging)
#pragma unroll
      for (int c = 0; c < (NKX + 3) / 4; ++c) {
        if (4 * c >= ns) break;   // early exit (not a guarded body): one scalar compare per group, no predicate kept live
        double a[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[i] = ap[4 * (4 * c + i)];
#pragma unroll
        for (int i = 0; i < 4; ++i)
          if (4 * c + i < NKX) QMFMA(a[i], kx[4 * c + i], p);
      }
      if (rmask >= 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) p[r] = (4 * r + l4 < rmask) ? p[r] : 0.0;
      }
    }
    if constexpr (ISREF) {
      double *xp = xch + (u * 2 + slot) * 256 + lane;
      if (jt == 1 && wact) { xp[0] = p[0]; xp[64] = p[1]; xp[128] = p[2]; xp[192] = p[3]; }
      lds_barrier();
      if (jt == 0 && two) {
        const double q0 = xp[0], q1 = xp[64], q2 = xp[128], q3 = xp[192];
        QMFMA(q0, p[0], rcross); QMFMA(q1, p[1], rcross); QMFMA(q2, p[2], rcross); QMFMA(q3, p[3], rcross);
      }
    }
    if (wact) {
      // T[column][chain k] += V' Linv: A = the V tile (C layout read as A: contraction over the tile's rows), B from LDS
      const double *b0 = tg + l4 * ldS + l15;
#pragma unroll
      for (int n = 0; n < NKT; ++n) {
        if (n * 16 >= KbT) break;
        const double x0 = b0[16 * n], x1 = b0[4 * ldS + 16 * n], x2 = b0[8 * ldS + 16 * n], x3 = b0[12 * ldS + 16 * n];
        QMFMA(p[0], x0, tacc[n]);
        if (nk > 1) QMFMA(p[1], x1, tacc[n]);
        if (nk > 2) QMFMA(p[2], x2, tacc[n]);
        if (nk > 3) QMFMA(p[3], x3, tacc[n]);
      }
      if (isref) {
        QMFMA(p[0], p[0], rown); QMFMA(p[1], p[1], rown); QMFMA(p[2], p[2], rown); QMFMA(p[3], p[3], rown);
      } else {
        dacc += p[0] * p[0] + p[1] * p[1] + p[2] * p[2] + p[3] * p[3];
      }
    }
  };
  // one step (sr <= 32 rows of Linv staged at stg with stride ldS; Kb / KbA: length of the longest row of the step / of its
  // first tile -- shorter when the step straddles ancestors; columns up to Kb + 24 are zero beyond a row's own length).
  // Called by every wave of the workgroup (barriers inside for reference units); sr is the same for all of them.
  auto compute = [&](const double *stg, int sr, int Kb, int KbA) __attribute__((always_inline)) {
    tile(stg, (min(sr, 16) + 3) >> 2, KbA, 0, -1);
    if (sr > 16) tile(stg + 16 * ldS, (sr - 16 + 3) >> 2, Kb, 1, -1);
  };

  // ---- private (last) ancestors (leaf quads): every unit's sub-panel staged side by side (LDS-DMA), all waves busy.
  // Sub-panel 0 (16-row slots, the whole arena) was requested before the covariance pass.  Sub-panel 1, when it has <= 12
  // rows per unit, is packed WITHOUT padding rows into rows [0, 48): rows [48, 64) then take the first step of the shared
  // chain (<= 16 rows: 175 chain rows = 5 x 32 + 15), requested when the matrix cores start on sub-panel 1.
  bool pf = false;   // the shared chain's first step sits at row 48 (workgroup-uniform)
  if constexpr (!ISREF) if (pmmax > 0) {
    const int ts1 = pmmax > 16 ? pmmax >> 1 : 0;                      // rows of sub-panel 1, longest unit
    const bool tight = ts1 > 0 && ts1 <= 12;
    pf = tight && nit > 0 && Pc - 32 * (nit - 1) <= 16;
#pragma unroll
    for (int sp = 0; sp < 2; ++sp) {
      if (sp == 0 || pmmax > 16) {
        const int stride = (sp == 1 && tight) ? ts1 : 16;
        const int sr = priv_rows(sp);
        double *buf = arena + (size_t)u * stride * ldS;
        if (sp == 1) priv_issue(1, stride);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int rr = 0; rr < 8; ++rr) {
          const int row = jt + 2 * rr;
          if (row < sr && lane < 24) buf[(size_t)row * ldS + p_Kb + lane] = 0.0;
          if (row >= sr && stride == 16) for (int k = lane; k < p_Kb + 24; k += 64) buf[(size_t)row * ldS + k] = 0.0;   // absent rows of the tile
        }
        lds_barrier();
        if (sp == 1 && pf) issue(0, arena + (size_t)48 * ldS);
        // (row i of the sub-panel is chain row Pc + r0 + i of a LOWER-TRIANGULAR factor: nothing beyond column Pc + r0 + sr)
        if (sr > 0) tile(buf, (sr + 3) >> 2, min(p_Kb, Pc + (sp == 0 ? 0 : p_sr0) + sr), 0, sr);
        lds_barrier();
      }
    }
  }

  STAMP(2);
  // ---- the shared chain, last rows first, 32 rows of the concatenated chain per step (two 16-row MFMA tiles; the step
  // that holds the chain's last rows may be shorter).  Rows travel from global memory straight into one of two LDS
  // buffers (LDS-DMA, no registers): the next step is requested when the matrix cores start on the current one; one
  // LDS-only barrier per step.  Row c of the chain belongs to ancestor t (s_ao[t] <= c < s_ao[t+1]) and has s_ao[t+1]
  // entries; a step's row length Kb is that of its last row, shorter rows are zero-filled up to Kb + 24.
  {
    if constexpr (!ISREF) { if (nit > 0 && !pf) issue(0, arena); }   // (reference quads: requested before the covariance pass)
    int cur = pf ? 1 : 0;
    for (int i = 0; i < nit; ++i) {
      const int c0 = 32 * (nit - 1 - i), sr = min(32, Pc - c0);
      const int kb_v = s_rlen[c0 + sr - 1], kba_v = s_rlen[c0 + min(sr, 16) - 1];   // (both reads in flight together)
      const int Kb = RFL(kb_v), KbA0 = RFL(kba_v);
      double *buf = arena + (size_t)((pf && i == 0) ? 48 : cur * 32) * ldS;
#pragma unroll
      for (int rr = 0; rr < RP; ++rr) rlen_cur[rr] = rlen_nx[rr];   // the lengths of the rows this wave requested for THIS step
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of the current step have landed
#pragma unroll
      for (int rr = 0; rr < RP; ++rr) {
        const int row = wid + NW * rr;
        if (row < sr) {
          const int len = rlen_cur[rr];
          // zero from the row's own end (this also wipes the DMA's odd-length overshoot) to the step's Kb + 24
          if (len + lane < Kb + 24) buf[(size_t)row * ldS + len + lane] = 0.0;   // Kb - len <= 32 (blocks of a quad level)
        } else if (row < (sr > 16 ? 32 : 16)) {
          for (int k = lane; k < Kb + 24; k += 64) buf[(size_t)row * ldS + k] = 0.0;   // absent rows of a tile in use (first step only)
        }
      }
      STAMP(5);
      lds_barrier();
      STAMP(3);
      if (i + 1 < nit) issue(i + 1, arena + (size_t)(cur ^ 1) * 32 * ldS);
      STAMP(6);
      // the chain's factor is lower triangular in chain order: a row's stored length runs to the end of its ancestor's block,
      // but beyond the tile's last row index there are only (explicit) zeros -- 11 % of the V and 16 % of the T MFMAs at 25-row blocks
      compute(buf, sr, min(Kb, c0 + 32), min(KbA0, c0 + 16));
      STAMP(4);
      cur ^= 1;
    }
  }
  lds_barrier();

  // ---- hv = T w_pa for this wave's columns (tile rows l4 + 4 r), summed over the 16 chain columns of a tile row
  double h0 = 0.0, h1 = 0.0, h2 = 0.0, h3 = 0.0;
  {
#pragma unroll
    for (int n = 0; n < NKT; ++n) {
      const int k = n * 16 + l15;
      double wv = k < Pc ? s_wpa[k] : 0.0;
      if constexpr (!ISREF) { if (k >= Pc && k < Pu) wv = s_pw[u][k - Pc]; }
      h0 += tacc[n][0] * wv; h1 += tacc[n][1] * wv; h2 += tacc[n][2] * wv; h3 += tacc[n][3] * wv;
    }
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
      h0 += __shfl_xor(h0, o, 64); h1 += __shfl_xor(h1, o, 64); h2 += __shfl_xor(h2, o, 64); h3 += __shfl_xor(h3, o, 64);
    }
    if (l15 == 0) {
      const int j = jt * 16 + l4;
      s_hv[u][j] = h0; s_hv[u][j + 4] = h1; s_hv[u][j + 8] = h2; s_hv[u][j + 12] = h3;
    }
  }

  STAMP(10);
  double *Ri = arena + (size_t)u * 16 * ldS;   // per unit: Ri [0,1056)  R [1056,2112)  elimination scratch [2112,2328)
  double *R = Ri + 32 * CH_LD;
  double *pub = R + 32 * CH_LD;
  if constexpr (!ISREF) {
    // ---- leaf units: r_j = 1 / sqrt(K_jj - sum_k V[k][j]^2); panel row of column j = [ -r_j T[j][:] | r_j ]
    double dsum = dacc;
    dsum += __shfl_xor(dsum, 16, 64);
    dsum += __shfl_xor(dsum, 32, 64);
    const int jc = jt * 16 + l15;
    double rj = 0.0;
    double dj = 1.0;
    if (jc < Mu) {
      dj = cov_entry(cp, s_colx[u][jc], s_coly[u][jc], s_colmv[u][jc], s_colx[u][jc], s_coly[u][jc], s_colmv[u][jc]) - dsum;
      if (!(dj > 0.0) && !A.predict) s_fail[u] = 1;
      rj = 1.0 / sqrt(dj);
    }
    // hv of column l15 sits in h_{l15 >> 2} of the lanes with l4 == (l15 & 3)
    const int srcl = ((l15 & 3) << 4) | l15;
    const double t0 = __shfl(h0, srcl, 64), t1 = __shfl(h1, srcl, 64), t2 = __shfl(h2, srcl, 64), t3 = __shfl(h3, srcl, 64);
    const double hvc = (l15 >> 2) == 0 ? t0 : ((l15 >> 2) == 1 ? t1 : ((l15 >> 2) == 2 ? t2 : t3));
    if (A.predict) {   // (workgroup-uniform) phase P, spamtree_model.cpp:1306-1326: w_j = H_j w_pa + sqrt(max(K_jj - H_j K_pa,j, 0)) z_j;
      // no panel, no scalars, no failure: prediction blocks own neither
      if (jc < Mu && l4 == 0) {
        const long long r = s_urow0[u] + jc;
        A.w_out[r] = hvc + (dj > 0.0 ? sqrt(dj) : 0.0) * A.z[r];
      }
    } else {
      if (jc < Mu && l4 == 0) {
        const double e = rj * (s_colw[u][jc] - hvc);
        s_e2[u][jc] = e * e;
        s_lg[u][jc] = log(rj);
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = jt * 16 + l4 + 4 * r;
        const double rr = __shfl(rj, l4 + 4 * r, 64);   // lane (0, column) holds that column's r
        if (j < Mu) {
          const int bi = s_colblk[u][j];
          double *prow = A.panels + s_bpan[u][bi] + (size_t)(s_urow0[u] + j - s_brow[u][bi]) * s_bld[u][bi];
#pragma unroll
          for (int n = 0; n < NKT; ++n) {
            const int k = n * 16 + l15;
            if (k < Pu) prow[k] = -rr * tacc[n][r];
          }
          if (l15 == 0) prow[Pu] = rr;
        }
      }
    }
  } else {
    // ---- reference units: R = K_uu - V'V (lower triangle) from the Schur tiles
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = jt * 16 + l4 + 4 * r, j = jt * 16 + l15;
      if (i < Mu && j <= i)
        R[i * CH_LD + j] = cov_entry(cp, s_colx[u][i], s_coly[u][i], s_colmv[u][i], s_colx[u][j], s_coly[u][j], s_colmv[u][j]) - rown[r];
      if (jt == 0) {
        const int i2 = 16 + l4 + 4 * r, j2 = l15;
        if (i2 < Mu)
          R[i2 * CH_LD + j2] = cov_entry(cp, s_colx[u][i2], s_coly[u][i2], s_colmv[u][i2], s_colx[u][j2], s_coly[u][j2], s_colmv[u][j2]) - rcross[r];
      }
    }
  }
  STAMP(11);
  if constexpr (ISREF) {
    // element slots per thread: m (m + 1) <= 128 * slots.  
    if constexpr (WCH) {
      // one wave per unit, registers only (the jt = 0 waves sit on four different SIMDs, which the elimination keeps busy:
      // splitting the columns of L^{-1} over the unit's two waves would put two such waves on every SIMD)
      lds_barrier();   // R complete (both waves of the unit wrote parts of it)
      if (jt == 0) wave_chol_eliminate_blocked<11>(R, Ri, Mu, &s_fail[u], lane);   // 16 + 11 blocked, DPP broadcasts + MFMA coupling (chol_blocked.hpp)
      lds_barrier();
    } else team_chol_eliminate<9>(R, Ri, Mu, Mmax, pub, &s_fail[u], ttid);
    STAMP(7);
    double *xT = Ri + 32 * CH_LD;
    if constexpr (WCH) {
      // ---- N = -Ri T, blocks of <= 27 rows.  The jt = 1 wave owns columns 16 .. 26 only (accumulator registers r = 0 .. 2):
      // it hands ALL its T tiles to its partner in ONE exchange (192 doubles per chain tile, over R and the elimination scratch:
      // quad_lds_stride leaves room for every tile) and the jt = 0 wave forms every tile of N -- rows 0-15 from its own T alone
      // (Ri is lower triangular), rows 16+ from both.  A SIMD hosts one jt = 0 and one jt = 1 wave of different units, so the
      // matrix work per SIMD is what the alternating scheme had (121 against 132 MFMAs), without its second round, its
      // hand-over in both directions and two of its three workgroup barriers; the jt = 1 wave meanwhile writes Ri and forms the
      // residuals e = Ri (w - hv).
      static_assert(NKT * 192 <= 16 * quad_lds_stride(NKX) - 32 * CH_LD, "T hand-over slots must fit the unit's LDS region");
      if (jt == 1) {
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
          if (kt * 16 < Pu) {
            double *sl = xT + kt * 192 + lane;
            sl[0] = tacc[kt][0]; sl[64] = tacc[kt][1]; sl[128] = tacc[kt][2];
          }
        }
      }
      double nri0[2][4], nri1[3];   // -Ri as A operands: [row tile][K-step] over columns 0-15; row tile 1 over columns 16-27
      if (jt == 0) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
          const int i_a = it * 16 + l15;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int j = 4 * r + l4;
            nri0[it][r] = (i_a < Mu && j <= i_a) ? -Ri[i_a * CH_LD + j] : 0.0;
          }
        }
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const int i_a = 16 + l15, j = 16 + 4 * r + l4;
          nri1[r] = (i_a < Mu && j <= i_a) ? -Ri[i_a * CH_LD + j] : 0.0;
        }
      }
      lds_barrier();
      double *pu = A.panels + s_bpan[u][0];
      const int ld = s_bld[u][0];
      if (jt == 0) {
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
          if (kt * 16 < Pu) {
            const double *sl = xT + kt * 192 + lane;
            const double t10 = sl[0], t11 = sl[64], t12 = sl[128];
            const int k = kt * 16 + l15;
            d4 c = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int r = 0; r < 4; ++r) c = __builtin_amdgcn_mfma_f64_16x16x4f64(nri0[0][r], tacc[kt][r], c, 0, 0, 0);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int io = l4 + 4 * r;
              if (io < Mu && k < Pu) pu[(size_t)io * ld + k] = c[r];
            }
            if (two) {
              c = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
              for (int r = 0; r < 4; ++r) c = __builtin_amdgcn_mfma_f64_16x16x4f64(nri0[1][r], tacc[kt][r], c, 0, 0, 0);
              c = __builtin_amdgcn_mfma_f64_16x16x4f64(nri1[0], t10, c, 0, 0, 0);
              if (20 < Mu) c = __builtin_amdgcn_mfma_f64_16x16x4f64(nri1[1], t11, c, 0, 0, 0);
              if (24 < Mu) c = __builtin_amdgcn_mfma_f64_16x16x4f64(nri1[2], t12, c, 0, 0, 0);
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int io = 16 + l4 + 4 * r;
                if (io < Mu && k < Pu) pu[(size_t)io * ld + k] = c[r];
              }
            }
          }
        }
      } else {
        {   // Ri out: lane (row parity, column)
          const int j = lane & 31;
          if (j < Mu)
            for (int i = lane >> 5; i < Mu; i += 2) pu[(size_t)i * ld + Pu + j] = (j <= i) ? Ri[i * CH_LD + j] : 0.0;
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {   // e_i = sum_{j <= i} Ri[i][j] (w_j - hv_j): four lanes per row, the same partial sums as the 128-thread form
          const int i = half * 16 + (lane >> 2), pq = lane & 3;
          double e = 0.0;
          if (i < Mu)
            for (int j = pq; j <= i; j += 4) e += Ri[i * CH_LD + j] * (s_colw[u][j] - s_hv[u][j]);
          e += __shfl_xor(e, 1, 64);
          e += __shfl_xor(e, 2, 64);
          if (i < Mu && pq == 0) {
            s_e2[u][i] = e * e;
            s_lg[u][i] = log(Ri[i * CH_LD + i]);
          }
        }
      }
      STAMP(8);
    } else {
    // ---- N = -Ri T.  Chain tiles alternate between the unit's two waves; the non-owner hands its T tile over through
    // LDS (slots overlay R and the elimination scratch), so the owner holds T for all of the unit's columns.
    constexpr int NR = (NKT + 6) / 7;
    // -Ri as A operands, the same for every chain tile: [row tile it][K-step r] (column tile 0), [r] (row tile 1, column tile 1)
    double nri0[2][4], nri1[4];
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int i_a = it * 16 + l15;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = 4 * r + l4;
        nri0[it][r] = (i_a < Mu && j <= i_a) ? -Ri[i_a * CH_LD + j] : 0.0;
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i_a = 16 + l15, j = 16 + 4 * r + l4;
      nri1[r] = (i_a < Mu && j <= i_a) ? -Ri[i_a * CH_LD + j] : 0.0;
    }
#pragma unroll
    for (int rho = 0; rho < NR; ++rho) {
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        const int kt = rho * 7 + i;
        if (kt < NKT) {
          if (isref && kt * 16 < Pu && (i & 1) != jt) {
            double *sl = xT + i * 256 + lane;
            sl[0] = tacc[kt][0]; sl[64] = tacc[kt][1]; sl[128] = tacc[kt][2]; sl[192] = tacc[kt][3];
          }
        }
      }
      lds_barrier();
#pragma unroll
      for (int i = 0; i < 7; ++i) {
        const int kt = rho * 7 + i;
        if (kt < NKT) {
          if (isref && kt * 16 < Pu && (i & 1) == jt) {
            const double *sl = xT + i * 256 + lane;
            d4 other;
            other[0] = sl[0]; other[1] = sl[64]; other[2] = sl[128]; other[3] = sl[192];
            const d4 T0 = jt == 0 ? tacc[kt] : other, T1 = jt == 0 ? other : tacc[kt];
            double *pu = A.panels + s_bpan[u][0];
            const int ld = s_bld[u][0];
            const int k = kt * 16 + l15;
#pragma unroll
            for (int it = 0; it < 2; ++it) {
              if (it * 16 < Mu) {
                d4 c = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
                for (int r = 0; r < 4; ++r) c = __builtin_amdgcn_mfma_f64_16x16x4f64(nri0[it][r], T0[r], c, 0, 0, 0);
                if (it == 1) {
#pragma unroll
                  for (int r = 0; r < 4; ++r)
                    if (16 + 4 * r < Mu) c = __builtin_amdgcn_mfma_f64_16x16x4f64(nri1[r], T1[r], c, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  const int io = it * 16 + l4 + 4 * r;
                  if (io < Mu && k < Pu) pu[(size_t)io * ld + k] = c[r];
                }
              }
            }
          }
        }
      }
      lds_barrier();
    }
    STAMP(8);
    if (isref) {
      double *pu = A.panels + s_bpan[u][0];
      const int ld = s_bld[u][0];
      {   // Ri out: thread (row group, column), no division by the block size
        const int j = ttid & 31;
        if (j < Mu)
          for (int i = ttid >> 5; i < Mu; i += 4) pu[(size_t)i * ld + Pu + j] = (j <= i) ? Ri[i * CH_LD + j] : 0.0;
      }
      {   // e_i = sum_{j <= i} Ri[i][j] (w_j - hv_j): four threads per row, partial sums joined by two shuffles
        const int i = ttid >> 2, pq = ttid & 3;
        double e = 0.0;
        if (i < Mu)
          for (int j = pq; j <= i; j += 4) e += Ri[i * CH_LD + j] * (s_colw[u][j] - s_hv[u][j]);
        e += __shfl_xor(e, 1, 64);
        e += __shfl_xor(e, 2, 64);
        if (i < Mu && pq == 0) {
          s_e2[u][i] = e * e;
          s_lg[u][i] = log(Ri[i * CH_LD + i]);
        }
      }
    }
    }   // !WCH
  }
  STAMP(12);
  lds_barrier();
  STAMP(9);
  // ---- per-block scalars (logdetCi_comps, loglik_w_comps) and the failure word
  if constexpr (ISREF) {
    if (ttid == 0 && Mu > 0) {
      double wc = 0.0, ldt = 0.0;
      for (int i = 0; i < Mu; ++i) { wc += s_e2[u][i]; ldt += s_lg[u][i]; }
      A.logdet_c[s_ublk0[u]] = ldt;
      A.loglik_c[s_ublk0[u]] = (double)Mu * HL2PI - 0.5 * wc;
      if (s_fail[u]) atomicMin(A.errflag, s_level * 16 + (s_uJ[u] == 0 ? 1 : 2));
    }
  } else if (!A.predict) {
    if (ttid < s_unblk[u]) {
      const int bi = ttid;
      double wc = 0.0, ldt = 0.0;
      // the block's columns are consecutive (device rows s_brow[bi] .. s_brow[bi + 1] - 1 of the unit): walk those only, in
      // the same ascending order as a scan of all Mu columns would (3.5 % of a leaf quad's life went into that scan)
      const int j0 = (int)(s_brow[u][bi] - s_urow0[u]);
      const int j1 = bi + 1 < s_unblk[u] ? (int)(s_brow[u][bi + 1] - s_urow0[u]) : Mu;
      const int cnt = j1 - j0;
      for (int j = j0; j < j1; ++j) { wc += s_e2[u][j]; ldt += s_lg[u][j]; }
      A.logdet_c[s_ublk0[u] + bi] = ldt;
      A.loglik_c[s_ublk0[u] + bi] = (double)cnt * HL2PI - 0.5 * wc;
    }
    if (ttid == 0 && Mu > 0 && s_fail[u]) atomicMin(A.errflag, s_level * 16 + 3);
  }
  STAMP(13);
  STAMP_FLUSH_LEVEL(s_level);
}
template __global__ void k_factor_quad<4, 32, 8, true, true>(QuadArgs, CovPar);
template __global__ void k_factor_quad<4, 32, 8, true, false>(QuadArgs, CovPar);
template __global__ void k_factor_quad<4, 32, 8, false, true>(QuadArgs, CovPar);
template __global__ void k_factor_quad<4, 38, 10, true, true>(QuadArgs, CovPar);
template __global__ void k_factor_quad<4, 38, 10, true, false>(QuadArgs, CovPar);
template __global__ void k_factor_quad<4, 38, 10, false, true>(QuadArgs, CovPar);
template __global__ void k_factor_quad<4, 44, 11, true, true>(QuadArgs, CovPar);
template __global__ void k_factor_quad<4, 44, 11, true, false>(QuadArgs, CovPar);
template __global__ void k_factor_quad<4, 44, 11, false, true>(QuadArgs, CovPar);
template __global__ void k_factor_quad<4, 50, 13, true, true>(QuadArgs, CovPar);
template __global__ void k_factor_quad<4, 50, 13, true, false>(QuadArgs, CovPar);
template __global__ void k_factor_quad<4, 50, 13, false, true>(QuadArgs, CovPar);
#else   // host side: prototypes only
template <int NU, int NKX, int NKT, bool ISREF, bool WCH> __global__ void k_factor_quad(QuadArgs A, CovPar cp);
#endif
