// Included by spamtree_hip.hip after factor_wide.hpp (needs FactorArgs, Blk, CovPar, cov_entry, d4, dma typedefs, lds_barrier).
#pragma once

// Phase A of NON-REFERENCE blocks behind long ancestor chains (the leaves of the default multivariate tree, config #4:
// 36-column blocks, chains of 525 rows; spamtree_model.cpp:923-963), third generation.  k_factor_bigmfma / k_factor_wide
// keep K_{pa,u} and V in a global scratch slice and feed the matrix cores with 8-byte loads from L2 (56 % of the leaf level),
// with three workgroup barriers per 16 chain rows.  Here a workgroup is a slab of up to 64 columns of a sibling group (LcSlab)
// and a wave owns 16 of its columns over the
// WHOLE chain, as in k_factor_quad -- but a 525-row chain leaves no room for K AND the T = H_u accumulators in one wave's
// registers, so the chain's inverse Cholesky factor (one lower-triangular P x P matrix kept as one row panel per ancestor)
// is streamed through LDS TWICE and T is never held:
//   phase 1, by 16-row tiles, last rows first:  V_r = Linv[r, 0..r] K   (A from LDS, B = kx: K in REGISTERS, evaluated once
//            from the coordinates); V_r overwrites K's rows of tile r in place (the accumulator layout of a V tile IS the
//            B-operand layout of those four K-steps; later tiles need only K's rows above), sum_k V^2 on the fly;
//   r_j = 1 / sqrt(K_jj - sum_k V_kj^2)  -- complete before any T entry exists;
//   phase 2, by 16-column blocks of Linv:  T[:, kt] = sum_{r >= kt} V_r' Linv[r, kt]  (A = V from registers, B from LDS);
//            each finished tile leaves as panel entries -r_j T and as its share of hv = T w_pa: no T accumulators.
// Only structurally non-zero tiles are touched (Linv is lower triangular in chain order): 2 x (NTL (NTL + 1) / 2) x 4 MFMAs per
// wave for NTL = ceil(P / 16) row tiles.  One wave per SIMD (4 waves, up to 512 registers each: K alone is P / 2 VGPRs), one
// workgroup per CU; both phases double-buffer their LDS-DMA under the matrix work; one barrier per tile.
// P <= 4 NKX, m <= 64 columns, J >= 1 ancestors (host check).  Results equal k_factor_bigmfma's up to summation order.
#define LC_NT 256
#ifndef LC_RSH
#define LC_RSH 17
#endif
// LC_RSH: phase 1: tiles up to this index are staged by all four waves (above: by the loader waves alone)
#ifndef LC_G
#define LC_G 2
#endif
// LC_G: phase 2 takes its row tiles in groups of LC_G (one compare + branch + operand hand-over per group instead of per tile); a
// group that starts above the column block's diagonal tile multiplies the rows in between by zeros the loader put there

__host__ __device__ constexpr int lc_lds_stride(int nkx) {   // phase-1 row stride: >= 4 nkx + 24, 2 x odd (conflict-free A reads)
  int s = 4 * nkx + 24;
  while ((s & 1) || ((s >> 1) & 1) == 0) ++s;
  return s;
}
__host__ __device__ constexpr size_t lc_dyn_doubles(int nkx) {   // dynamic LDS: two buffers, max over the phases
  const size_t a = (size_t)2 * 16 * lc_lds_stride(nkx), b = (size_t)2 * (4 * nkx + 16) * 16;
  return a > b ? a : b;
}

// cov_entry with the per-pair constants in LDS (tab: rate | amp | amp2 | phi, QMAX^2 / QMAX entries each).  MV: branch-free
// multivariate form (the second exponential is always evaluated and enters with amplitude amp2 = 0 where the reference has
// no such term: the same value) so that several independent evaluations can be interleaved; !MV: cexpcov.
struct LcSlab {
  long long row0;   // first device row (column) of the slab
  long long pan0;   // panel arena offset of that row's panel row (rows of the group follow each other with stride ld)
  int blk0;         // a block of the group: chain, level
  int ncol, ld, vcol0;
  long long vs0;    // reference levels: the V scratch of the group's first block (the blocks' matrices follow each other); vcol0 = the
                    // slab's first column counted over the group (blocks of a group are equally wide)
};

struct LcArgs {
  const Blk *blks;
  const int *anc_idx;
  const LcSlab *slabs;
  int nslab;
  const double *cx, *cy;
  const int *mv;
  const double *w_in;
  double *panels;
  double *rowtmp;   // per device row: e_j^2 | log r_j (2 x n_rows), summed per block by k_lchain_scalars in row order
  long long n_rows;
  int *errflag;
  double *vscr;     // reference levels: V = Linv_pa K_pa,u goes here for k_factor_ref_finish's Schur complement (nullptr: not stored)
  int errcode;      // what a non-positive conditional variance reports: 3 on non-reference levels (spamtree_model.cpp:958); 2 when the
                    // kernel runs as the first half of a REFERENCE level (k_factor_ref_finish completes it; :919)
};

// k_factor_ref_finish (second half of a REFERENCE level on this route): the Schur complement R = K_uu - V'V walks the chain in
// chunks of RF_KC rows of V.  k_factor_lchain leaves V per block as a [P rounded up to 4][RF_LDB] row-major matrix in a scratch, so
// that a chunk is ONE contiguous run that LDS-DMA copies verbatim (no registers, no LDS stores) into a ring of RF_NBUF buffers
// overlaid on R / Ri (free until the product is complete): three chunks are in flight while one is multiplied.
#define RF_KC 32
#define RF_LDB 80      // row stride of V in the scratch and in LDS: 80 = 16 mod 32 (the four K rows of an operand fall into alternate bank halves)
#define RF_NBUF 4
#define RF_BUFD 3072   // doubles per buffer: 24 pieces of 1 KB = three per wave (the chunk's 20 KB and the first rows of the next one)
__host__ __device__ constexpr size_t rf_work_doubles(int maxM) {
  const size_t a = (size_t)2 * maxM * maxM, b = (size_t)RF_NBUF * RF_BUFD;
  return a > b ? a : b;
}
#define RF_NOPS 60     // A operands of N = -Ri T: (row tile it, K-step s2 < 4 (it + 1)), it < 5, at 2 it (it + 1) + s2
__host__ __device__ constexpr size_t rf_lds_bytes(int maxM) {   // w, x, y, T w_pa, 1 / r of the block | work | -Ri in operand order | outcome ids
  return (5 * (size_t)maxM + rf_work_doubles(maxM) + RF_NOPS * 64) * 8 + (size_t)((maxM + 1) & ~1) * 4 + 64;
}
__host__ __device__ constexpr long long rf_vsize(int P) { return (long long)((P + 3) & ~3) * RF_LDB; }   // a block's V in the scratch (doubles)

#ifdef ST_DEFS_FACTOR_WIDE
#define LC_CPT (3 * QMAX * QMAX + QMAX + 64)   // the table: rate | amp | amp2 per outcome pair, phi per outcome, 2^(j/64) for cov_exp_tab
template <bool MV>
__device__ __forceinline__ double lc_cov(const double *tab, int q, double s2, double nphi, double xi, double yi, int vi, double xj, double yj, int vj) {
  const double dx = xi - xj, dy = yi - yj;
  const double h = cov_sqrt(dx * dx + dy * dy);
  const double *e64 = tab + 3 * QMAX * QMAX + QMAX;
  if constexpr (!MV) return s2 * cov_exp_tab(nphi * h, e64);
  const int ij = vi * q + vj;
  const double r = tab[QMAX * QMAX + ij] * cov_exp_tab(-tab[ij] * h, e64);
  const double a2 = tab[2 * QMAX * QMAX + ij];
  const double e2 = cov_exp_tab(-tab[3 * QMAX * QMAX + vi] * h, e64);
  return a2 != 0.0 ? fma(a2, e2, r) : r;   // (a select, not a branch; r + a2 e2 rounds as the reference's sum does)
}
__device__ __forceinline__ void lc_cov_table(double *tab, const CovPar &cp, int tid, int nt) {
  for (int i = tid; i < LC_CPT; i += nt) {
    const int a = i / (QMAX * QMAX), ij = i - a * (QMAX * QMAX);
    tab[i] = i >= 3 * QMAX * QMAX + QMAX ? EXP2_64[i - (3 * QMAX * QMAX + QMAX)] : (a == 0 ? cp.rate[ij] : (a == 1 ? cp.amp[ij] : (a == 2 ? cp.amp2[ij] : cp.phi[ij])));
  }
}

// A workgroup's columns: a SLAB of a sibling group -- up to 64 consecutive columns of the concatenated rows of consecutive
// blocks with one parent (= one chain; device order keeps siblings, their rows and their panels contiguous).  Tiles of 16
// columns may straddle blocks: four 36-column leaves are nine full tiles instead of twelve (16 + 16 + 4 each).

// per-block scalars from the per-row values (fixed order: independent of the slabs' composition and of the launch geometry)
__global__ void k_lchain_scalars(const Blk *blks, const int *list, int nlist, const double *rowtmp, long long n_rows, double *logdet_c, double *loglik_c) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= nlist) return;
  const int b = list[i];
  const Blk B = blks[b];
  double wc = 0.0, ldt = 0.0;
  for (int j = 0; j < B.m; ++j) { wc += rowtmp[B.row0 + j]; ldt += rowtmp[n_rows + B.row0 + j]; }
  logdet_c[b] = ldt;
  loglik_c[b] = (double)B.m * HL2PI - 0.5 * wc;
}

template <int NKX>
__global__ __launch_bounds__(LC_NT) void k_factor_lchain(LcArgs A, CovPar cp) {
  constexpr int PMAX = 4 * NKX, NTMAX = (PMAX + 15) / 16, ldS = lc_lds_stride(NKX);
  constexpr int B1 = 16 * ldS;             // doubles per phase-1 buffer (one 16-row tile, full length)
  constexpr int B2 = (PMAX + 16) * 16;     // doubles per phase-2 buffer (one 16-column block, row c at c * 16)
  constexpr int KH = (B1 - (5 * PMAX) / 2 - 8) / 256;   // covariance scratch: K-steps per pass ([KH][64] doubles per wave)
  constexpr int NPASS = (NKX + KH - 1) / KH;
  static_assert(KH >= 8, "covariance scratch too small");
  static_assert(NTMAX % LC_G == 0, "phase 2 takes its row tiles in whole groups");
  extern __shared__ double lds[];
  __shared__ int s_am[MAXJ], s_ao[MAXJ + 1];
  __shared__ long long s_arow[MAXJ], s_apan[MAXJ];
  __shared__ double s_wpa[PMAX];
  __shared__ long long s_rsrc[PMAX];   // chain row c: where it starts in the panel arena ...
  __shared__ int s_rlen[PMAX];         // ... and its length (entries up to the end of its own ancestor's rows)
  __shared__ double s_cpt[LC_CPT];   // rate, amp, amp2 per outcome pair, phi per outcome, the exp table: the covariance pass reads
                                                     // them from LDS (as kernel arguments indexed per lane they are global loads,
                                                     // one dependent round trip per entry with a single wave per SIMD)
  __shared__ int s_fail;

  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  int li = blockIdx.x;
  {
    const int per = A.nslab >> 3;   // one contiguous run of slabs per XCD (private L2): siblings share their chain's panels
    if (li < per * 8) li = (li & 7) * per + (li >> 3);
  }
  const LcSlab S = A.slabs[li];
  const Blk B = A.blks[S.blk0];
  const int m = S.ncol, P = B.P, J = B.nanc;
  const int NTL = (P + 15) >> 4;

  for (int i = tid; i < (int)lc_dyn_doubles(NKX); i += LC_NT) lds[i] = 0.0;   // never NaN garbage under a zero multiplier
  if (tid < J) {
    const int a = A.anc_idx[B.anc_ptr + tid];
    s_am[tid] = A.blks[a].m; s_arow[tid] = A.blks[a].row0; s_apan[tid] = A.blks[a].chain_off;
  }
  if (tid == 0) s_fail = 0;
  lc_cov_table(s_cpt, cp, tid, LC_NT);
  STAMP_DECL
  __syncthreads();
  if (tid == 0) {
    int o = 0;
    for (int t = 0; t < J; ++t) { s_ao[t] = o; o += s_am[t]; }
    s_ao[J] = o;
  }
  __syncthreads();
  // coordinates of the chain -> the second buffer's first part (free until step 1 is requested), w of the chain, row table
  double *sx = lds + B1, *sy = sx + PMAX;
  int *smv = (int *)(sy + PMAX);
  for (int k = tid; k < P; k += LC_NT) {
    int t = 0;
    while (t + 1 < J && k >= s_ao[t + 1]) ++t;
    const long long r = s_arow[t] + (k - s_ao[t]);
    sx[k] = A.cx[r]; sy[k] = A.cy[r]; smv[k] = A.mv[r]; s_wpa[k] = A.w_in[r];
    const int len = s_ao[t + 1];
    s_rlen[k] = len; s_rsrc[k] = s_apan[t] + (long long)(k - s_ao[t]) * len;
  }
  __syncthreads();

  // phase-1 staging: tile r = chain rows [16 r, 16 r + 16), moved in pieces of 128 doubles.
  // One address and one LDS base (M0) per row: the pieces differ by the instruction's immediate offset, which advances the
  // global and the LDS address alike.  Whole pieces are fetched (up to 127 doubles past the row's end: its successor in the
  // arena; the padding pass wipes what the tile can read of them) -- only a fifth piece is cut at the row's end, because it
  // would run into the next staged row.
  // Who stages: a block of <= 48 columns leaves waves without columns -- THEY move the data (loader waves: requests, landing
  // wait, padding), the column waves only meet them at the barrier; a CU's LDS-DMA path takes about 23 B per cycle, and a wave
  // that requests blocks on it: 14 % of a column wave's time when everybody requests.  With four column waves everybody
  // stages a quarter, as before.
  const int JTb = (m + 15) >> 4;
  const int nload = JTb < 4 ? 4 - JTb : 4, lidx = JTb < 4 ? wid - JTb : wid;
  const bool isload = lidx >= 0;
  // Tiles r <= LC_RSH are requested (and padded) by ALL FOUR waves, four rows each: their matrix work is shorter than one
  // wave's sixteen rows of requests (about 2.4k cycles of table reads, address set-up and DMA issue), so the column waves
  // would only wait for the loader at the barrier -- 14 % of their time before.
  auto issue1 = [&](int r, double *buf, int li_, int nl_) {
    for (int row = li_; row < 16; row += nl_) {
      const int c = min(16 * r + row, P - 1);   // rows beyond the chain (last tile): any row, zero-filled afterwards
      const int ln = min(__builtin_amdgcn_readfirstlane(s_rlen[c]), 16 * (r + 1));   // the tile reads columns < 16 (r + 1) only
      const double *srow = A.panels + s_rsrc[c] + 2 * lane;
      q_lds_void *dst = (q_lds_void *)(buf + (size_t)row * ldS);
      q_glb_void *sp = (q_glb_void *)srow;
      __builtin_amdgcn_global_load_lds(sp, dst, 16, 0, 0);
      if (ln > 128) __builtin_amdgcn_global_load_lds(sp, dst, 16, 1024, 0);
      if (ln > 256) __builtin_amdgcn_global_load_lds(sp, dst, 16, 2048, 0);
      if (ln > 384) __builtin_amdgcn_global_load_lds(sp, dst, 16, 3072, 0);
      if (ln > 512) {
        if (512 + 2 * lane < ln)
          __builtin_amdgcn_global_load_lds((q_glb_void *)(srow + 512), (q_lds_void *)(buf + (size_t)row * ldS + 512), 16, 0, 0);
      }
    }
  };
  STAMP(0);
  if (JTb == 4 || NTL - 1 <= LC_RSH) issue1(NTL - 1, lds, wid, 4);   // lands under the covariance pass
  else if (isload) issue1(NTL - 1, lds, lidx, nload);

  // ---- K_{pa,u}: kx[st] = K[4 st + l4][column 16 wid + l15]  (covariance_functions.cpp:95-111 / :213-286), rolled loop
  // through lane-private LDS slots, picked up with static register indices
  const int jc = 16 * wid + l15;
  const bool cok = jc < m;
  const bool wact = 16 * wid < m;   // this wave owns at least one column
  const long long jrow = S.row0 + min(jc, m - 1);
  const double mx = A.cx[jrow], my = A.cy[jrow], wj = A.w_in[jrow];
  const int mvj = A.mv[jrow];
  double kx[NKX];
  const int qn = cp.q;
  const double s2 = cp.ai1[0], nphi = -cp.tmv[0];
  {
    double *kb = sy + PMAX + (PMAX + 1) / 2 + 2 + (size_t)wid * (KH * 64) + lane;
#pragma unroll
    for (int hp = 0; hp < NPASS; ++hp) {
      const int st0 = hp * KH;
      if (4 * st0 < P && wact) {
        // four independent chains per trip: one wave per SIMD has nobody else to hide FP64 latency
#define LC_COVLOOP(MV_)                                                                                         \
  _Pragma("unroll 1") for (int i = 0; i < KH; i += 4) {                                                         \
    double v[4];                                                                                                \
    _Pragma("unroll") for (int e = 0; e < 4; ++e) {                                                             \
      const int k = 4 * (st0 + i + e) + l4, kc = min(k, PMAX - 1);                                              \
      const double c0 = lc_cov<MV_>(s_cpt, qn, s2, nphi, sx[kc], sy[kc], smv[kc], mx, my, mvj);                 \
      v[e] = (cok && k < P) ? c0 : 0.0;                                                                         \
    }                                                                                                           \
    _Pragma("unroll") for (int e = 0; e < 4; ++e)                                                               \
      if (i + e < KH) kb[(i + e) * 64] = v[e];                                                                  \
  }
        if (qn == 1) { LC_COVLOOP(false) } else { LC_COVLOOP(true) }
#undef LC_COVLOOP
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

  STAMP(1);
#define LCMFMA(a_, b_, c_) c_ = __builtin_amdgcn_mfma_f64_16x16x4f64(a_, b_, c_, 0, 0, 0)
  double dacc = 0.0;   // sum_k V[k][column l15]^2 over this lane's rows
  double rj = 0.0, rq[4] = {0.0, 0.0, 0.0, 0.0};
  double h0 = 0.0, h1 = 0.0, h2 = 0.0, h3 = 0.0;   // hv = T w_pa for columns 4 q + l4, this lane's chain columns
  // phase 2, LC_G > 1: the row tiles between the start of the diagonal tile's group and the diagonal tile itself (structural zeros of
  // the factor's column block kt, not part of what is staged: the buffer holds an older block's rows there) are cleared by the
  // loaders, 8-row group g by loader g % nload
  auto zero_above = [&](int kt, double *buf) {
    if (LC_G > 1) {
      for (int g = 2 * (kt - kt % LC_G); g < 2 * kt; ++g)
        if (lidx == g % nload) {
          double *e = buf + (size_t)(8 * g + (lane >> 3)) * 16 + 2 * (lane & 7);
          e[0] = 0.0; e[1] = 0.0;
        }
    }
  };
  if (isload && JTb < 4) {
    // ---- a LOADER wave (no columns): both phases as ROLLED loops of their own -- it needs no static tile index -- that meet the
    // column waves at the same barriers.  What bounds a step is this wave's round trip (request, flight, padding), not the
    // matrix work: without a single MFMA the kernel took 70 % of its time.  The requests therefore take their table entries four
    // rows / eight row groups at a time (one LDS latency per batch instead of one per row).
    auto fire1 = [&](int r, double *buf) {
      const int li_ = r <= LC_RSH ? wid : lidx, nl_ = r <= LC_RSH ? 4 : nload;   // short tiles: this wave's quarter only
      for (int row0 = li_; row0 < 16; row0 += 4 * nl_) {
        int ln[4];
        const double *src[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int c = min(16 * r + min(row0 + j * nl_, 15), P - 1);
          ln[j] = s_rlen[c];
          src[j] = A.panels + s_rsrc[c] + 2 * lane;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int row = row0 + j * nl_;
          if (row < 16) {
            const int lu = min(__builtin_amdgcn_readfirstlane(ln[j]), 16 * (r + 1));
            q_lds_void *dst = (q_lds_void *)(buf + (size_t)row * ldS);
            q_glb_void *sp = (q_glb_void *)src[j];
            __builtin_amdgcn_global_load_lds(sp, dst, 16, 0, 0);
            if (lu > 128) __builtin_amdgcn_global_load_lds(sp, dst, 16, 1024, 0);
            if (lu > 256) __builtin_amdgcn_global_load_lds(sp, dst, 16, 2048, 0);
            if (lu > 384) __builtin_amdgcn_global_load_lds(sp, dst, 16, 3072, 0);
            if (lu > 512) {
              if (512 + 2 * lane < lu)
                __builtin_amdgcn_global_load_lds((q_glb_void *)(src[j] + 512), (q_lds_void *)(buf + (size_t)row * ldS + 512), 16, 0, 0);
            }
          }
        }
      }
    };
    int cur = 0;
    for (int r = NTL - 1; r >= 0; --r) {
      double *buf = lds + (size_t)cur * B1;
      STAMP(15);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      STAMP(12);
      // padding, eight rows per trip: one LDS latency per trip instead of one dependent table read per row
      const int li_ = r <= LC_RSH ? wid : lidx, nl_ = r <= LC_RSH ? 4 : nload;
      for (int row0 = li_; row0 < 16; row0 += 8 * nl_) {
        int ln[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) ln[j] = s_rlen[min(16 * r + min(row0 + j * nl_, 15), P - 1)];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int row = row0 + j * nl_, c = 16 * r + row;
          if (row < 16) {
            if (c < P) {
              if (ln[j] + lane < 16 * (r + 1)) buf[(size_t)row * ldS + ln[j] + lane] = 0.0;
            } else {
              for (int k = lane; k < 16 * (r + 1); k += 64) buf[(size_t)row * ldS + k] = 0.0;
            }
          }
        }
      }
      STAMP(13);
      lds_barrier();
      STAMP(14);
      if (r > 0) fire1(r - 1, lds + (size_t)(cur ^ 1) * B1);
      cur ^= 1;
    }
    lds_barrier();   // phase boundary
    auto fire2 = [&](int kt, double *buf) {   // 8-row group g belongs to loader g % nload; eight groups per trip
      const int g1 = (P + 7) >> 3, g0 = 2 * kt;
      const int koff = 16 * kt + 2 * (lane & 7);
      int g = g0 + lidx - (g0 % nload);
      if (g < g0) g += nload;
      for (; g < g1; g += 8 * nload) {
        long long so[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) so[e] = s_rsrc[min(8 * min(g + e * nload, g1 - 1) + (lane >> 3), P - 1)];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int ge = g + e * nload;
          if (ge < g1)
            __builtin_amdgcn_global_load_lds((q_glb_void *)(A.panels + so[e] + koff), (q_lds_void *)(buf + (size_t)ge * 128), 16, 0, 0);
        }
      }
    };
    fire2(0, lds);
    cur = 0;
    for (int kt = 0; kt < NTL; ++kt) {
      double *buf = lds + (size_t)cur * B2;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int g = 2 * kt + hh;
        if (lidx == g % nload) {
          const int i = 8 * hh + (lane >> 3), j0 = 2 * (lane & 7);
          double *e = buf + (size_t)(16 * kt + i) * 16 + j0;
          if (j0 > i) e[0] = 0.0;
          if (j0 + 1 > i) e[1] = 0.0;
        }
      }
      zero_above(kt, buf);
      lds_barrier();
      if (kt + 1 < NTL) fire2(kt + 1, lds + (size_t)(cur ^ 1) * B2);
      cur ^= 1;
    }
  } else {
    // ---- phase 1: V_r = Linv[r, 0 .. r] K, last tile first; V_r replaces kx[4 r .. 4 r + 3]
    {
      int cur = 0;
  #pragma unroll
      for (int r = NTMAX - 1; r >= 0; --r) {
        if (r < NTL) {   // workgroup-uniform
          double *buf = lds + (size_t)cur * B1;
          if (JTb == 4 || r <= LC_RSH) {   // (a column wave takes part in the staging of the short tiles only)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of tile r have landed
            for (int row = wid; row < 16; row += 4) {
              const int c = 16 * r + row;
              if (c < P) {
                const int len = __builtin_amdgcn_readfirstlane(s_rlen[c]);
                // the tile reads columns < 16 (r + 1) only; zero from the row's own end (what was fetched past it)
                if (len + lane < 16 * (r + 1)) buf[(size_t)row * ldS + len + lane] = 0.0;
              } else {
                for (int k = lane; k < 16 * (r + 1); k += 64) buf[(size_t)row * ldS + k] = 0.0;   // rows beyond the chain (last tile)
              }
            }
          }
          STAMP(2);
          lds_barrier();
          STAMP(3);
          if (r > 0 && (JTb == 4 || r - 1 <= LC_RSH)) issue1(r - 1, lds + (size_t)(cur ^ 1) * B1, wid, 4);
          STAMP(4);
          if (wact) {
            d4 p = (d4){0.0, 0.0, 0.0, 0.0};
            const double *ap = buf + l15 * ldS + l4;
  #pragma unroll
            for (int st = 0; st < 4 * (r + 1); ++st)
              if (st < NKX) LCMFMA(ap[4 * st], kx[st], p);
            dacc += p[0] * p[0] + p[1] * p[1] + p[2] * p[2] + p[3] * p[3];
  #pragma unroll
            for (int q = 0; q < 4; ++q)
              if (4 * r + q < NKX) kx[4 * r + q] = p[q];
          }
          STAMP(5);
          cur ^= 1;
        }
      }
    }
    lds_barrier();   // phase 2 reuses the buffers
    if (A.vscr && cok) {   // reference level: V (the B-operand layout: kx[st] = V[4 st + l4][column]) for k_factor_ref_finish; rows up
                           // to the next multiple of four are exact zeros and are written too (its last K-step reads them)
      const int gcol = S.vcol0 + jc, bi = gcol / B.m;
      double *vp = A.vscr + S.vs0 + bi * rf_vsize(P) + (gcol - bi * B.m) + (size_t)l4 * RF_LDB;
#pragma unroll
      for (int st = 0; st < NKX; ++st)
        if (4 * st < P) vp[(size_t)4 * st * RF_LDB] = kx[st];
    }
    STAMP(6);

    // ---- r_j = 1 / sqrt(K_jj - sum_k V_kj^2)  (spamtree_model.cpp:944-951)
    {
      double dsum = dacc;
      dsum += __shfl_xor(dsum, 16, 64);
      dsum += __shfl_xor(dsum, 32, 64);
      if (cok) {
        const double d = (qn == 1 ? lc_cov<false>(s_cpt, qn, s2, nphi, mx, my, mvj, mx, my, mvj) : lc_cov<true>(s_cpt, qn, s2, nphi, mx, my, mvj, mx, my, mvj)) - dsum;
        if (!(d > 0.0)) s_fail = 1;
        rj = 1.0 / sqrt(d);
      }
    }
    // rq[q]: r of column 16 wid + 4 q + l4 (lane (0, column) holds that column's r)
  #pragma unroll
    for (int q = 0; q < 4; ++q) rq[q] = __shfl(rj, l4 + 4 * q, 64);

    // ---- phase 2: T[:, kt] = sum_{r >= kt} V_r' Linv[r, kt]; column block kt of Linv = chain rows [16 kt, P) x 16 columns,
    // row c at offset c * 16 of the buffer (static operand offsets); one LDS-DMA instruction moves 8 rows (8 lanes x 16 B each)
    auto issue2 = [&](int kt, double *buf) {   // 8-row group g belongs to loader g % nload
      const int g1 = (P + 7) >> 3, g0 = 2 * kt;
      const int koff = 16 * kt + 2 * (lane & 7);
      int g = g0 + lidx - (g0 % nload);
      if (g < g0) g += nload;
      for (; g < g1; g += nload) {   // rows beyond the chain: any row (their V rows are zero)
        const double *rp = A.panels + s_rsrc[min(8 * g + (lane >> 3), P - 1)] + koff;
        __builtin_amdgcn_global_load_lds((q_glb_void *)rp, (q_lds_void *)(buf + (size_t)g * 128), 16, 0, 0);
      }
    };
    {
      double *pu = A.panels + S.pan0;
      const int ld = S.ld;
      if (isload) issue2(0, lds);
      int cur = 0;
      for (int kt = 0; kt < NTL; ++kt) {
        double *buf = lds + (size_t)cur * B2;
        if (isload) {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          // the diagonal tile's entries above the diagonal are structural zeros, but rows that END inside this column block were
          // fetched past their end: wipe (the loader of the 8-row group does it, after its own data has landed)
  #pragma unroll
          for (int hh = 0; hh < 2; ++hh) {
            const int g = 2 * kt + hh;
            if (lidx == g % nload) {
              const int i = 8 * hh + (lane >> 3), j0 = 2 * (lane & 7);
              double *e = buf + (size_t)(16 * kt + i) * 16 + j0;
              if (j0 > i) e[0] = 0.0;
              if (j0 + 1 > i) e[1] = 0.0;
            }
          }
          zero_above(kt, buf);
        }
        STAMP(7);
        lds_barrier();
        STAMP(8);
        if (kt + 1 < NTL && isload) issue2(kt + 1, lds + (size_t)(cur ^ 1) * B2);
        STAMP(9);
        if (wact) {
          d4 t = (d4){0.0, 0.0, 0.0, 0.0};
          const double *bp = buf + l4 * 16 + l15;
          // the B operands of row tile r + 1 are requested BEFORE the MFMAs of row tile r (two register sets, taken in turn; a
          // scheduling barrier keeps the requests in front): every row tile is a basic block of its own (skipped for r < kt),
          // and the compiler neither moves loads across those nor hoists them above MFMAs that still read the same registers
          double b0[4 * LC_G], b1[4 * LC_G];
          const int r0 = kt - kt % LC_G;   // the group of the diagonal tile
          if ((r0 / LC_G) & 1) {
  #pragma unroll
            for (int q = 0; q < 4 * LC_G; ++q) { b1[q] = bp[(size_t)(16 * r0 + 4 * q) * 16]; b0[q] = 0.0; }
          } else {
  #pragma unroll
            for (int q = 0; q < 4 * LC_G; ++q) { b0[q] = bp[(size_t)(16 * r0 + 4 * q) * 16]; b1[q] = 0.0; }
          }
  #pragma unroll
          for (int r = 0; r < NTMAX; r += LC_G) {
            // (a computed entry -- switch (kt) with fall-through cases -- would save the compare + branch of every skipped group,
            // but the register allocator then needs > 512 VGPRs: measured, dropped)
            if (r + LC_G - 1 >= kt && r < NTL) {   // workgroup-uniform
              __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): this group's operands (requested one group ago) are here; said
                                                    // explicitly (and visibly to the compiler), or it waits AFTER the new requests
              if (r + LC_G < NTMAX) {      // inside the buffer (B2 holds PMAX + 16 rows)
  #pragma unroll
                for (int q = 0; q < 4 * LC_G; ++q) {
                  if (((r / LC_G) & 1) == 0) b1[q] = bp[(16 * (r + LC_G) + 4 * q) * 16]; else b0[q] = bp[(16 * (r + LC_G) + 4 * q) * 16];
                }
              }
              __builtin_amdgcn_sched_barrier(0);
  #pragma unroll
              for (int q = 0; q < 4 * LC_G; ++q)
                if (4 * r + q < NKX) LCMFMA(kx[4 * r + q], ((r / LC_G) & 1) == 0 ? b0[q] : b1[q], t);
            }
          }
          const int k = 16 * kt + l15;
          if (k < P) {
            const double wv = s_wpa[k];
            h0 = fma(t[0], wv, h0); h1 = fma(t[1], wv, h1); h2 = fma(t[2], wv, h2); h3 = fma(t[3], wv, h3);
  #pragma unroll
            for (int q = 0; q < 4; ++q) {
              const int j = 16 * wid + 4 * q + l4;
              if (j < m) pu[(size_t)j * ld + k] = -rq[q] * t[q];
            }
          }
        }
        STAMP(10);
        cur ^= 1;
      }
    }
  }
#undef LCMFMA
  // ---- per-row scalars: e_j = r_j (w_j - hv_j), log r_j; per-block sums in a fixed order
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) {
    h0 += __shfl_xor(h0, o, 64); h1 += __shfl_xor(h1, o, 64); h2 += __shfl_xor(h2, o, 64); h3 += __shfl_xor(h3, o, 64);
  }
  {
    // hv of column l15 sits in h_{l15 >> 2} of the lanes with l4 == (l15 & 3)
    const int srcl = ((l15 & 3) << 4) | l15;
    const double t0 = __shfl(h0, srcl, 64), t1 = __shfl(h1, srcl, 64), t2 = __shfl(h2, srcl, 64), t3 = __shfl(h3, srcl, 64);
    const double hvc = (l15 >> 2) == 0 ? t0 : ((l15 >> 2) == 1 ? t1 : ((l15 >> 2) == 2 ? t2 : t3));
    if (cok && l4 == 0) {
      const double e = rj * (wj - hvc);
      A.rowtmp[jrow] = A.vscr ? hvc : e * e;   // (reference level: T w_pa itself, for k_factor_ref_finish's e = Ri (w_u - T w_pa))
      A.rowtmp[A.n_rows + jrow] = log(rj);
      A.panels[S.pan0 + (size_t)jc * S.ld + P] = rj;
    }
  }
  __syncthreads();
  if (tid == 0 && s_fail) atomicMin(A.errflag, B.level * 16 + A.errcode);
  STAMP(11);
#ifdef FM_STAMPS
  STAMP_FLUSH_IF((g_stamp_level < 0 || g_stamp_level == B.level) && A.errcode == 3);   // (first half of a reference level: k_factor_ref_finish's stamps)
  if (tid == 64 * JTb && JTb < 4 && A.errcode == 3 && (g_stamp_level < 0 || g_stamp_level == B.level)) { for (int q_ = 12; q_ < 16; ++q_) atomicAdd(&g_stamps[q_], st_acc[q_]); }
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// Second half of phase A for REFERENCE levels of wide-block trees (round 3).  k_factor_bigmfma walks a block's chain as one
// latency chain per workgroup (28 sub-panels of ~22 us at config #4: 13-15 % of the FP64 pipe); k_factor_lchain, built for the
// non-reference levels, runs the same chain pass at 36 %.  A reference level therefore takes k_factor_lchain FIRST -- it treats
// the columns as conditionally independent and leaves, per column j, the panel row [ -r_j T_j | r_j ] with T = H_u and
// r_j = (K_jj - sum_k V_kj^2)^(-1/2), and V = Linv_pa K_pa,u in a scratch -- and this kernel finishes the block
// (spamtree_model.cpp:880-922):
//   R = K_uu - V'V   (a GEMM-style K loop: chunks of RF_KC rows of V through LDS, requested one chunk ahead, one barrier per chunk, the
//                     15 lower tile pairs dealt over the eight waves; both operands of a K-step come from the same chunk);
//   Ri = chol(R)^-1  (blocked, LDS);
//   panel <- [ -Ri T | Ri ] with T = -(row) / r_j, IN PLACE (a wave owns 16-column blocks of the chain: it holds the block's 75 x 16
//                     entries of T before it writes them; the next block's entries are requested before the current block's MFMAs);
//   e = Ri w_u + N w_pa;   logdet, loglik.
__global__ __launch_bounds__(BM_NT) void k_factor_ref_finish(FactorArgs A, CovPar cp) {
  extern __shared__ double lds[];
  __shared__ int s_fail;
  __shared__ double s_red[BM_NT / 64];
  __shared__ double s_cpt[LC_CPT];
  static_assert(RF_BUFD == (BM_NT / 64) * 3 * 128 && RF_BUFD >= RF_KC * RF_LDB && (RF_NBUF & (RF_NBUF - 1)) == 0, "chunk shape");
  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int maxM = A.maxM;
  double *wv = lds;                    // maxM each: the block's w, coordinates, T w_pa, 1 / r_j
  double *sx = wv + maxM;
  double *sy = sx + maxM;
  double *hv = sy + maxM;
  double *rinv = hv + maxM;
  double *work = rinv + maxM;   // the Schur product's staging buffers, then R | Ri
  double *Rl = work;
  double *Ril = Rl + (size_t)maxM * maxM;
  double *nri = work + rf_work_doubles(maxM);   // RF_NOPS x 64: -Ri as the MFMA A operands of the N phase (zero above the diagonal / beyond the block)
  int *smv = (int *)(nri + RF_NOPS * 64);
  lc_cov_table(s_cpt, cp, tid, BM_NT);
  const int qn = cp.q;
  const double s2 = cp.ai1[0], nphi = -cp.tmv[0];
  STAMP_DECL
  int st_level = 0;
  for (int li = blockIdx.x; li < A.nlist; li += gridDim.x) {
    const int b = A.list[li];
    const Blk B = A.blks[b];
    const int m = B.m, P = B.P, ld = B.ld;
    const int JT = (m + 15) >> 4;
    double *pu = A.panels + B.panel_off;
    const double *Vb = A.vscr + A.voff[li];
    st_level = B.level;
    __syncthreads();
    if (tid == 0) s_fail = 0;
    // V's first chunks are requested before anything else.  Chunk c = rows [RF_KC c, RF_KC (c + 1)) = one contiguous run of the
    // block's V; wave w moves pieces 3 w .. 3 w + 2 of 1 KB (one address, immediate offsets)
    const int nch = (P + RF_KC - 1) / RF_KC;
    auto dmaV = [&](int c) {
      const double *src = Vb + (size_t)c * (RF_KC * RF_LDB) + wid * 384 + 2 * lane;
      q_lds_void *dst = (q_lds_void *)(work + (size_t)(c & (RF_NBUF - 1)) * RF_BUFD + wid * 384);
      __builtin_amdgcn_global_load_lds((q_glb_void *)src, dst, 16, 0, 0);
      __builtin_amdgcn_global_load_lds((q_glb_void *)src, dst, 16, 1024, 0);
      __builtin_amdgcn_global_load_lds((q_glb_void *)src, dst, 16, 2048, 0);
    };
#pragma unroll
    for (int c = 0; c < RF_NBUF - 1; ++c)
      if (c < nch) dmaV(c);
    __syncthreads();
    for (int i = tid; i < m; i += BM_NT) {
      sx[i] = A.cx[B.row0 + i]; sy[i] = A.cy[B.row0 + i]; smv[i] = A.mv[B.row0 + i]; wv[i] = A.w_in[B.row0 + i]; hv[i] = A.hvrow[B.row0 + i];
      const double r = pu[(size_t)i * ld + P];
      if (!(r > 0.0) || !(r < 1e300)) s_fail = 1;    // k_factor_lchain met a non-positive conditional variance (it flagged it too)
      rinv[i] = 1.0 / r;
    }
    STAMP(0);
    // ---- R = K_uu - V'V (lower tile pairs (it, jt <= it); a wave owns pairs wid and wid + 8)
    {
      const int npair = JT * (JT + 1) / 2;
      int pit[2], pjt[2];
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        const int e = wid + 8 * h2;
        int it = 0;
        while ((it + 1) * (it + 2) / 2 <= e) ++it;
        pit[h2] = it; pjt[h2] = e - it * (it + 1) / 2;
      }
      const bool has0 = wid < npair, has1 = wid + 8 < npair;
      d4 c0 = (d4){0.0, 0.0, 0.0, 0.0}, c1 = (d4){0.0, 0.0, 0.0, 0.0};
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the block's own vectors above (the waits below count this wave's DMA only)
      for (int c = 0; c < nch; ++c) {
        const double *Vc = work + (size_t)(c & (RF_NBUF - 1)) * RF_BUFD;
        // this wave's pieces of chunk c have landed: the two younger chunks (three instructions each) may still be in flight
        if (c + RF_NBUF - 2 < nch) __builtin_amdgcn_s_waitcnt(0x0F76);        // vmcnt(6)
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // the chain's tail
        lds_barrier();   // ... and everybody else's; everybody is done with chunk c - 1, whose buffer chunk c + 3 takes
        if (c + RF_NBUF - 1 < nch) dmaV(c + RF_NBUF - 1);
        const int ns = min(RF_KC / 4, (P - RF_KC * c + 3) >> 2);
        const double *ap0 = Vc + l4 * RF_LDB + pit[0] * 16 + l15, *bp0 = Vc + l4 * RF_LDB + pjt[0] * 16 + l15;
        const double *ap1 = Vc + l4 * RF_LDB + pit[1] * 16 + l15, *bp1 = Vc + l4 * RF_LDB + pjt[1] * 16 + l15;
        if (ns == RF_KC / 4) {
          if (has0) {
#pragma unroll
            for (int s = 0; s < RF_KC / 4; ++s) c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ap0[4 * s * RF_LDB], bp0[4 * s * RF_LDB], c0, 0, 0, 0);
          }
          if (has1) {
#pragma unroll
            for (int s = 0; s < RF_KC / 4; ++s) c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ap1[4 * s * RF_LDB], bp1[4 * s * RF_LDB], c1, 0, 0, 0);
          }
        } else {   // the chain's last rows (whole K-steps only: V's rows up to the next multiple of four are zeros)
          for (int s = 0; s < ns; ++s) {
            if (has0) c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(ap0[4 * s * RF_LDB], bp0[4 * s * RF_LDB], c0, 0, 0, 0);
            if (has1) c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(ap1[4 * s * RF_LDB], bp1[4 * s * RF_LDB], c1, 0, 0, 0);
          }
        }
      }
      __syncthreads();   // the staging buffers are done with: R takes their place
      for (int idx = tid; idx < m * m; idx += BM_NT) Rl[idx] = 0.0;
      __syncthreads();
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        if (h2 == 0 ? has0 : has1) {
          const d4 c = h2 == 0 ? c0 : c1;
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = pit[h2] * 16 + l4 + 4 * r, j = pjt[h2] * 16 + l15;
            if (i < m && j <= i) {
              const double kij = qn == 1 ? lc_cov<false>(s_cpt, qn, s2, nphi, sx[i], sy[i], smv[i], sx[j], sy[j], smv[j])
                                         : lc_cov<true>(s_cpt, qn, s2, nphi, sx[i], sy[i], smv[i], sx[j], sy[j], smv[j]);
              Rl[i * m + j] = kij - c[r];
            }
          }
        }
      }
    }
    __syncthreads();
    STAMP(2);
    block_chol_invert_mfma(Rl, Ril, m, &s_fail);
    STAMP(3);
    // ---- panel <- -Ri T in place: wave w owns chain column blocks kt = w, w + 8, ...; it holds the block's T entries (all rows of
    // the unit: <= 80 = 20 K-steps) in registers before any of them is overwritten.  The next column block's entries are requested
    // before the current block's MFMAs and stores, and the wait for them one round later must not wait for those stores as well
    // (vmcnt counts loads and stores in issue order; measured: 37 % of the kernel when it was vmcnt(0)).  The compiler emits the
    // partial wait only if it can COUNT the younger operations on every path into the loop header, so the five-tile form (blocks of
    // 65-80 rows: the default multivariate tree) has no branch around a memory operation: loads take clamped addresses, every
    // lane stores exactly four values per row tile (lanes outside the block into the block's V scratch, which the Schur product
    // has consumed), and the last round requests its own block once more (the first round's wait is the loop's peeled prologue).
    {
      const int nkt = (P + 15) >> 4;
      double *dump = const_cast<double *>(Vb) + lane;   // (inside the block's own V for any P: it holds at least 4 x 80 doubles)
      // -Ri in operand order, once per block: an A operand of the loop below is ONE LDS read at an immediate offset (built per MFMA --
      // clamped address, two compares, select, negation -- it cost about ten VALU instructions on the pipe the FP64 MFMAs use)
      for (int idx = tid; idx < RF_NOPS * 64; idx += BM_NT) {
        const int o = idx >> 6, ln = idx & 63;
        const int it = o >= 40 ? 4 : (o >= 24 ? 3 : (o >= 12 ? 2 : (o >= 4 ? 1 : 0)));
        const int s2 = o - 2 * it * (it + 1), ia = it * 16 + (ln & 15), j = 4 * s2 + (ln >> 4);
        nri[idx] = (ia < m && j <= ia) ? -Ril[ia * m + j] : 0.0;
      }
      __syncthreads();
      auto apply = [&](auto fullc) {
        constexpr bool FULL = decltype(fullc)::value;
        double tn[20];   // B operands of the wave's NEXT column block: T[j = 4 s + l4][kb], raw panel entries
        auto loadTb = [&](int kt) {
          const int kbc = min(kt * 16 + l15, P - 1);
#pragma unroll
          for (int s2 = 0; s2 < 20; ++s2) tn[s2] = pu[(size_t)min(4 * s2 + l4, m - 1) * ld + kbc];
        };
        if (wid >= nkt) return;
        loadTb(wid);
        for (int kt = wid; kt < nkt; kt += BM_NT / 64) {
          const int kb = kt * 16 + l15;
          double tb[20];
#pragma unroll
          for (int s2 = 0; s2 < 20; ++s2) {
            const double rv = rinv[min(4 * s2 + l4, m - 1)];   // (read by every lane: no branch around it)
            tb[s2] = (4 * s2 + l4 < m && kb < P) ? -tn[s2] * rv : 0.0;
          }
          if constexpr (FULL) loadTb(min(kt + BM_NT / 64, nkt - 1));
          else if (kt + BM_NT / 64 < nkt) loadTb(kt + BM_NT / 64);   // another column block: no entry of it is written by this round
#pragma unroll
          for (int it = 0; it < 5; ++it) {
            if (FULL || it < JT) {
              d4 c = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
              for (int s2 = 0; s2 < 20; ++s2) {
                if (4 * s2 < it * 16 + 16)   // (compile time) Ri is lower triangular: row tile `it` needs columns < 16 (it + 1); rows of
                                             // T beyond the block are zero operands: no run-time guard, a tile's MFMAs are one basic block
                  c = __builtin_amdgcn_mfma_f64_16x16x4f64(nri[(2 * it * (it + 1) + s2) * 64 + lane], tb[s2], c, 0, 0, 0);
              }
#pragma unroll
              for (int r = 0; r < 4; ++r) {
                const int i = it * 16 + l4 + 4 * r;
                if constexpr (FULL) {
                  double *dst = (i < m && kb < P) ? pu + (size_t)i * ld + kb : dump;
                  *dst = c[r];
                } else {
                  if (i < m && kb < P) pu[(size_t)i * ld + kb] = c[r];
                }
              }
              __builtin_amdgcn_sched_barrier(0);   // a row tile at a time
            }
          }
        }
      };
      if (JT == 5) apply(std::true_type{}); else apply(std::false_type{});
    }
    STAMP(4);
    for (int idx = tid; idx < m * m; idx += BM_NT) {
      const int i = idx / m, j = idx - i * m;
      pu[(size_t)i * ld + P + j] = (j <= i) ? Ril[idx] : 0.0;
    }
    double wcore_part = 0.0, logdet_part = 0.0;
    for (int i = tid; i < m; i += BM_NT) {   // e = Ri (w_u - T w_pa)
      double acc = 0.0;
      for (int j = 0; j <= i; ++j) acc += Ril[i * m + j] * (wv[j] - hv[j]);
      wcore_part += acc * acc;
      logdet_part += log(Ril[i * m + i]);
    }
    const double wcore = block_sum(wcore_part, s_red);
    const double logdet = block_sum(logdet_part, s_red);
    __syncthreads();
    if (tid == 0) {
      A.logdet_c[b] = logdet;
      A.loglik_c[b] = (double)m * HL2PI - 0.5 * wcore;
      if (s_fail) atomicMin(A.errflag, B.level * 16 + 2);
    }
    STAMP(5);
  }
  STAMP_FLUSH_LEVEL(st_level);
}
template __global__ void k_factor_lchain<96>(LcArgs, CovPar);
template __global__ void k_factor_lchain<136>(LcArgs, CovPar);
#else   // host side: prototypes only
__global__ void k_lchain_scalars(const Blk *blks, const int *list, int nlist, const double *rowtmp, long long n_rows, double *logdet_c, double *loglik_c);
template <int NKX> __global__ void k_factor_lchain(LcArgs A, CovPar cp);
__global__ void k_factor_ref_finish(FactorArgs A, CovPar cp);
#endif
