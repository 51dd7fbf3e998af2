// Included by spamtree_hip.hip after factor_wide.hpp (needs FactorArgs, Blk, CovPar, cov_entry, d4, dma typedefs, lds_barrier).
#pragma once

// Phase A of NON-REFERENCE blocks behind long ancestor chains (the leaves of the default multivariate tree, config #4:
// 36-column blocks, chains of 525 rows; spamtree_model.cpp:923-963), third generation.  k_factor_bigmfma / k_factor_wide
// keep K_{pa,u} and V in a global scratch slice and feed the matrix cores with 8-byte loads from L2 (56 % of the leaf level),
// with three workgroup barriers per 16 chain rows.  Here a workgroup is ONE block and a wave owns 16 of its columns over the
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

__host__ __device__ constexpr int lc_lds_stride(int nkx) {   // phase-1 row stride: >= 4 nkx + 24, 2 x odd (conflict-free A reads)
  int s = 4 * nkx + 24;
  while ((s & 1) || ((s >> 1) & 1) == 0) ++s;
  return s;
}
__host__ __device__ constexpr size_t lc_dyn_doubles(int nkx) {   // dynamic LDS: two buffers, max over the phases
  const size_t a = (size_t)2 * 16 * lc_lds_stride(nkx), b = (size_t)2 * (4 * nkx + 16) * 16;
  return a > b ? a : b;
}

template <int NKX>
__global__ __launch_bounds__(LC_NT) void k_factor_lchain(FactorArgs A, CovPar cp) {
  constexpr int PMAX = 4 * NKX, NTMAX = (PMAX + 15) / 16, ldS = lc_lds_stride(NKX);
  constexpr int B1 = 16 * ldS;             // doubles per phase-1 buffer (one 16-row tile, full length)
  constexpr int B2 = (PMAX + 16) * 16;     // doubles per phase-2 buffer (one 16-column block, row c at c * 16)
  constexpr int KH = (B1 - (5 * PMAX) / 2 - 8) / 256;   // covariance scratch: K-steps per pass ([KH][64] doubles per wave)
  constexpr int NPASS = (NKX + KH - 1) / KH;
  static_assert(KH >= 8, "covariance scratch too small");
  extern __shared__ double lds[];
  __shared__ int s_am[MAXJ], s_ao[MAXJ + 1];
  __shared__ long long s_arow[MAXJ], s_apan[MAXJ];
  __shared__ double s_wpa[PMAX];
  __shared__ long long s_rsrc[PMAX];   // chain row c: where it starts in the panel arena ...
  __shared__ int s_rlen[PMAX];         // ... and its length (entries up to the end of its own ancestor's rows)
  __shared__ double s_e2[64], s_lg[64];
  __shared__ int s_fail;

  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  int li = blockIdx.x;
  {
    const int per = A.nlist >> 3;   // one contiguous run of blocks per XCD (private L2): siblings share their chain's panels
    if (li < per * 8) li = (li & 7) * per + (li >> 3);
  }
  const int b = A.list[li];
  const Blk B = A.blks[b];
  const int m = B.m, P = B.P, J = B.nanc;
  const int NTL = (P + 15) >> 4;

  for (int i = tid; i < (int)lc_dyn_doubles(NKX); i += LC_NT) lds[i] = 0.0;   // never NaN garbage under a zero multiplier
  if (tid < J) {
    const int a = A.anc_idx[B.anc_ptr + tid];
    s_am[tid] = A.blks[a].m; s_arow[tid] = A.blks[a].row0; s_apan[tid] = A.blks[a].chain_off;
  }
  if (tid == 0) s_fail = 0;
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

  // phase-1 staging: tile r = chain rows [16 r, 16 r + 16); wave w moves rows w, w + 4, w + 8, w + 12 (128 doubles per piece)
  auto issue1 = [&](int r, double *buf) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int row = wid + 4 * j, c = 16 * r + row;
      if (c < P) {
        const int len = __builtin_amdgcn_readfirstlane(s_rlen[c]);
        const double *src = A.panels + s_rsrc[c];
        double *dst = buf + (size_t)row * ldS;
        for (int pc = 0; 128 * pc < len; ++pc)
          if (128 * pc + 2 * lane < len)
            __builtin_amdgcn_global_load_lds((q_glb_void *)(src + 128 * pc + 2 * lane), (q_lds_void *)(dst + 128 * pc), 16, 0, 0);
      }
    }
  };
  issue1(NTL - 1, lds);   // lands under the covariance pass

  // ---- K_{pa,u}: kx[st] = K[4 st + l4][column 16 wid + l15]  (covariance_functions.cpp:95-111 / :213-286), rolled loop
  // through lane-private LDS slots, picked up with static register indices
  const int jc = 16 * wid + l15;
  const bool cok = jc < m;
  const bool wact = 16 * wid < m;   // this wave owns at least one column
  const long long jrow = B.row0 + min(jc, m - 1);
  const double mx = A.cx[jrow], my = A.cy[jrow], wj = A.w_in[jrow];
  const int mvj = A.mv[jrow];
  double kx[NKX];
  {
    double *kb = sy + PMAX + (PMAX + 1) / 2 + 2 + (size_t)wid * (KH * 64) + lane;
#pragma unroll
    for (int hp = 0; hp < NPASS; ++hp) {
      const int st0 = hp * KH;
      if (4 * st0 < P && wact) {
#pragma unroll 1
        for (int i = 0; i < KH; ++i) {
          const int k = 4 * (st0 + i) + l4;
          double v = 0.0;
          if (cok && k < P) v = cov_entry(cp, sx[k], sy[k], smv[k], mx, my, mvj);
          kb[i * 64] = v;
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

#define LCMFMA(a_, b_, c_) c_ = __builtin_amdgcn_mfma_f64_16x16x4f64(a_, b_, c_, 0, 0, 0)
  // ---- phase 1: V_r = Linv[r, 0 .. r] K, last tile first; V_r replaces kx[4 r .. 4 r + 3]
  double dacc = 0.0;   // sum_k V[k][column l15]^2 over this lane's rows
  {
    int cur = 0;
#pragma unroll
    for (int r = NTMAX - 1; r >= 0; --r) {
      if (r < NTL) {   // workgroup-uniform
        double *buf = lds + (size_t)cur * B1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of tile r have landed
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int row = wid + 4 * j, c = 16 * r + row;
          if (c < P) {
            const int len = __builtin_amdgcn_readfirstlane(s_rlen[c]);
            // the tile reads columns < 16 (r + 1) only; zero from the row's own end (also wipes the DMA's odd-length overshoot)
            if (len + lane < 16 * (r + 1)) buf[(size_t)row * ldS + len + lane] = 0.0;
          } else {
            for (int k = lane; k < 16 * (r + 1); k += 64) buf[(size_t)row * ldS + k] = 0.0;   // rows beyond the chain (last tile)
          }
        }
        lds_barrier();
        if (r > 0) issue1(r - 1, lds + (size_t)(cur ^ 1) * B1);
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
        cur ^= 1;
      }
    }
  }
  lds_barrier();   // phase 2 reuses the buffers

  // ---- r_j = 1 / sqrt(K_jj - sum_k V_kj^2)  (spamtree_model.cpp:944-951)
  double rj = 0.0;
  {
    double dsum = dacc;
    dsum += __shfl_xor(dsum, 16, 64);
    dsum += __shfl_xor(dsum, 32, 64);
    if (cok) {
      const double d = cov_entry(cp, mx, my, mvj, mx, my, mvj) - dsum;
      if (!(d > 0.0)) s_fail = 1;
      rj = 1.0 / sqrt(d);
    }
  }
  double rq[4];   // r of column 16 wid + 4 q + l4 (lane (0, column) holds that column's r)
#pragma unroll
  for (int q = 0; q < 4; ++q) rq[q] = __shfl(rj, l4 + 4 * q, 64);

  // ---- phase 2: T[:, kt] = sum_{r >= kt} V_r' Linv[r, kt]; column block kt of Linv = chain rows [16 kt, P) x 16 columns,
  // row c at offset c * 16 of the buffer (static operand offsets); one LDS-DMA instruction moves 8 rows (8 lanes x 16 B each)
  auto issue2 = [&](int kt, double *buf) {
    const int g0 = 2 * kt, g1 = (P + 7) >> 3;   // groups of 8 rows; group g goes to wave g & 3
    for (int g = g0 + ((wid - g0) & 3); g < g1; g += 4) {
      const int c = 8 * g + (lane >> 3);
      if (c < P) {
        const double *src = A.panels + s_rsrc[c] + 16 * kt + 2 * (lane & 7);
        __builtin_amdgcn_global_load_lds((q_glb_void *)src, (q_lds_void *)(buf + (size_t)g * 128), 16, 0, 0);
      }
    }
  };
  double h0 = 0.0, h1 = 0.0, h2 = 0.0, h3 = 0.0;   // hv = T w_pa for columns 4 q + l4, this lane's chain columns
  {
    double *pu = A.panels + B.panel_off;
    const int ld = B.ld;
    issue2(0, lds);
    int cur = 0;
    for (int kt = 0; kt < NTL; ++kt) {
      double *buf = lds + (size_t)cur * B2;
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      // the diagonal tile's entries above the diagonal are structural zeros, but rows that END inside this column block were
      // fetched past their end: wipe (the owner of the 8-row group does it, after its own data has landed)
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int g = 2 * kt + hh;
        if (wid == (g & 3)) {
          const int i = 8 * hh + (lane >> 3), j0 = 2 * (lane & 7);
          double *e = buf + (size_t)(16 * kt + i) * 16 + j0;
          if (j0 > i) e[0] = 0.0;
          if (j0 + 1 > i) e[1] = 0.0;
        }
      }
      lds_barrier();
      if (kt + 1 < NTL) issue2(kt + 1, lds + (size_t)(cur ^ 1) * B2);
      if (wact) {
        d4 t = (d4){0.0, 0.0, 0.0, 0.0};
        const double *bp = buf + l4 * 16 + l15;
#pragma unroll
        for (int r = 0; r < NTMAX; ++r) {
          if (r >= kt && r < NTL) {   // workgroup-uniform
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (4 * r + q < NKX) LCMFMA(kx[4 * r + q], bp[(16 * r + 4 * q) * 16], t);
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
      cur ^= 1;
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
      s_e2[jc] = e * e;
      s_lg[jc] = log(rj);
      A.panels[B.panel_off + (size_t)jc * B.ld + P] = rj;
    }
  }
  __syncthreads();
  if (tid == 0) {
    double wc = 0.0, ldt = 0.0;
    for (int j = 0; j < m; ++j) { wc += s_e2[j]; ldt += s_lg[j]; }
    A.logdet_c[b] = ldt;
    A.loglik_c[b] = (double)m * HL2PI - 0.5 * wc;
    if (s_fail) atomicMin(A.errflag, B.level * 16 + 3);
  }
}
