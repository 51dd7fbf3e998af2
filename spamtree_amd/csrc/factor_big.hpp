// Included by spamtree_hip.hip after factor_quad.hpp (needs FactorArgs, Blk, CovPar, cov_entry, d4, dma typedefs, chol helpers).
#pragma once

// Phase A for blocks wider than 32 columns or chains longer than 256 rows (default multivariate trees: 75-row blocks,
// chains up to 525 rows -- config #4).  Same per-block workgroup, scratch arena and epilogue as k_factor<true, MODE_FACTOR>;
// the pass over the ancestor chain runs on the FP64 matrix cores:
//   * 16-row sub-panels of the ancestors' panels staged by LDS-DMA;
//   * V_sub = Linv_sub K: A from LDS, B straight from K in the workgroup's scratch slice (L2-resident), one column tile per
//     wave (the fifth tile goes to wave 0);
//   * T[column][chain] += V_sub' Linv_sub with the accumulators in REGISTERS: wave w owns the chain tiles kt = w, w+8, ...
//     for all column tiles; BM_JT x BM_KTW tiles per wave cover BM_KTP chain tiles per pass, longer chains take two passes -- the second re-stages the panels of the ancestors that reach past the first pass's rows and reads V back from the
//     scratch slice (V replaced the rows of K it was computed from, as in the generic kernel).
// M <= 80 columns, P <= 544 rows (host check); anything else stays on k_factor<true, MODE_FACTOR>.
#define BM_MAXP 544
#define BM_NT 512   // 8 waves: two per SIMD, so that one wave's LDS-DMA / L2 / LDS waits overlap the other's MFMAs
#define BM_KS 4     // the V phase splits a sub-panel's K range (its chain columns) four ways: (column tile, K slice) items are
                    // dealt over all 8 waves -- a block's 3-5 column tiles alone keep only 3-5 of them busy for the whole
                    // dependent MFMA chain (stamps: 56 % / 32 % of the leaf / reference level at config #4)
// BM_JT column tiles x BM_KTW chain tiles per wave in registers; BM_KTP <= 8 BM_KTW chain tiles per pass: <5, 3, 24> for
// blocks up to 80 columns (two passes beyond 384 rows), <4, 5, 34> / <3, 5, 34> for blocks up to 64 / 48 columns (the
// leaves of the default multivariate tree: one pass).
#ifdef ST_DEFS_FACTOR_WIDE
template <int BM_JT, int BM_KTW, int BM_KTP>
__global__ __launch_bounds__(BM_NT, 2) void k_factor_bigmfma(FactorArgs A, CovPar cp) {
  extern __shared__ double lds[];
  __shared__ int s_anc[MAXJ], s_am[MAXJ], s_ao[MAXJ + 1];
  __shared__ long long s_arow[MAXJ], s_apan[MAXJ];
  __shared__ int s_fail;
  __shared__ double s_red[BM_NT / 64];
  constexpr bool BIG = true;
  constexpr int MODE = MODE_FACTOR;

  const int tid = threadIdx.x, lane = tid & 63, l15 = lane & 15, l4 = lane >> 4;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int maxP = A.maxP, maxM = A.maxM, maxMa = A.maxMa;
  const int ldS = A.SR;                  // staged row stride (the host passes it in SR): >= maxP + 24, 2 * odd
  // LDS carve
  double *sx = lds;
  double *sy = sx + (maxP + maxM);
  double *wv = sy + (maxP + maxM);
  double *hv = wv + (maxP + maxM);     // maxM
  double *ev = hv + maxM;              // maxM
  double *rd = ev + maxM;              // maxM
  double *stage = rd + maxM;           // 16 * ldS (+ zrow + VpL >= 2 maxM^2 doubles: the epilogue factorises there)
  double *zrow = stage + (size_t)16 * ldS;   // ldS zeros
  double *VpL = zrow + ldS;            // 16 x 80: the current sub-panel's V
  double *red = VpL + 16 * 80;         // BM_KS x BM_JT x 256: the V phase's partial tiles (K range split over the waves)
  const size_t work = max((size_t)17 * ldS + 16 * 80 + BM_KS * BM_JT * 256, (size_t)2 * maxM * maxM + 64);   // the epilogue's R, Ri overlay stage .. red
  int *smv = (int *)(stage + work);
  double *KV, *Tt, *Vp, *R, *Ri;
  {
    double *g = A.scratch + (size_t)blockIdx.x * A.scratch_stride;
    KV = g; Tt = KV + (size_t)maxP * maxM; Vp = Tt + (size_t)maxP * maxM; R = Vp + (size_t)maxMa * maxM; Ri = R + (size_t)maxM * maxM;
  }
  for (int i = tid; i < (int)work; i += BM_NT) stage[i] = 0.0;   // stage, zero row, VpL: never NaN garbage
  STAMP_DECL
  int st_level = 0;
  for (int li = blockIdx.x; li < A.nlist; li += gridDim.x) {
    const int b = A.list[li];
    const Blk B = A.blks[b];
    const int m = B.m, P = B.P, J = B.nanc;
    st_level = B.level;
    __syncthreads();
    for (int i = tid; i < ldS + 16 * 80; i += BM_NT) zrow[i] = 0.0;   // the previous block's epilogue wrote over the zero row and the V tile
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
      for (int i = tid; i < s_am[t]; i += BM_NT) {
        sx[oa + i] = A.cx[r0 + i]; sy[oa + i] = A.cy[r0 + i]; smv[oa + i] = A.mv[r0 + i]; wv[oa + i] = A.w_in[r0 + i];
      }
    }
    for (int i = tid; i < m; i += BM_NT) {
      sx[P + i] = A.cx[B.row0 + i]; sy[P + i] = A.cy[B.row0 + i]; smv[P + i] = A.mv[B.row0 + i]; wv[P + i] = A.w_in[B.row0 + i];
    }
    __syncthreads();
    STAMP(0);
    // K_{pa,u}  (covariance_functions.cpp:95-111 / :213-286), T = 0
    for (int idx = tid; idx < P * m; idx += BM_NT) {
      const int k = idx / m, j = idx - k * m;
      KV[idx] = cov_entry(cp, sx[k], sy[k], smv[k], sx[P + j], sy[P + j], smv[P + j]);
    }
    __syncthreads();
    STAMP(1);
    // ---- the ancestor chain on the matrix cores
    const int JT = (m + 15) >> 4;
    const int npass = P > 16 * BM_KTP ? 2 : 1;
    for (int pass = 0; pass < npass; ++pass) {
      d4 tacc[BM_JT][BM_KTW];
#pragma unroll
      for (int a = 0; a < BM_JT; ++a)
#pragma unroll
        for (int c = 0; c < BM_KTW; ++c) tacc[a][c] = (d4){0.0, 0.0, 0.0, 0.0};
      const int kt0 = pass * BM_KTP;
      for (int t = J - 1; t >= 0; --t) {
        const int ma = s_am[t], oa = s_ao[t], Kb = oa + ma;
        if (Kb <= 16 * kt0) continue;   // second pass: this ancestor does not reach the upper chain tiles (uniform)
        const double *pa = A.panels + s_apan[t];
        for (int r0 = 0; r0 < ma; r0 += 16) {
          const int sr = min(16, ma - r0);
          __syncthreads();   // everyone is done with the previous sub-panel's stage / VpL
          // stage rows r0 .. r0+sr-1 (wave w: rows w, w+4, ...) by LDS-DMA, 128 doubles per piece
#pragma unroll
          for (int rr = 0; rr < 16 / (BM_NT / 64); ++rr) {
            const int row = wid + (BM_NT / 64) * rr;
            if (row < sr) {
              const double *src = pa + (size_t)(r0 + row) * Kb;
              for (int c = 0; 128 * c < Kb; ++c)
                if (128 * c + 2 * lane < Kb)
                  __builtin_amdgcn_global_load_lds((q_glb_void *)(src + 128 * c + 2 * lane), (q_lds_void *)(stage + (size_t)row * ldS + 128 * c), 16, 0, 0);
            }
          }
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
          for (int rr = 0; rr < 16 / (BM_NT / 64); ++rr) {
            const int row = wid + (BM_NT / 64) * rr;
            if (lane < 20) stage[(size_t)row * ldS + Kb + lane] = 0.0;   // K-step / tile overshoot reads zeros
          }
          if (pass == 1) {   // V of this sub-panel comes back from the scratch slice (it replaced K's rows oa ..)
            for (int idx = tid; idx < 16 * 80; idx += BM_NT) {
              const int i = idx / 80, j = idx - i * 80;
              VpL[idx] = (i < sr && j < m) ? KV[(size_t)(oa + r0 + i) * m + j] : 0.0;
            }
          }
          __syncthreads();
          STAMP(2);
          // the chain's factor is lower triangular in chain order: rows oa + r0 .. oa + r0 + sr - 1 hold nothing but (explicit)
          // zeros beyond column oa + r0 + sr - 1, although their stored length runs to the end of the ancestor's block (Kb)
          const int Kbe = min(Kb, oa + r0 + sr);
          if (pass == 0) {
            // V_sub = Linv_sub[:, 0:Kbe] K[0:Kbe, :]: items (column tile jt, K slice ks), partial tiles summed through LDS
            const int ns = (Kbe + 3) >> 2;
            const int per = (ns + BM_KS - 1) / BM_KS;   // K-steps per slice
            const double *ap = ((l15 < sr) ? stage + (size_t)l15 * ldS : zrow) + l4;
            for (int it = wid; it < JT * BM_KS; it += BM_NT / 64) {
              const int jt = it / BM_KS, ks = it - jt * BM_KS;
              const int s0 = ks * per, s1 = min(ns, s0 + per);
              const int j = jt * 16 + l15;
              const bool jok = j < m;
              const double *bp = KV + (size_t)l4 * m + min(j, m - 1);
              d4 p = (d4){0.0, 0.0, 0.0, 0.0};
              int st = s0;
              for (; st + 8 <= s1; st += 8) {   // eight B operands (L2) in flight per lane
                double a4[8], b4[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                  const int k = 4 * (st + q) + l4;
                  b4[q] = (jok && k < Kb) ? bp[(size_t)4 * (st + q) * m] : 0.0;
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) a4[q] = ap[4 * (st + q)];
#pragma unroll
                for (int q = 0; q < 8; ++q) p = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[q], b4[q], p, 0, 0, 0);
              }
              for (; st < s1; ++st) {
                const int k = 4 * st + l4;
                const double b1 = (jok && k < Kb) ? bp[(size_t)4 * st * m] : 0.0;
                p = __builtin_amdgcn_mfma_f64_16x16x4f64(ap[4 * st], b1, p, 0, 0, 0);
              }
              double *rp = red + (size_t)(ks * BM_JT + jt) * 256 + lane;
              rp[0] = p[0]; rp[64] = p[1]; rp[128] = p[2]; rp[192] = p[3];
            }
            __syncthreads();
            // the slices' sum, in slice order: element (i = l4 + 4 r, j = jt * 16 + l15) of the sub-panel's V
            for (int e = tid; e < JT * 256; e += BM_NT) {
              const int jt = e >> 8, q = e & 255, r = q >> 6, ln = q & 63;
              double v = red[(size_t)jt * 256 + q];
#pragma unroll
              for (int ks = 1; ks < BM_KS; ++ks) v += red[(size_t)(ks * BM_JT + jt) * 256 + q];
              const int i = (ln >> 4) + 4 * r, j = jt * 16 + (ln & 15);
              VpL[i * 80 + j] = v;
              if (i < sr && j < m) Vp[(size_t)(r0 + i) * m + j] = v;
            }
            __syncthreads();
            STAMP(3);
          }
          // T[column tile jt][chain tile kt] += V_sub' Linv_sub for this wave's chain tiles kt = kt0 + wid + 8 c.  The V
          // operands (all column tiles, all K-steps) are read once per sub-panel and reused for every chain tile.
          {
            const int nst = (sr + 3) >> 2;
            double av[BM_JT][4];
#pragma unroll
            for (int a = 0; a < BM_JT; ++a)
#pragma unroll
              for (int r = 0; r < 4; ++r) av[a][r] = (a < JT) ? VpL[(4 * r + l4) * 80 + a * 16 + l15] : 0.0;
#pragma unroll
            for (int c = 0; c < BM_KTW; ++c) {
              const int kt = kt0 + wid + (BM_NT / 64) * c;
              if (wid + (BM_NT / 64) * c < BM_KTP && kt * 16 < Kbe) {
                const double *b0 = stage + (size_t)l4 * ldS + kt * 16 + l15;
                double bv[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) bv[r] = b0[(size_t)4 * r * ldS];
#pragma unroll
                for (int a = 0; a < BM_JT; ++a) {
                  if (a < JT) {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                      if (r < nst) tacc[a][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[a][r], bv[r], tacc[a][c], 0, 0, 0);
                  }
                }
              }
            }
          }
        }
        STAMP(4);
        if (pass == 0) {   // the ancestor's V rows replace the K rows they were computed from
          __syncthreads();
          for (int idx = tid; idx < ma * m; idx += BM_NT) KV[(size_t)oa * m + idx] = Vp[idx];
          STAMP(5);
        }
      }
      // this pass's T tiles -> the scratch slice (the epilogue reads T from there)
#pragma unroll
      for (int c = 0; c < BM_KTW; ++c) {
        const int kt = kt0 + wid + (BM_NT / 64) * c;
        if (wid + (BM_NT / 64) * c < BM_KTP) {
#pragma unroll
          for (int a = 0; a < BM_JT; ++a) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int j = a * 16 + l4 + 4 * r, k = kt * 16 + l15;
              if (j < m && k < P) Tt[(size_t)j * P + k] = tacc[a][c][r];
            }
          }
        }
      }
    }
    __syncthreads();
    // hv = H w_pa  (wave per row)
    {
      const int lane = tid & 63, wid = tid >> 6;
      for (int j = wid; j < m; j += BM_NT / 64) {
        double acc = 0.0;
        for (int k = lane; k < P; k += 64) acc += Tt[j * P + k] * wv[k];
        acc = wave_sum(acc);
        if (lane == 0) hv[j] = acc;
      }
    }
    __syncthreads();
    STAMP(6);

    double *pu = A.panels + B.panel_off;
    const int ld = B.ld;
    double wcore_part = 0.0, logdet_part = 0.0;
    if (B.isref) {
      // R = K_uu - V'V (lower) on the matrix cores: both operands are columns of V in the scratch slice; the m x m
      // factorisation and inversion run in LDS (the stage area is free now), not in the scratch slice
      double *Rl = stage, *Ril = stage + (size_t)m * m;
      for (int idx = tid; idx < m * m; idx += BM_NT) Rl[idx] = 0.0;
      __syncthreads();
      {
        const int ns = (P + 3) >> 2;
        for (int e = wid; e < JT * (JT + 1) / 2; e += BM_NT / 64) {
          int it = 0;
          while ((it + 1) * (it + 2) / 2 <= e) ++it;
          const int jt = e - it * (it + 1) / 2;
          const int ci = it * 16 + l15, cj = jt * 16 + l15;
          const double *ap = KV + (size_t)l4 * m + min(ci, m - 1), *bp = KV + (size_t)l4 * m + min(cj, m - 1);
          d4 c = (d4){0.0, 0.0, 0.0, 0.0};
          int st = 0;
          for (; st + 4 <= ns; st += 4) {
            double a4[4], b4[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const bool kok = 4 * (st + q) + l4 < P;
              a4[q] = (kok && ci < m) ? ap[(size_t)4 * (st + q) * m] : 0.0;
              b4[q] = (kok && cj < m) ? bp[(size_t)4 * (st + q) * m] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) c = __builtin_amdgcn_mfma_f64_16x16x4f64(a4[q], b4[q], c, 0, 0, 0);
          }
          for (; st < ns; ++st) {
            const bool kok = 4 * st + l4 < P;
            const double a1 = (kok && ci < m) ? ap[(size_t)4 * st * m] : 0.0, b1 = (kok && cj < m) ? bp[(size_t)4 * st * m] : 0.0;
            c = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, c, 0, 0, 0);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = it * 16 + l4 + 4 * r, j = jt * 16 + l15;
            if (i < m && j <= i) Rl[i * m + j] = cov_entry(cp, sx[P + i], sy[P + i], smv[P + i], sx[P + j], sy[P + j], smv[P + j]) - c[r];
          }
        }
      }
      __syncthreads();
      STAMP(7);
      block_chol_invert_mfma(Rl, Ril, m, &s_fail);
      STAMP(8);
      // panel_u = [ -Ri*T | Ri ]: tiles (row tile it, chain tile kt), A = -Ri from LDS, B = T from the scratch slice
      {
        const int nkt = (P + 15) >> 4;
        for (int e = wid; e < JT * nkt; e += BM_NT / 64) {
          const int it = e % JT, kt = e / JT;
          const int ia = it * 16 + l15, kb = kt * 16 + l15;
          const int njs = (min(m, it * 16 + 16) + 3) >> 2;
          d4 c = (d4){0.0, 0.0, 0.0, 0.0};
          for (int st = 0; st < njs; ++st) {
            const int j = 4 * st + l4;
            const double a1 = (ia < m && j <= ia) ? -Ril[ia * m + j] : 0.0;
            const double b1 = (j < m && kb < P) ? Tt[(size_t)j * P + kb] : 0.0;
            c = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, c, 0, 0, 0);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int i = it * 16 + l4 + 4 * r;
            if (i < m && kb < P) pu[(size_t)i * ld + kb] = c[r];
          }
        }
      }
      for (int idx = tid; idx < m * m; idx += BM_NT) {
        const int i = idx / m, j = idx - i * m;
        pu[(size_t)i * ld + P + j] = (j <= i) ? Ril[idx] : 0.0;
      }
      // e = Ri (w_u - H w_pa)
      for (int i = tid; i < m; i += BM_NT) {
        double acc = 0.0;
        for (int j = 0; j <= i; ++j) acc += Ril[i * m + j] * (wv[P + j] - hv[j]);
        wcore_part += acc * acc;
        logdet_part += log(Ril[i * m + i]);
      }
    } else {
      // non-reference level: rows conditionally independent (spamtree_model.cpp:923-963).  sum_k V[k][i]^2 in four
      // interleaved partial sums per column (stage area), added in a fixed order
      double *part4 = stage;   // (BM_NT / 64) x 128
      for (int i = tid & 63; i < m; i += 64) {
        const int q = tid >> 6;
        double acc = 0.0;
        for (int k = q; k < P; k += BM_NT / 64) { const double v = KV[(size_t)k * m + i]; acc += v * v; }
        part4[q * 128 + i] = acc;
      }
      __syncthreads();
      for (int i = tid; i < m; i += BM_NT) {
        double acc = cov_entry(cp, sx[P + i], sy[P + i], smv[P + i], sx[P + i], sy[P + i], smv[P + i]);
        for (int q = 0; q < BM_NT / 64; ++q) acc -= part4[q * 128 + i];
        if (!(acc > 0.0)) s_fail = 1;
        const double r = 1.0 / sqrt(acc);
        rd[i] = r;
        pu[(size_t)i * ld + P] = r;
        const double e = r * (wv[P + i] - hv[i]);
        wcore_part += e * e;
        logdet_part += log(r);
      }
      __syncthreads();
      for (int idx = tid; idx < m * P; idx += BM_NT) {
        const int i = idx / P, k = idx - i * P;
        pu[(size_t)i * ld + k] = -rd[i] * Tt[idx];
      }
    }
    const double wcore = block_sum(wcore_part, s_red);
    const double logdet = block_sum(logdet_part, s_red);
    __syncthreads();
    STAMP(9);
    if (tid == 0) {
      A.logdet_c[b] = logdet;
      A.loglik_c[b] = (double)m * HL2PI - 0.5 * wcore;
      if (s_fail) atomicMin(A.errflag, B.level * 16 + (J == 0 ? 1 : (B.isref ? 2 : 3)));
    }
  }
  STAMP_FLUSH_LEVEL(st_level);
}
template __global__ void k_factor_bigmfma<3, 5, 34>(FactorArgs, CovPar);
template __global__ void k_factor_bigmfma<4, 5, 34>(FactorArgs, CovPar);
template __global__ void k_factor_bigmfma<5, 3, 24>(FactorArgs, CovPar);
#else   // host side: prototypes only
template <int BM_JT, int BM_KTW, int BM_KTP> __global__ void k_factor_bigmfma(FactorArgs A, CovPar cp);
#endif
