// Ri = chol(R)^{-1} of ONE m x m block (m <= 27) by ONE wave, blocked 16 + (m - 16)
// (reference math: Ri = inv(trimatl(chol(symmatu(Kcc - H Kxc), "lower"))), /root/reference/src/spamtree_model.cpp:896-897).
//
// wave_chol_eliminate (spamtree_hip.hip) runs the 27-pivot elimination of [R | I] with every broadcast through v_readlane:
// two v_readlane_b32 + one v_fma_f64 per updated register, about 2 500 issue-bound instructions.  Here
//   * the elimination of a <= 16-row diagonal tile keeps row i in lane 16 g + i of ALL FOUR 16-lane rows g of the wave, so
//     the broadcast of A[j][k] to every row is the DPP control row_newbcast:j of a v_fmac_f64 -- ONE instruction instead of
//     three (64-bit DPP has no other control on gfx950, which is why the tile must fit a 16-lane row);
//   * the four lane rows repeat the elimination of A (free: same instructions) and each carries a QUARTER of the columns of
//     B = I -> L^{-1} (lane row g: columns 4 jj + g), which is exactly the layout of an FP64 MFMA operand
//     (A[i = lane & 15][k = lane >> 4], B[k = lane >> 4][j = lane & 15] per K-step jj);
//   * the coupling of the two diagonal tiles runs on the matrix cores:  L21 = R21 X11',  S22 = R22 - L21 L21',
//     Ri21 = -X22 (L21 X11), with L21 / S22 exchanged in place through the LDS image of R.
// About 600 DPP instructions + 15 MFMAs + three LDS round trips.
//
//   Rm: LDS, row stride CH_LD, lower triangle valid; destroyed.   Bm: LDS, receives L^{-1} (lower triangle only; what lies
//   above the diagonal is not written).  All 64 lanes of the wave must call; no barrier inside (single wave: LDS program order).
#pragma once
#if defined(ST_DEFS_FACTOR_QUAD) || defined(ST_DEFS_FACTOR_GENERIC)

// (fmac_bcast / dpp_tile_eliminate live in st_device.hpp: the blocked 16 x 16-tile factorisation of wide blocks uses them too)

template <int M2MAX = 11>
__device__ __forceinline__ void wave_chol_eliminate_blocked(double *Rm, double *Bm, int m, int *fail, int lane) {
  const int i = lane & 15, g = lane >> 4;
  const int m1 = min(m, 16), m2 = m - m1;
  // ---- X11 = chol(R11)^{-1}
  double a[16], b[4];
#pragma unroll
  for (int j = 0; j < 16; ++j) a[j] = (i < m1 && j <= i) ? Rm[i * CH_LD + j] : (j == i ? 1.0 : 0.0);
  bool bad = dpp_tile_eliminate<16>(a, b, m1, lane);
  if (i < m1) {
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
      if (4 * jj + g <= i) Bm[i * CH_LD + 4 * jj + g] = b[jj];
  }
  if (m2 > 0) {   // wave-uniform
    // ---- L21 = R21 X11'  (A operand: R21 rows from LDS; B[k][j] = X11[j][k] = this lane's b[k-step], zero above the diagonal)
    d4 c = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const double av = i < m2 ? Rm[(16 + i) * CH_LD + 4 * s + g] : 0.0;
      c = __builtin_amdgcn_mfma_f64_16x16x4f64(av, b[s], c, 0, 0, 0);
    }
    // in place: R21 <- L21 (C layout: rows g + 4 r, column i)
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (g + 4 * r < m2) Rm[(16 + g + 4 * r) * CH_LD + i] = c[r];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    double l21[4], x11t[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      l21[s] = i < m2 ? Rm[(16 + i) * CH_LD + 4 * s + g] : 0.0;                // L21[i][4 s + g]: A operand, and B operand of L21 L21'
      x11t[s] = (i <= 4 * s + g) ? Bm[(4 * s + g) * CH_LD + i] : 0.0;          // X11[4 s + g][i]: B operand of W = L21 X11 (rows < 16 <= m)
    }
    // ---- S22 = R22 - L21 L21'  (accumulator starts as R22; only its lower triangle is meaningful)
    d4 s22;
#pragma unroll
    for (int r = 0; r < 4; ++r) s22[r] = (g + 4 * r < m2 && i <= g + 4 * r) ? Rm[(16 + g + 4 * r) * CH_LD + 16 + i] : 0.0;
#pragma unroll
    for (int s = 0; s < 4; ++s) s22 = __builtin_amdgcn_mfma_f64_16x16x4f64(-l21[s], l21[s], s22, 0, 0, 0);
    // ---- W = L21 X11 (independent of the second elimination: issued before it, consumed after it)
    d4 w = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < 4; ++s) w = __builtin_amdgcn_mfma_f64_16x16x4f64(l21[s], x11t[s], w, 0, 0, 0);
    // S22 back to LDS (in place), then one row per lane again
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (g + 4 * r < m2 && i <= g + 4 * r) Rm[(16 + g + 4 * r) * CH_LD + 16 + i] = s22[r];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    double a2[M2MAX], b2[(M2MAX + 3) / 4];
#pragma unroll
    for (int j = 0; j < M2MAX; ++j) a2[j] = (i < m2 && j <= i) ? Rm[(16 + i) * CH_LD + 16 + j] : (j == i ? 1.0 : 0.0);
    bad = dpp_tile_eliminate<M2MAX>(a2, b2, m2, lane) || bad;
    if (i < m2) {
#pragma unroll
      for (int jj = 0; jj < (M2MAX + 3) / 4; ++jj)
        if (4 * jj + g <= i) Bm[(16 + i) * CH_LD + 16 + 4 * jj + g] = b2[jj];
    }
    // ---- Ri21 = -X22 W  (A[i][k] = X22[i][4 s + g] = b2[s], zero beyond the diagonal and for rows >= m2; B = W in its
    // accumulator layout, which is the B-operand layout of K-step s)
    d4 r21 = (d4){0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < (M2MAX + 3) / 4; ++s) {
      const double xa = (i < m2 && 4 * s + g <= i) ? -b2[s] : 0.0;
      r21 = __builtin_amdgcn_mfma_f64_16x16x4f64(xa, w[s], r21, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (g + 4 * r < m2) Bm[(16 + g + 4 * r) * CH_LD + i] = r21[r];
  }
  if (bad && lane == 0) *fail = 1;
}
#endif   // ST_DEFS_FACTOR_QUAD
