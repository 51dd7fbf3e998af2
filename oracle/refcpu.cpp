// ORACLE / CPU BASELINE -- test infrastructure only, never linked into the product.
//
// refcpu: C++17/OpenMP restatement of the reference's per-sweep algorithm AS WRITTEN, with the same dense
// materialisations (Kxx_invchol and Kxx_inv of size (P+m)^2, the P x P message matrix per block, message cubes,
// two cache copies) and the same `#pragma omp parallel for` over the blocks of one level.  It is (a) the timed
// CPU baseline of bench.py ("cpu_baseline.kind" = "port") and (b) a second, fast oracle for mid-size parity tests.
// STATUS: parity unpinned (the reference has no golden vectors; see oracle/spamtree_oracle.py header); this
// file is itself checked against the NumPy restatement in tests/test_refcpu.py.
//
// Restated from /root/reference/src (nothing is copied; Armadillo/LAPACK calls are replaced by the small
// column-major kernels below because no BLAS/LAPACK exists in the image):
//   covariance_functions.cpp:34-75, 95-135, 213-286        -> CovPar, cov_entry, covmat
//   tree_utils.cpp:194-208                                 -> invchol_block_inplace_direct
//   spamtree_model.cpp:194-301, 303-313, 315-353, 355-420, 422-503 -> RefModel ctor
//   spamtree_model.cpp:834-998 (A)  :1011-1226 (B)  :781-826 (C)  :1364-1417 (S1, S2 statistics)
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

const double HL2PI = -0.5 * std::log(2.0 * M_PI);

struct Mat {  // column-major
  int r = 0, c = 0;
  std::vector<double> a;
  Mat() {}
  Mat(int r_, int c_) : r(r_), c(c_), a((size_t)r_ * c_, 0.0) {}
  double &operator()(int i, int j) { return a[(size_t)j * r + i]; }
  double operator()(int i, int j) const { return a[(size_t)j * r + i]; }
};

// C = A' * B
Mat gemm_tn(const Mat &A, const Mat &B) {
  Mat C(A.c, B.c);
  for (int j = 0; j < B.c; ++j)
    for (int i = 0; i < A.c; ++i) {
      double s = 0;
      const double *ai = &A.a[(size_t)i * A.r], *bj = &B.a[(size_t)j * B.r];
      for (int k = 0; k < A.r; ++k) s += ai[k] * bj[k];
      C(i, j) = s;
    }
  return C;
}
// C = A * B
Mat gemm_nn(const Mat &A, const Mat &B) {
  Mat C(A.r, B.c);
  for (int j = 0; j < B.c; ++j)
    for (int k = 0; k < A.c; ++k) {
      const double b = B(k, j);
      if (b == 0.0) continue;
      const double *ak = &A.a[(size_t)k * A.r];
      double *cj = &C.a[(size_t)j * C.r];
      for (int i = 0; i < A.r; ++i) cj[i] += ak[i] * b;
    }
  return C;
}
std::vector<double> gemv_n(const Mat &A, const std::vector<double> &x) {
  std::vector<double> y(A.r, 0.0);
  for (int k = 0; k < A.c; ++k) {
    const double *ak = &A.a[(size_t)k * A.r];
    for (int i = 0; i < A.r; ++i) y[i] += ak[i] * x[k];
  }
  return y;
}
std::vector<double> gemv_t(const Mat &A, const std::vector<double> &x) {
  std::vector<double> y(A.c, 0.0);
  for (int j = 0; j < A.c; ++j) {
    const double *aj = &A.a[(size_t)j * A.r];
    double s = 0;
    for (int i = 0; i < A.r; ++i) s += aj[i] * x[i];
    y[j] = s;
  }
  return y;
}
// lower Cholesky of symmatu(A); false when not positive definite (dpotrf's test)
bool chol_lower(const Mat &A, Mat &L) {
  const int n = A.r;
  L = Mat(n, n);
  for (int j = 0; j < n; ++j)
    for (int i = j; i < n; ++i) L(i, j) = A(j, i);  // symmatu: upper triangle is the source
  for (int j = 0; j < n; ++j) {
    double d = L(j, j);
    for (int k = 0; k < j; ++k) d -= L(j, k) * L(j, k);
    if (!(d > 0.0)) return false;
    d = std::sqrt(d);
    L(j, j) = d;
    for (int i = j + 1; i < n; ++i) {
      double s = L(i, j);
      for (int k = 0; k < j; ++k) s -= L(i, k) * L(j, k);
      L(i, j) = s / d;
    }
  }
  return true;
}
Mat inv_trimatl(const Mat &L) {
  const int n = L.r;
  Mat X(n, n);
  for (int j = 0; j < n; ++j)
    for (int i = j; i < n; ++i) {
      double s = (i == j) ? 1.0 : 0.0;
      for (int k = j; k < i; ++k) s -= L(i, k) * X(k, j);
      X(i, j) = s / L(i, i);
    }
  return X;
}
// Li' * Li for lower-triangular Li (the reference forms the dense product, spamtree_model.cpp:867, 906, 912)
Mat gram_lower(const Mat &Li) {
  const int n = Li.r;
  Mat G(n, n);
  for (int j = 0; j < n; ++j)
    for (int i = 0; i <= j; ++i) {
      double s = 0;
      for (int k = j; k < n; ++k) s += Li(k, i) * Li(k, j);
      G(i, j) = s;
      G(j, i) = s;
    }
  return G;
}

struct CovPar {
  int q = 1, ncb = 1;
  std::vector<double> ai1, ai2, phi, tmv;
  Mat D;
  void transform(const double *theta, int ntheta) {  // covariance_functions.cpp:34-75, 77-92
    const int npars = 3 * q + ncb;
    ai1.assign(theta, theta + q);
    ai2.assign(theta + q, theta + 2 * q);
    phi.assign(theta + 2 * q, theta + 3 * q);
    tmv.assign(theta + 3 * q, theta + 3 * q + ncb);
    const int k = ntheta - npars;
    if (k > 0) {
      D = Mat(q, q);
      int ix = 0;
      for (int j = 0; j < q; ++j)
        for (int i = j + 1; i < q; ++i) { D(i, j) = theta[npars + ix]; D(j, i) = theta[npars + ix]; ++ix; }
    } else {
      D = Mat(1, 1);
    }
  }
};

struct Data {  // SpamTreeMVData, tree_utils.h:63-102
  std::vector<Mat> Kxc, Kxx_inv, H, prec, Kxx_invchol, Rcc_invchol, AK_uP_all, AK_uP_u_all, Smu_children;
  std::vector<std::vector<Mat>> Sigi_children;
  std::vector<std::vector<double>> prec_noref, ccholprecdiag;
  std::vector<double> wcore, logdet_c, loglik_c;
  std::vector<int> has_updated;
  std::vector<double> theta;
  double logdetCi = 0, loglik_w = 0;
};

struct RefModel {
  long long n_all = 0;
  int q = 1, p = 1, nb = 0, n_actual = 0;
  bool reference_distance = false, quirks = true, limited = false;
  std::vector<double> y, X, cx, cy, w, XB, tausq_inv_long, tausq_inv;
  std::vector<int> mv;
  std::vector<char> avail;
  std::vector<long long> na_ix_all;
  std::vector<std::vector<int>> indexing, parents, children, parents_indexing, u_by_group;
  std::vector<std::vector<int>> this_is_jth_child, dim_by_parent;
  std::vector<int> block_ct_obs, block_is_reference, group_of, blocks_not_empty;
  std::vector<int> res_is_ref;
  CovPar cp;
  Data dat[2];
  int slot_map[2] = {0, 1};
  int last_errtype = -1;
  std::string err;

  double cov_entry(int i, int j) const {
    double h;
    if (q == 1 && reference_distance) {  // cexpcov's cancellation form (covariance_functions.cpp:98-108), plain arithmetic
      const double pm = cx[i] * cx[i] + cy[i] * cy[i], qm = cx[j] * cx[j] + cy[j] * cy[j];
      volatile double xy = cx[i] * cx[j];
      volatile double xy2 = cy[i] * cy[j];
      h = std::sqrt(std::fabs(qm + pm - 2.0 * (xy + xy2)));
    } else {
      const double dx = cx[i] - cx[j], dy = cy[i] - cy[j];
      h = std::sqrt(dx * dx + dy * dy);
    }
    if (q == 1) return cp.ai1[0] * std::exp(-cp.tmv[0] * h);
    const int vi = mv[i], vj = mv[j];
    const double v = cp.D(vi, vj);
    double cb;
    if (q > 2) {
      const double ps = std::exp(0.5 * cp.tmv[1] * std::log1p(cp.tmv[0] * v));
      cb = std::exp(-cp.tmv[2] * (h / ps)) / (ps * ps);
    } else {
      const double ps = std::sqrt(v + 1.0);
      cb = std::exp(-cp.tmv[0] * (h / ps)) / (v + 1.0);
    }
    if (v == 0.0) return cp.ai1[vi] * cp.ai1[vi] * cb + cp.ai2[vi] * cp.ai2[vi] * std::exp(-cp.phi[vi] * h);
    return cp.ai1[vi] * cp.ai1[vj] * cb;
  }
  Mat covmat(const std::vector<int> &i1, const std::vector<int> &i2) const {
    Mat K((int)i1.size(), (int)i2.size());
    for (int j = 0; j < K.c; ++j)
      for (int i = 0; i < K.r; ++i) K(i, j) = cov_entry(i1[i], i2[j]);
    return K;
  }
  std::vector<double> gather(const std::vector<double> &v, const std::vector<int> &ix) const {
    std::vector<double> o(ix.size());
    for (size_t k = 0; k < ix.size(); ++k) o[k] = v[ix[k]];
    return o;
  }

  void init_data(Data &d) {  // spamtree_model.cpp:422-503 (allocations are the reference's)
    d.Kxc.resize(nb); d.Kxx_inv.resize(nb); d.H.resize(nb); d.prec.resize(nb); d.Kxx_invchol.resize(nb); d.Rcc_invchol.resize(nb);
    d.AK_uP_all.resize(nb); d.AK_uP_u_all.resize(nb); d.Smu_children.resize(nb); d.Sigi_children.resize(nb);
    d.prec_noref.resize(nb); d.ccholprecdiag.resize(nb);
    d.wcore.assign(nb, 0); d.logdet_c.assign(nb, 0); d.loglik_c.assign(nb, 0); d.has_updated.assign(nb, 0);
    for (int u = 0; u < nb; ++u) {
      const int m = (int)indexing[u].size(), P = (int)parents_indexing[u].size();
      if (!children[u].empty()) {
        d.Sigi_children[u].assign(children[u].size(), Mat(m, m));
        d.Smu_children[u] = Mat(m, (int)children[u].size());
      }
      if (block_ct_obs[u] > 0) d.Kxx_invchol[u] = Mat(P + m, P + m);
      d.H[u] = Mat(m, P);
      d.Kxc[u] = Mat(P, m);
      d.ccholprecdiag[u].assign(m, 0.0);
      if (block_is_reference[u]) { d.prec[u] = Mat(m, m); d.Rcc_invchol[u] = Mat(m, m); }
      else if (block_ct_obs[u] > 0) d.prec_noref[u].assign(m, 0.0);
      d.AK_uP_all[u] = Mat(P, m);
      d.AK_uP_u_all[u] = Mat(P, P);
    }
  }

  // ---- phase A: spamtree_model.cpp:834-998
  bool factor(Data &d) {
    cp.transform(d.theta.data(), (int)d.theta.size());
    int errtype = -1;
    for (int g = 0; g < n_actual; ++g) {
      const auto &lst = u_by_group[g];
#pragma omp parallel for schedule(dynamic, 1)
      for (int li = 0; li < (int)lst.size(); ++li) {
        const int u = lst[li];
        const auto &iu = indexing[u];
        const int m = (int)iu.size();
        std::vector<double> w_x = gather(w, iu);
        if (parents[u].empty()) {
          Mat Kcc = covmat(iu, iu), L;
          if (chol_lower(Kcc, L)) {
            d.Kxx_invchol[u] = inv_trimatl(L);
            d.Kxx_inv[u] = gram_lower(d.Kxx_invchol[u]);
            d.Rcc_invchol[u] = d.Kxx_invchol[u];
            d.prec[u] = d.Kxx_inv[u];
            std::vector<double> t = gemv_n(d.prec[u], w_x);
            d.wcore[u] = std::inner_product(w_x.begin(), w_x.end(), t.begin(), 0.0);
            for (int i = 0; i < m; ++i) d.ccholprecdiag[u][i] = d.Rcc_invchol[u](i, i);
          } else {
#pragma omp critical
            errtype = 1;
          }
          d.has_updated[u] = 1;
        } else {
          const int last_par = parents[u].back();
          const auto &pi = parents_indexing[u];
          const int P = (int)pi.size();
          d.Kxc[u] = covmat(pi, iu);
          std::vector<double> w_pars = gather(w, pi);
          d.H[u] = gemm_tn(d.Kxc[u], d.Kxx_inv[last_par]);  // m x P  (:887)
          std::vector<double> hw = gemv_n(d.H[u], w_pars);
          for (int i = 0; i < m; ++i) w_x[i] -= hw[i];
          if (res_is_ref[g] == 1) {
            Mat Kcc = covmat(iu, iu);
            Mat HK = gemm_nn(d.H[u], d.Kxc[u]);
            for (size_t k = 0; k < Kcc.a.size(); ++k) Kcc.a[k] -= HK.a[k];
            Mat L;
            if (chol_lower(Kcc, L)) {
              d.Rcc_invchol[u] = inv_trimatl(L);
              if (!children[u].empty() && limited) {
                Mat Kuu = covmat(iu, iu), Lu;                     // :901-903  Kxx_inv(u) = inv_sympd(Kcc)
                if (chol_lower(Kuu, Lu)) { d.Kxx_inv[u] = gram_lower(inv_trimatl(Lu)); d.has_updated[u] = 1; }
                else {
#pragma omp critical
                  errtype = 2;
                }
              } else if (!children[u].empty()) {
                // invchol_block_inplace_direct (tree_utils.cpp:194-208) + dense Gram (:904-906)
                Mat &O = d.Kxx_invchol[u];
                const Mat &LAi = d.Kxx_invchol[last_par];
                for (int j = 0; j < P; ++j)
                  for (int i = 0; i < P; ++i) O(i, j) = LAi(i, j);
                Mat RH = gemm_nn(d.Rcc_invchol[u], d.H[u]);
                for (int j = 0; j < P; ++j)
                  for (int i = 0; i < m; ++i) O(P + i, j) = -RH(i, j);
                for (int j = 0; j < m; ++j)
                  for (int i = 0; i < m; ++i) O(P + i, P + j) = d.Rcc_invchol[u](i, j);
                d.Kxx_inv[u] = gram_lower(O);
                d.has_updated[u] = 1;
              }
              d.prec[u] = gram_lower(d.Rcc_invchol[u]);
              std::vector<double> t = gemv_n(d.prec[u], w_x);
              d.wcore[u] = std::inner_product(w_x.begin(), w_x.end(), t.begin(), 0.0);
              for (int i = 0; i < m; ++i) d.ccholprecdiag[u][i] = d.Rcc_invchol[u](i, i);
            } else {
#pragma omp critical
              errtype = 2;
            }
          } else {
            d.wcore[u] = 0;
            for (int ix = 0; ix < m; ++ix) {
              const double kcc = cov_entry(iu[ix], iu[ix]);
              double s = 0;
              for (int k = 0; k < P; ++k) s += d.H[u](ix, k) * d.Kxc[u](k, ix);
              const double rr = kcc - s;
              if (rr > 0.0) {
                const double ri = 1.0 / std::sqrt(rr);
                d.ccholprecdiag[u][ix] = ri;
                d.prec_noref[u][ix] = ri * ri;
                d.wcore[u] += w_x[ix] * d.prec_noref[u][ix] * w_x[ix];
              } else {
#pragma omp critical
                errtype = 3;
              }
            }
          }
        }
        double ld = 0;
        for (int i = 0; i < m; ++i) ld += std::log(d.ccholprecdiag[u][i]);
        d.logdet_c[u] = ld;
        d.loglik_c[u] = (m + 0.0) * HL2PI - 0.5 * d.wcore[u];
      }
      if (errtype > 0) { last_errtype = errtype; return false; }
    }
    d.logdetCi = std::accumulate(d.logdet_c.begin(), d.logdet_c.end(), 0.0);
    d.loglik_w = d.logdetCi + std::accumulate(d.loglik_c.begin(), d.loglik_c.end(), 0.0);
    last_errtype = -1;
    return true;
  }

  // ---- phase B: spamtree_model.cpp:1011-1226 (need_update = true)
  int sample_w(Data &pd, const double *z) {
    int errtype = -1;
    for (int g = n_actual - 1; g >= 0; --g) {
      const auto &lst = u_by_group[g];
#pragma omp parallel for schedule(dynamic, 1)
      for (int li = 0; li < (int)lst.size(); ++li) {
        const int u = lst[li];
        const auto &iu = indexing[u];
        const auto &pi = parents_indexing[u];
        const int m = (int)iu.size(), P = (int)pi.size();
        if (res_is_ref[g] == 1) {
          std::vector<double> Smu(m, 0.0);
          Mat Sigi = pd.prec[u];
          if (!parents[u].empty()) {  // AK_uP_all = H' * prec  (:1046)
            Mat &A = pd.AK_uP_all[u];
            for (int j = 0; j < m; ++j)
              for (int i = 0; i < P; ++i) {
                double s = 0;
                for (int k = 0; k < m; ++k) s += pd.H[u](k, i) * pd.prec[u](k, j);
                A(i, j) = s;
              }
          }
          for (size_t c = 0; c < children[u].size(); ++c)  // arma::sum(cube, 2)  (:1049)
            for (size_t k = 0; k < Sigi.a.size(); ++k) Sigi.a[k] += pd.Sigi_children[u][c].a[k];
          for (int i = 0; i < m; ++i) Sigi(i, i) += tausq_inv_long[iu[i]];
          Mat L, Sc;
          if (chol_lower(Sigi, L)) Sc = inv_trimatl(L);
          else {
#pragma omp critical
            errtype = 10;
            Sc = Mat(m, m);
          }
          if (!parents[u].empty()) {
            std::vector<double> t = gemv_t(pd.AK_uP_all[u], gather(w, pi));
            for (int i = 0; i < m; ++i) Smu[i] += t[i];
          }
          if (!children[u].empty())
            for (int c = 0; c < pd.Smu_children[u].c; ++c)
              for (int i = 0; i < m; ++i) Smu[i] += pd.Smu_children[u](i, c);
          for (int i = 0; i < m; ++i) Smu[i] += tausq_inv_long[iu[i]] * (y[iu[i]] - XB[iu[i]]);
          std::vector<double> t = gemv_n(Sc, Smu);
          for (int i = 0; i < m; ++i) t[i] += z[iu[i]];
          std::vector<double> wn = gemv_t(Sc, t);  // Sigi_chol' (Sigi_chol Smu + z)  (:1086)
          for (int i = 0; i < m; ++i) w[iu[i]] = wn[i];
        } else {
          std::vector<double> hw = gemv_n(pd.H[u], gather(w, pi));
          for (int ix = 0; ix < m; ++ix) {
            const double tsqi = tausq_inv_long[iu[ix]];
            const double sig = pd.prec_noref[u][ix] + tsqi;
            const double smu = pd.prec_noref[u][ix] * hw[ix] + tsqi * (y[iu[ix]] - XB[iu[ix]]);
            double c = 0;
            if (sig > 0.0) c = 1.0 / std::sqrt(sig);
            else {
#pragma omp critical
              errtype = 11;
            }
            w[iu[ix]] = c * c * smu + c * z[iu[ix]];
            for (int k = 0; k < P; ++k) pd.AK_uP_all[u](k, ix) = pd.H[u](ix, k) * pd.prec_noref[u][ix];
          }
        }
        if (!parents[u].empty()) {
          pd.AK_uP_u_all[u] = gemm_nn(pd.AK_uP_all[u], pd.H[u]);  // P x P  (:1162)
          const Mat &G = pd.AK_uP_u_all[u];
          std::vector<double> w_par = gather(w, pi), w_u = gather(w, iu);
          for (size_t pp = 0; pp < parents[u].size(); ++pp) {
            const int up = parents[u][pp];
            const int c_ix = this_is_jth_child[u][pp];
            const int first = dim_by_parent[u][pp], last = dim_by_parent[u][pp + 1], ma = last - first;
            Mat &S = pd.Sigi_children[up][c_ix];
            for (int j = 0; j < ma; ++j)
              for (int i = 0; i < ma; ++i) S(i, j) = G(first + i, first + j);
            for (int i = 0; i < ma; ++i) {
              double s = 0;
              for (int k = 0; k < m; ++k) s += pd.AK_uP_all[u](first + i, k) * w_u[k];
              for (int k = 0; k < P; ++k)
                if (k < first || k >= last) s -= G(first + i, k) * w_par[k];
              pd.Smu_children[up](i, c_ix) = s;
            }
          }
        }
      }
    }
    return errtype > 0 ? errtype : 0;
  }

  // ---- phase C: spamtree_model.cpp:781-826
  void loglik_w(Data &d) {
#pragma omp parallel for schedule(dynamic, 16)
    for (int bi = 0; bi < (int)blocks_not_empty.size(); ++bi) {
      const int u = blocks_not_empty[bi];
      const auto &iu = indexing[u];
      const int m = (int)iu.size();
      std::vector<double> w_x = gather(w, iu);
      if (!parents[u].empty()) {
        std::vector<double> hw = gemv_n(d.H[u], gather(w, parents_indexing[u]));
        for (int i = 0; i < m; ++i) w_x[i] -= hw[i];
      }
      if (block_is_reference[u]) {
        std::vector<double> t = gemv_n(d.prec[u], w_x);
        d.wcore[u] = std::inner_product(w_x.begin(), w_x.end(), t.begin(), 0.0);
      } else {
        d.wcore[u] = 0;
        for (int ix = 0; ix < m; ++ix) d.wcore[u] += w_x[ix] * d.prec_noref[u][ix] * w_x[ix];
      }
      d.loglik_c[u] = (m + 0.0) * HL2PI - 0.5 * d.wcore[u];
    }
    d.logdetCi = std::accumulate(d.logdet_c.begin(), d.logdet_c.end(), 0.0);
    d.loglik_w = d.logdetCi + std::accumulate(d.loglik_c.begin(), d.loglik_c.end(), 0.0);
  }
};

}  // namespace

extern "C" {

// Inputs as spamtree_mv_mcmc receives them (CSR lists, 0-based; mv_id 1-based; y NaN = NA).
void *refcpu_create(long long n_all, int q, int p, long long n_blocks, int n_groups, const double *y, const double *X, const double *coords,
                    const int64_t *mv_id, const int64_t *res_is_ref, const int64_t *block_names, const int64_t *block_groups,
                    const int64_t *idx_ptr, const int64_t *idx, const int64_t *par_ptr, const int64_t *par, const int64_t *chi_ptr,
                    const int64_t *chi, int reference_distance, int reference_quirks, int num_threads) {
#ifdef _OPENMP
  if (num_threads > 0) omp_set_num_threads(num_threads);
#endif
  RefModel *M = new RefModel();
  M->n_all = n_all; M->q = q; M->p = p; M->nb = (int)n_blocks;
  M->reference_distance = reference_distance != 0; M->quirks = (reference_quirks & 1) != 0;
  M->limited = (reference_quirks & 2) != 0;   // limited_tree = TRUE (spamtree_model.cpp:901-903): single parents, Kxx_inv(u) = inv_sympd(K_uu)
  M->cp.q = q; M->cp.ncb = q > 2 ? 3 : 1;
  M->y.assign(y, y + n_all); M->X.assign(X, X + (size_t)n_all * p);
  M->cx.assign(coords, coords + n_all); M->cy.assign(coords + n_all, coords + 2 * n_all);
  M->mv.resize(n_all); M->avail.resize(n_all);
  for (long long i = 0; i < n_all; ++i) {
    M->mv[i] = (int)mv_id[i] - 1;
    M->avail[i] = std::isfinite(y[i]) ? 1 : 0;
    if (M->avail[i]) M->na_ix_all.push_back(i); else M->y[i] = 0.0;
  }
  M->w.assign(n_all, 0.0); M->XB.assign(n_all, 0.0); M->tausq_inv_long.assign(n_all, 1.0); M->tausq_inv.assign(q, 1.0);
  M->res_is_ref.assign(res_is_ref, res_is_ref + n_groups);
  const int nb = M->nb;
  M->indexing.resize(nb); M->parents.resize(nb); M->children.resize(nb); M->parents_indexing.resize(nb);
  for (int u = 0; u < nb; ++u) {
    for (long long k = idx_ptr[u]; k < idx_ptr[u + 1]; ++k) M->indexing[u].push_back((int)idx[k]);
    for (long long k = par_ptr[u]; k < par_ptr[u + 1]; ++k) M->parents[u].push_back((int)par[k]);
    for (long long k = chi_ptr[u]; k < chi_ptr[u + 1]; ++k) M->children[u].push_back((int)chi[k]);
  }
  for (int u = 0; u < nb; ++u)
    for (int a : M->parents[u]) M->parents_indexing[u].insert(M->parents_indexing[u].end(), M->indexing[a].begin(), M->indexing[a].end());
  M->block_ct_obs.assign(nb, 0);
  for (int u = 0; u < nb; ++u)
    for (int r : M->indexing[u]) M->block_ct_obs[u] += M->avail[r];
  // groups (make_gibbs_groups :194-301)
  std::vector<long long> labels(block_groups, block_groups + nb);
  std::sort(labels.begin(), labels.end());
  labels.erase(std::unique(labels.begin(), labels.end()), labels.end());
  M->group_of.resize(nb);
  std::vector<std::vector<int>> tmp(labels.size());
  for (int i = 0; i < nb; ++i) {
    const int u = (int)block_names[i] - 1;
    const int g = (int)(std::lower_bound(labels.begin(), labels.end(), block_groups[u]) - labels.begin());
    M->group_of[u] = g;
    if (M->block_ct_obs[u] > 0) tmp[g].push_back(u);
  }
  M->n_actual = 0;
  for (auto &t : tmp) M->n_actual += t.empty() ? 0 : 1;
  M->u_by_group.assign(tmp.begin(), tmp.begin() + M->n_actual);
  M->block_is_reference.assign(nb, 1);
  for (int i = 0; i < nb; ++i) {
    const int u = (int)block_names[i] - 1;
    if (M->block_ct_obs[u] > 0) {
      M->blocks_not_empty.push_back(u);
      const int g = M->group_of[u];
      if (g < n_groups && res_is_ref[g] == 0 && g < M->n_actual) M->block_is_reference[u] = 0;
    } else {
      M->block_is_reference[u] = 0;
    }
  }
  // init_finalize (:355-420)
  M->dim_by_parent.resize(nb); M->this_is_jth_child.resize(nb);
  for (int u = 0; u < nb; ++u) {
    M->dim_by_parent[u].assign(M->parents[u].size() + 1, 0);
    for (size_t j = 0; j < M->parents[u].size(); ++j)
      M->dim_by_parent[u][j + 1] = M->dim_by_parent[u][j] + (int)M->indexing[M->parents[u][j]].size();
    M->this_is_jth_child[u].assign(M->parents[u].size(), 0);
    if (M->block_ct_obs[u] > 0)
      for (size_t pp = 0; pp < M->parents[u].size(); ++pp) {
        const auto &ch = M->children[M->parents[u][pp]];
        M->this_is_jth_child[u][pp] = (int)(std::lower_bound(ch.begin(), ch.end(), u) - ch.begin());
      }
  }
  M->init_data(M->dat[0]);
  M->init_data(M->dat[1]);  // alter_data = param_data (:499): same all-zero caches
  return M;
}
void refcpu_destroy(void *h) { delete (RefModel *)h; }

// thread count of the per-level `#pragma omp parallel for` loops from now on (the reference's num_threads argument,
// /root/reference/src/spamtree_fit.cpp:56-58): bench.py times one model at several counts without re-allocating its caches
void refcpu_set_threads(int num_threads) {
#ifdef _OPENMP
  if (num_threads > 0) omp_set_num_threads(num_threads);
#else
  (void)num_threads;
#endif
}
int refcpu_factor(void *h, int slot, const double *theta, int ntheta, double *loglik) {
  RefModel *M = (RefModel *)h;
  Data &d = M->dat[M->slot_map[slot]];
  d.theta.assign(theta, theta + ntheta);
  const bool ok = M->factor(d);
  if (loglik) *loglik = d.loglik_w;
  return ok ? 0 : M->last_errtype;
}
int refcpu_sample_w(void *h, const double *z) { RefModel *M = (RefModel *)h; return M->sample_w(M->dat[M->slot_map[0]], z); }
double refcpu_loglik_w(void *h, int slot) {
  RefModel *M = (RefModel *)h;
  Data &d = M->dat[M->slot_map[slot]];
  M->loglik_w(d);
  return d.loglik_w;
}
void refcpu_swap(void *h) { RefModel *M = (RefModel *)h; std::swap(M->slot_map[0], M->slot_map[1]); }
void refcpu_set_w(void *h, const double *w) { RefModel *M = (RefModel *)h; M->w.assign(w, w + M->n_all); }
void refcpu_get_w(void *h, double *w) { RefModel *M = (RefModel *)h; std::memcpy(w, M->w.data(), M->n_all * sizeof(double)); }
void refcpu_set_tausq_inv(void *h, const double *t) {
  RefModel *M = (RefModel *)h;
  M->tausq_inv.assign(t, t + M->q);
  for (long long i = 0; i < M->n_all; ++i) M->tausq_inv_long[i] = t[M->mv[i]];
}
void refcpu_set_beta(void *h, const double *B) {  // p x q column-major; XB = X * Bcoeff[:, mv]  (:127, 1382)
  RefModel *M = (RefModel *)h;
  for (long long i = 0; i < M->n_all; ++i) {
    double s = 0;
    for (int j = 0; j < M->p; ++j) s += M->X[(size_t)j * M->n_all + i] * B[(size_t)M->mv[i] * M->p + j];
    M->XB[i] = s;
  }
}
// X_avail_j' (y_avail_j - w[ix_by_q_a(j)]) with the reference's index quirk (:1374-1375), and sum (y - XB - w)^2 (:1397-1400)
void refcpu_stats(void *h, double *xty, double *ssq) {
  RefModel *M = (RefModel *)h;
  std::fill(xty, xty + (size_t)M->p * M->q, 0.0);
  std::fill(ssq, ssq + M->q, 0.0);
  for (size_t t = 0; t < M->na_ix_all.size(); ++t) {
    const long long r = M->na_ix_all[t];
    const int v = M->mv[r];
    const double wq = M->quirks ? M->w[t] : M->w[r];
    for (int j = 0; j < M->p; ++j) xty[(size_t)v * M->p + j] += M->X[(size_t)j * M->n_all + r] * (M->y[r] - wq);
    const double e = M->y[r] - M->XB[r] - M->w[r];
    ssq[v] += e * e;
  }
}
// H (m x P, column-major) and Ri (m x m, or m diagonal values for a non-reference block) of block u
void refcpu_get_block(void *h, int slot, int u, double *H, double *Ri) {
  RefModel *M = (RefModel *)h;
  Data &d = M->dat[M->slot_map[slot]];
  if (H) std::memcpy(H, d.H[u].a.data(), d.H[u].a.size() * sizeof(double));
  if (Ri) {
    if (M->block_is_reference[u]) std::memcpy(Ri, d.Rcc_invchol[u].a.data(), d.Rcc_invchol[u].a.size() * sizeof(double));
    else std::memcpy(Ri, d.ccholprecdiag[u].data(), d.ccholprecdiag[u].size() * sizeof(double));
  }
}
void refcpu_get_comps(void *h, int slot, double *logdet_c, double *loglik_c) {
  RefModel *M = (RefModel *)h;
  Data &d = M->dat[M->slot_map[slot]];
  std::memcpy(logdet_c, d.logdet_c.data(), M->nb * sizeof(double));
  std::memcpy(loglik_c, d.loglik_c.data(), M->nb * sizeof(double));
}
int refcpu_threads() {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
}
