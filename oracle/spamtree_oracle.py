"""ORACLE -- test infrastructure only.  NumPy/SciPy FP64 restatement of spamtree's per-sweep hot path.

STATUS: **parity unpinned**.  The reference (/root/reference, R package spamtree 0.2.1) ships no tests, no
golden vectors and no recorded outputs, and it cannot be built here (needs R, Rcpp,
RcppArmadillo, LAPACK/BLAS; none present, no network).  This restatement is therefore pinned only by
(a) the dense brute-force identities in tests/test_oracle_identities.py (exact GP on a one-level tree, the
treed-DAG precision matrix, the exact Gaussian full conditional of a block, inverse-Cholesky extension),
(b) 50-digit mpmath values of the Apanasovich-Genton cross-covariance on the man-page inputs, and
(c) the Random123 known-answer vectors for the Philox generator.  It follows the reference statement by
statement, quirks included; each function cites the lines it restates.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.  The product
(spamtree_amd/) never does.

Restated files (paths relative to /root/reference/):
  src/covariance_functions.{h,cpp}   -> CovarianceParams, cexpcov, C_base, mvCovAG20107, Covariancef,
                                        CrossCovarianceAG10
  src/tree_utils.cpp:194-208         -> invchol_block_inplace_direct
  src/spamtree_model.cpp             -> SpamTreeMV (ctor :8-192, groups :194-301, na_study :303-313,
                                        init_indexing :315-353, init_finalize :355-420, init_model_data :422-503,
                                        get_loglik_w_std :781-826, get_loglik_comps_w_std :834-998,
                                        gibbs_sample_w_std :1011-1226, predict_std :1234-1358,
                                        gibbs_sample_beta :1364-1391, gibbs_sample_tausq :1393-1417)
  src/mh_adapt.{h,cpp}               -> RAMAdapt, do_I_accept, par_huvtransf_*, unif_bounds, calc_jacobian
  src/spamtree_fit.cpp:5-430         -> spamtree_mv_mcmc
The R-level RNG (arma::randn -> R's generator, R::runif, R::rgamma) is replaced by the explicit counter-based
Philox4x32-10 streams of :class:`StRng`; the draw ORDER per iteration is the reference's (Q6).
"""
from __future__ import annotations

import copy
import math
from typing import List, Optional

import numpy as np
from scipy.linalg import solve_triangular

HL2PI = -0.5 * math.log(2.0 * math.pi)          # spamtree_model.h:20


# ----------------------------------------------------------------------------------------------------------
# Counter-based RNG (replaces R's generator; documented contract shared with the HIP build)
# ----------------------------------------------------------------------------------------------------------
_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = 0x9E3779B9, 0xBB67AE85
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox4x32-10 (Salmon et al. 2011, Random123).  Inputs broadcastable uint32-valued arrays."""
    c0 = np.asarray(c0, dtype=np.uint64) & _MASK
    c1 = np.asarray(c1, dtype=np.uint64) & _MASK
    c2 = np.asarray(c2, dtype=np.uint64) & _MASK
    c3 = np.asarray(c3, dtype=np.uint64) & _MASK
    k0 = int(k0) & 0xFFFFFFFF
    k1 = int(k1) & 0xFFFFFFFF
    c0, c1, c2, c3 = np.broadcast_arrays(c0, c1, c2, c3)
    for _ in range(10):
        p0 = _M0 * c0
        p1 = _M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & _MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & _MASK
        c0, c1, c2, c3 = (hi1 ^ c1 ^ np.uint64(k0)), lo1, (hi0 ^ c3 ^ np.uint64(k1)), lo0
        k0 = (k0 + _W0) & 0xFFFFFFFF
        k1 = (k1 + _W1) & 0xFFFFFFFF
    return c0, c1, c2, c3


def _u01(a, b):
    """53-bit uniform in (0,1) from two 32-bit words."""
    a = np.asarray(a, dtype=np.uint64)
    b = np.asarray(b, dtype=np.uint64)
    return (((a >> np.uint64(5)) * np.uint64(67108864) + (b >> np.uint64(6))).astype(np.float64) + 0.5) \
        * (1.0 / 9007199254740992.0)


class StRng:
    """Streams: 0 sweep normals, 1 theta proposal, 2 MH uniform, 3 gamma, 4 beta normals, 5 yhat noise.

    counter = (index_lo, index_hi_or_outcome, iteration, stream); key = (seed_lo, seed_hi).
    normal = sqrt(-2 ln u1) cos(2 pi u2) with (u1,u2) from words (0,1) and (2,3) of one Philox block.
    """

    def __init__(self, seed: int):
        self.seed = int(seed)
        self.k0 = self.seed & 0xFFFFFFFF
        self.k1 = (self.seed >> 32) & 0xFFFFFFFF

    def _blk(self, idx, hi, it, stream):
        idx = np.asarray(idx, dtype=np.uint64)
        return philox4x32_10(idx & _MASK, np.asarray(hi, dtype=np.uint64), np.uint64(it), np.uint64(stream),
                             self.k0, self.k1)

    def normal(self, idx, hi, it, stream):
        x0, x1, x2, x3 = self._blk(idx, hi, it, stream)
        u1, u2 = _u01(x0, x1), _u01(x2, x3)
        return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * math.pi * u2)

    def uniform(self, idx, hi, it, stream):
        x0, x1, _, _ = self._blk(idx, hi, it, stream)
        return _u01(x0, x1)

    # --- the draws of one MCMC iteration, in the reference's order (Q6) ---
    def sweep_normals(self, it: int, n_all: int) -> np.ndarray:            # spamtree_model.cpp:1018
        rows = np.arange(n_all, dtype=np.uint64)
        return self.normal(rows, rows >> np.uint64(32), it, 0)

    def theta_normals(self, it: int, k: int) -> np.ndarray:                # spamtree_fit.cpp:211
        return self.normal(np.arange(k), 0, it, 1)

    def mh_uniform(self, it: int) -> float:                                # mh_adapt.h:30
        return float(self.uniform(0, 0, it, 2))

    def gamma(self, it: int, j: int, shape: float, scale: float) -> float:  # spamtree_model.cpp:1405
        """Marsaglia-Tsang (2000) for shape >= 1; attempt t uses counters 2t (normal) and 2t+1 (uniform)."""
        d = shape - 1.0 / 3.0
        c = 1.0 / math.sqrt(9.0 * d)
        t = 0
        while True:
            x = float(self.normal(2 * t, j, it, 3))
            u = float(self.uniform(2 * t + 1, j, it, 3))
            t += 1
            v = 1.0 + c * x
            if v <= 0.0:
                continue
            v = v * v * v
            if math.log(u) < 0.5 * x * x + d - d * v + d * math.log(v):
                return d * v * scale

    def beta_normals(self, it: int, j: int, p: int) -> np.ndarray:          # spamtree_model.cpp:1378
        return self.normal(np.arange(p), j, it, 4)

    def yhat_normals(self, it: int, n_all: int) -> np.ndarray:              # spamtree_fit.cpp:384
        rows = np.arange(n_all, dtype=np.uint64)
        return self.normal(rows, rows >> np.uint64(32), it, 5)


# ----------------------------------------------------------------------------------------------------------
# covariance_functions.{h,cpp}
# ----------------------------------------------------------------------------------------------------------
def vec_to_symmat(x):
    """covariance_functions.cpp:77-92 (column-wise fill of the strict lower triangle, then symmatl)."""
    x = np.asarray(x, dtype=np.float64)
    k = x.size
    p = int((1 + math.sqrt(1 + 8 * k)) / 2)
    res = np.zeros((p, p))
    ix = 0
    start_i = 1
    for j in range(p):
        for i in range(start_i, p):
            res[i, j] = x[ix]
            ix += 1
        start_i += 1
    return np.tril(res) + np.tril(res, -1).T


class CovarianceParams:
    """covariance_functions.h:7-33, covariance_functions.cpp:10-75."""

    def __init__(self, dd: int, q: int, covmodel: int = -1):
        self.q = q
        self.covariance_model = covmodel
        self.npars = 0
        self.n_cbase = 1
        if self.covariance_model == -1:
            if dd == 2:
                self.covariance_model = 0
                self.n_cbase = 3 if q > 2 else 1
                self.npars = 3 * q + self.n_cbase
            else:
                if q > 1:
                    raise ValueError("Multivariate on many inputs not implemented yet.")
                self.covariance_model = 1
        if self.covariance_model == 2:
            self.n_cbase = 3 if q > 2 else 1
            self.npars = 3 * q + self.n_cbase + 1
        self.ai1 = self.ai2 = self.phi_i = self.thetamv = None
        self.Dmat = np.zeros((1, 1))

    def transform(self, theta):
        theta = np.asarray(theta, dtype=np.float64)
        q = self.q
        if self.covariance_model == 0:
            k = theta.size - self.npars
            cp = theta[: self.npars]
            self.ai1 = cp[0:q].copy()
            self.ai2 = cp[q:2 * q].copy()
            self.phi_i = cp[2 * q:3 * q].copy()
            self.thetamv = cp[3 * q:3 * q + self.n_cbase].copy()
            self.Dmat = vec_to_symmat(theta[self.npars:self.npars + k]) if k > 0 else np.zeros((1, 1))
        else:
            raise NotImplementedError("covariance models 1, 2 are unreachable from spamtree() (R stops for dd>2)")


def fphi(x, c):                         # covariance_functions.h:40-42
    return np.exp(-c * x)


def sqrt_fpsi(x, a, beta):              # covariance_functions.h:44-48
    return np.exp(0.5 * beta * np.log1p(a * x))


def _two_prod(a, b):
    """a*b = p + e exactly (Dekker / Veltkamp splitting), elementwise."""
    p = a * b
    c = 134217729.0                                      # 2^27 + 1
    a1 = c * a; ah = a1 - (a1 - a); al = a - ah
    b1 = c * b; bh = b1 - (b1 - b); bl = b - bh
    e = ((ah * bh - p) + ah * bl + al * bh) + al * bl
    return p, e


def _fma(a, b, c):
    """Emulation of a fused multiply-add (one rounding of a*b + c) by double-double accumulation."""
    p, e = _two_prod(a, b)
    s = p + c
    bb = s - p
    t = (p - (s - bb)) + (c - bb)                         # two-sum error of p + c
    return s + (t + e)


def cexpcov(x, y, sigmasq, phi, same=False, reference_distance=False):
    """covariance_functions.cpp:95-111.

    reference_distance=True evaluates the reference's cancellation form |x|^2+|y|^2-2x.y in plain (non-FMA)
    double arithmetic (Q1; R's reference BLAS); reference_distance="fma" accumulates the cross product x.y' the way
    an FMA BLAS kernel (OpenBLAS / MKL dgemm) does -- acc = fma(x_d, y_d, acc) -- while |x|^2 stays the plain
    sum(x % x, 1) of the source, which is what makes self-distances non-zero; the default computes
    h = sqrt(dx^2+dy^2) directly, which is what the HIP build does.
    """
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    if reference_distance:
        pmag = np.sum(x * x, axis=1)
        qmag = np.sum(y * y, axis=1)
        xy = np.zeros((x.shape[0], y.shape[0]))
        for d in range(x.shape[1]):
            if reference_distance == "fma":
                xy = _fma(x[:, d][:, None] + 0.0 * xy, y[:, d][None, :] + 0.0 * xy, xy)
            else:
                xy = xy + x[:, d][:, None] * y[:, d][None, :]
        h2 = np.abs(qmag[None, :] + pmag[:, None] - 2.0 * xy)
        return sigmasq * np.exp(-phi * np.sqrt(h2))
    dx = x[:, 0][:, None] - y[:, 0][None, :]
    dy = x[:, 1][:, None] - y[:, 1][None, :]
    return sigmasq * np.exp(-phi * np.sqrt(dx * dx + dy * dy))


def C_base(h, v, params, q):
    """covariance_functions.cpp:113-135 (u, dim unused on the reachable path)."""
    if q > 2:
        a_psi1, beta_psi1, c_phi1 = params[0], params[1], params[2]
        psi1_sqrt = sqrt_fpsi(v, a_psi1, beta_psi1)
        return fphi(h / psi1_sqrt, c_phi1) / (psi1_sqrt * psi1_sqrt)
    if q == 2:
        c_phi1 = params[0]
        psi1_sqrt = np.sqrt(v + 1.0)
        return fphi(h / psi1_sqrt, c_phi1) / (v + 1.0)
    return fphi(h, params[0])


def _ag_entries(c1, v1, c2, v2, ai1, ai2, phi_i, thetamv, Dmat):
    """Entry-wise Apanasovich-Genton form shared by mvCovAG20107_inplace (:213-286) and CrossCovarianceAG10."""
    p = Dmat.shape[1]
    dx = c1[:, 0][:, None] - c2[:, 0][None, :]
    dy = c1[:, 1][:, None] - c2[:, 1][None, :]
    h = np.sqrt(dx * dx + dy * dy)
    v = Dmat[np.ix_(v1, v2)]
    a1i = ai1[v1][:, None]
    a1j = ai1[v2][None, :]
    a2i = ai2[v1][:, None]
    same_v = (v == 0)
    res_same = a1i * a1i * C_base(h, 0.0, thetamv, p) + a2i * a2i * fphi(h, phi_i[v1][:, None])
    with np.errstate(all="ignore"):
        res_diff = a1i * a1j * C_base(h, v, thetamv, p)
    return np.where(same_v, res_same, res_diff)


def mvCovAG20107(coords, qv_block, ind1, ind2, covpars, same=False, reference_distance=False):
    """covariance_functions.cpp:213-298."""
    d = coords.shape[1]
    p = covpars.Dmat.shape[1]
    if d == 2 and p < 2:
        return cexpcov(coords[ind1], coords[ind2], covpars.ai1[0], covpars.thetamv[0], same, reference_distance)
    res = _ag_entries(coords[ind1], qv_block[ind1], coords[ind2], qv_block[ind2],
                      covpars.ai1, covpars.ai2, covpars.phi_i, covpars.thetamv, covpars.Dmat)
    if same:
        res = np.triu(res) + np.triu(res, 1).T          # symmatu (:258)
    return res


def Covariancef(coords, qv_block, ind1, ind2, covpars, same=False, reference_distance=False):
    """covariance_functions.cpp:397-436 (only model 0 is reachable)."""
    ind1 = np.asarray(ind1, dtype=np.int64)
    ind2 = np.asarray(ind2, dtype=np.int64)
    if covpars.covariance_model != 0:
        raise NotImplementedError
    return mvCovAG20107(coords, qv_block, ind1, ind2, covpars, same, reference_distance)


def CrossCovarianceAG10(coords1, mv1, coords2, mv2, ai1, ai2, phi_i, thetamv, Dmat):
    """covariance_functions.cpp:301-355 (mv ids 1-based, as exported to R)."""
    Dmat = np.atleast_2d(np.asarray(Dmat, dtype=np.float64))
    coords1 = np.asarray(coords1, dtype=np.float64)
    coords2 = np.asarray(coords2, dtype=np.float64)
    if coords1.shape[1] == 2 and Dmat.shape[1] < 2:
        raise ValueError("Invalid Dmat for multivariate data")
    return _ag_entries(coords1, np.asarray(mv1, dtype=np.int64) - 1, coords2, np.asarray(mv2, dtype=np.int64) - 1,
                       np.asarray(ai1, dtype=np.float64), np.asarray(ai2, dtype=np.float64),
                       np.asarray(phi_i, dtype=np.float64), np.asarray(thetamv, dtype=np.float64), Dmat)


# ----------------------------------------------------------------------------------------------------------
# small dense helpers with Armadillo's semantics
# ----------------------------------------------------------------------------------------------------------
def _chol_lower(A):
    """arma::chol(A,"lower"): throws (here: LinAlgError) when A is not positive definite."""
    A = np.asarray(A, dtype=np.float64)
    if not np.all(np.isfinite(A)):
        raise np.linalg.LinAlgError("non-finite")
    return np.linalg.cholesky(A)


def _inv_trimatl(L):
    """arma::inv(arma::trimatl(L))."""
    return solve_triangular(L, np.eye(L.shape[0]), lower=True)


def _symmatu(A):
    return np.triu(A) + np.triu(A, 1).T


def invchol_block_inplace_direct(output_inv, LAi, C_times_LAi, invcholSchur):
    """tree_utils.cpp:194-208."""
    a = LAi.shape[0]
    output_inv[:a, :a] = LAi
    output_inv[a:, :a] = -invcholSchur @ C_times_LAi
    output_inv[a:, a:] = invcholSchur


class SpamTreeMVData:
    """tree_utils.h:63-102 (per-block caches)."""

    def __init__(self, n_blocks):
        nb = n_blocks
        self.has_updated = np.zeros(nb, dtype=np.int64)
        self.wcore = np.zeros(nb)
        self.Kxc = [None] * nb
        self.Kxx_inv = [None] * nb
        self.w_cond_mean_K = [None] * nb
        self.w_cond_prec = [None] * nb
        self.w_cond_prec_noref = [None] * nb
        self.Kxx_invchol = [None] * nb
        self.Rcc_invchol = [None] * nb
        self.ccholprecdiag = [None] * nb
        self.Sigi_chol = [None] * nb
        self.Sigi_chol_noref = [None] * nb
        self.AK_uP_all = [None] * nb
        self.AK_uP_u_all = [None] * nb
        self.logdetCi_comps = np.zeros(nb)
        self.logdetCi = 0.0
        self.loglik_w_comps = np.zeros(nb)
        self.loglik_w = 0.0
        self.theta = None
        self.Sigi_children = [None] * nb
        self.Smu_children = [None] * nb


class SpamTreeMV:
    """Restatement of class SpamTreeMV (spamtree_model.h:22-212)."""

    def __init__(self, y, X, Z, coords, mv_id, blocking, gix_block, res_is_ref, parents, children, limited_tree,
                 block_names, block_groups, indexing, w, beta, theta, tausq_inv,
                 reference_distance=False, reference_quirks=True):
        # spamtree_model.cpp:8-192
        self.reference_distance = reference_distance
        self.reference_quirks = reference_quirks
        self.y = np.asarray(y, dtype=np.float64).reshape(-1).copy()
        self.X = np.asarray(X, dtype=np.float64).copy()
        self.Z = np.asarray(Z, dtype=np.float64)
        self.coords = np.asarray(coords, dtype=np.float64)
        self.mv_id = np.asarray(mv_id, dtype=np.int64)
        self.qvblock_c = self.mv_id - 1
        self.blocking = np.asarray(blocking)
        self.gix_block = np.asarray(gix_block)
        self.res_is_ref = np.asarray(res_is_ref, dtype=np.int64)
        self.parents = [np.asarray(p, dtype=np.int64) for p in parents]
        self.children = [np.asarray(c, dtype=np.int64) for c in children]
        self.limited_tree = bool(limited_tree)
        self.block_names = np.asarray(block_names, dtype=np.int64)
        self.block_groups = np.asarray(block_groups, dtype=np.float64)
        self.block_groups_labels = np.unique(self.block_groups)
        self.n_gibbs_groups = self.block_groups_labels.size
        self.n_actual_groups = self.n_gibbs_groups
        self.n_blocks = self.block_names.size

        self.na_ix_all = np.nonzero(np.isfinite(self.y))[0]
        self.y_available = self.y[self.na_ix_all]
        self.X_available = self.X[self.na_ix_all]
        self.n = self.na_ix_all.size
        self.p = self.X.shape[1]
        self.q = np.unique(self.mv_id).size
        self.dd = self.coords.shape[1]
        qv_av = self.qvblock_c[self.na_ix_all]
        self.ix_by_q = [np.nonzero(self.qvblock_c == j)[0] for j in range(self.q)]
        self.ix_by_q_a = [np.nonzero(qv_av == j)[0] for j in range(self.q)]
        self.indexing = [np.asarray(ix, dtype=np.int64) for ix in indexing]

        self.tausq_inv = np.ones(self.q) * tausq_inv
        self.tausq_inv_long = np.ones(self.y.size) * tausq_inv
        self.XB = np.zeros(self.coords.shape[0])
        self.Bcoeff = np.zeros((self.p, self.q))
        beta = np.asarray(beta, dtype=np.float64).reshape(-1)
        for j in range(self.q):
            self.XB[self.ix_by_q[j]] = self.X[self.ix_by_q[j]] @ beta
            self.Bcoeff[:, j] = beta
        self.w = np.asarray(w, dtype=np.float64).reshape(-1).copy()
        self.predicting = True
        self.bigrnorm = np.zeros(self.coords.shape[0])

        self.init_indexing()
        self.na_study()
        self.y[~np.isfinite(self.y)] = 0.0                                   # :146
        self.XtX = [self.X_available[self.ix_by_q_a[j]].T @ self.X_available[self.ix_by_q_a[j]]
                    for j in range(self.q)]                                   # :151-155
        self.Vi = 0.01 * np.eye(self.p)
        self.bprim = np.zeros(self.p)
        self.Vim = self.Vi @ self.bprim
        self.make_gibbs_groups()
        self.init_finalize()
        self.init_model_data(np.asarray(theta, dtype=np.float64))
        self.covariance_model = 2 if self.dd == 3 else -1
        self.covpars = CovarianceParams(self.dd, self.q, self.covariance_model)

    # -- spamtree_model.cpp:303-313
    def na_study(self):
        self.block_ct_obs = np.zeros(self.n_blocks, dtype=np.int64)
        for i in range(self.n_blocks):
            self.block_ct_obs[i] = np.isfinite(self.y[self.indexing[i]]).sum()

    # -- spamtree_model.cpp:315-353
    def init_indexing(self):
        empty = np.zeros(0, dtype=np.int64)
        self.parents_indexing = [empty] * self.n_blocks
        self.children_indexing = [empty] * self.n_blocks
        for i in range(self.n_blocks):
            u = self.block_names[i] - 1
            if self.parents[u].size > 0:
                self.parents_indexing[u] = np.concatenate([self.indexing[pp] for pp in self.parents[u]])
            if self.children[u].size > 0:
                self.children_indexing[u] = np.concatenate([self.indexing[cc] for cc in self.children[u]])

    # -- spamtree_model.cpp:194-301
    def make_gibbs_groups(self):
        for g in range(self.n_gibbs_groups):
            for i in range(self.n_blocks):
                u = self.block_names[i] - 1
                if self.block_groups[u] == self.block_groups_labels[g] and self.indexing[u].size > 0:
                    for pp in self.parents[u]:
                        if self.block_groups[pp] == self.block_groups_labels[g]:
                            raise RuntimeError("same group")          # throw 1 (:212)
                    for cc in self.children[u]:
                        if self.block_groups[cc] == self.block_groups_labels[g]:
                            raise RuntimeError("same group")          # throw 1 (:220)
        temp = []
        for g in range(self.n_gibbs_groups):
            lst = [self.block_names[i] - 1 for i in range(self.n_blocks)
                   if self.block_groups[self.block_names[i] - 1] == self.block_groups_labels[g]
                   and self.block_ct_obs[self.block_names[i] - 1] > 0]
            temp.append(np.asarray(lst, dtype=np.int64))
        self.n_actual_groups = sum(1 for t in temp if t.size > 0)
        self.u_by_block_groups = [temp[g] for g in range(self.n_actual_groups)]
        self.block_is_reference = np.ones(self.n_blocks, dtype=np.int64)
        which_not_reference = np.nonzero(self.res_is_ref == 0)[0]
        ne, pr = [], []
        for i in range(self.n_blocks):
            u = self.block_names[i] - 1
            if self.block_ct_obs[u] > 0:
                ne.append(u)
                for r in which_not_reference:
                    if r < len(self.u_by_block_groups):
                        if np.any(self.u_by_block_groups[r] == u):
                            self.block_is_reference[u] = 0
                            break
            else:
                pr.append(u)
                self.block_is_reference[u] = 0
        self.blocks_not_empty = np.asarray(ne, dtype=np.int64)
        self.blocks_predicting = np.asarray(pr, dtype=np.int64)

    # -- spamtree_model.cpp:355-420
    def init_finalize(self):
        nb = self.n_blocks
        self.dim_by_parent = [None] * nb
        self.u_is_which_col_f = [None] * nb
        self.this_is_jth_child = [None] * nb
        for i in range(nb):
            u = self.block_names[i] - 1
            if self.indexing[u].size > 0:
                d = np.zeros(self.parents[u].size + 1)
                for j, pp in enumerate(self.parents[u]):
                    d[j + 1] = self.indexing[pp].size
                self.dim_by_parent[u] = np.cumsum(d).astype(np.int64)
        for i in range(nb):
            u = self.block_names[i] - 1
            self.u_is_which_col_f[u] = [None] * self.children[u].size
            self.this_is_jth_child[u] = np.zeros(self.parents[u].size, dtype=np.int64)
            for c, child in enumerate(self.children[u]):
                u_is_which = np.nonzero(self.parents[child] == u)[0][0]
                firstcol = self.dim_by_parent[child][u_is_which]
                lastcol = self.dim_by_parent[child][u_is_which + 1]
                dimen = self.parents_indexing[child].size
                result = np.arange(dimen)
                rowsel = np.zeros(dimen, dtype=bool)
                rowsel[firstcol:lastcol] = True
                self.u_is_which_col_f[u][c] = (result[rowsel], result[~rowsel])
            if self.block_ct_obs[u] > 0:
                for p_, up in enumerate(self.parents[u]):
                    self.this_is_jth_child[u][p_] = np.nonzero(self.children[up] == u)[0][0]

    # -- spamtree_model.cpp:422-503
    def init_model_data(self, theta_in):
        nb = self.n_blocks
        d = SpamTreeMVData(nb)
        d.theta = theta_in.copy()
        for i in range(nb):
            mi = self.indexing[i].size
            if self.children[i].size > 0:
                d.Sigi_children[i] = np.zeros((mi, mi, self.children[i].size))
                d.Smu_children[i] = np.zeros((mi, self.children[i].size))
            u = self.block_names[i] - 1
            mu, Pu = self.indexing[u].size, self.parents_indexing[u].size
            if self.block_ct_obs[u] > 0:
                d.Kxx_invchol[u] = np.zeros((Pu + mu, Pu + mu))
            d.w_cond_mean_K[u] = np.zeros((mu, Pu))
            d.Kxc[u] = np.zeros((Pu, mu))
            d.ccholprecdiag[u] = np.zeros(mu)
            if self.block_is_reference[u] == 1:
                d.w_cond_prec[u] = np.zeros((mu, mu))
                d.Rcc_invchol[u] = np.zeros((mu, mu))
                d.Sigi_chol[u] = np.zeros((mu, mu))
            elif self.block_ct_obs[u] > 0:
                d.w_cond_prec_noref[u] = [np.zeros((self.q, self.q)) for _ in range(mu)]
                d.Sigi_chol_noref[u] = [np.zeros((self.q, self.q)) for _ in range(mu)]
            d.AK_uP_all[u] = np.zeros((Pu, mu))
            d.AK_uP_u_all[u] = d.AK_uP_all[u] @ d.w_cond_mean_K[u]
        self.param_data = d
        self.alter_data = copy.deepcopy(d)

    def _cov(self, ind1, ind2, same):
        return Covariancef(self.coords, self.qvblock_c, ind1, ind2, self.covpars, same, self.reference_distance)

    # -- spamtree_model.cpp:781-826
    def get_loglik_w(self, data):
        for u in self.blocks_not_empty:
            w_x = self.w[self.indexing[u]].copy()
            if self.parents[u].size > 0:
                w_x -= data.w_cond_mean_K[u] @ self.w[self.parents_indexing[u]]
            if self.block_is_reference[u] == 1:
                data.wcore[u] = float(w_x @ data.w_cond_prec[u] @ w_x)
            else:
                data.wcore[u] = 0.0
                for ix in range(self.indexing[u].size):
                    data.wcore[u] += w_x[ix] * data.w_cond_prec_noref[u][ix][0, 0] * w_x[ix]
            data.loglik_w_comps[u] = (self.indexing[u].size + 0.0) * HL2PI - 0.5 * data.wcore[u]
        data.logdetCi = float(np.sum(data.logdetCi_comps))
        data.loglik_w = data.logdetCi + float(np.sum(data.loglik_w_comps))

    # -- spamtree_model.cpp:834-998; returns (ok, errtype)
    def get_loglik_comps_w(self, data) -> bool:
        self.covpars.transform(data.theta)
        errtype = -1
        for g in range(self.n_actual_groups):
            for u in self.u_by_block_groups[g]:
                w_x = self.w[self.indexing[u]].copy()
                if self.parents[u].size == 0:
                    Kcc = self._cov(self.indexing[u], self.indexing[u], True)
                    try:
                        data.Kxx_invchol[u] = _inv_trimatl(_chol_lower(Kcc))
                        data.Kxx_inv[u] = data.Kxx_invchol[u].T @ data.Kxx_invchol[u]
                        data.Rcc_invchol[u] = data.Kxx_invchol[u]
                        data.w_cond_prec[u] = data.Kxx_inv[u]
                        data.wcore[u] = float(w_x @ data.w_cond_prec[u] @ w_x)
                        data.ccholprecdiag[u] = np.diag(data.Rcc_invchol[u]).copy()
                    except np.linalg.LinAlgError:
                        errtype = 1
                    data.has_updated[u] = 1
                else:
                    last_par = self.parents[u][-1]
                    data.Kxc[u] = self._cov(self.parents_indexing[u], self.indexing[u], False)
                    w_pars = self.w[self.parents_indexing[u]]
                    data.w_cond_mean_K[u] = data.Kxc[u].T @ data.Kxx_inv[last_par]
                    w_x -= data.w_cond_mean_K[u] @ w_pars
                    if self.res_is_ref[g] == 1:
                        Kcc = self._cov(self.indexing[u], self.indexing[u], True)
                        try:
                            data.Rcc_invchol[u] = _inv_trimatl(_chol_lower(_symmatu(
                                Kcc - data.w_cond_mean_K[u] @ data.Kxc[u])))
                            if self.children[u].size > 0:
                                if self.limited_tree:
                                    data.Kxx_inv[u] = np.linalg.inv(Kcc)
                                else:
                                    invchol_block_inplace_direct(data.Kxx_invchol[u], data.Kxx_invchol[last_par],
                                                                 data.w_cond_mean_K[u], data.Rcc_invchol[u])
                                    data.Kxx_inv[u] = data.Kxx_invchol[u].T @ data.Kxx_invchol[u]
                                data.has_updated[u] = 1
                            data.w_cond_prec[u] = data.Rcc_invchol[u].T @ data.Rcc_invchol[u]
                            data.wcore[u] = float(w_x @ data.w_cond_prec[u] @ w_x)
                            data.ccholprecdiag[u] = np.diag(data.Rcc_invchol[u]).copy()
                        except np.linalg.LinAlgError:
                            errtype = 2
                    else:
                        data.wcore[u] = 0.0
                        for ix in range(self.indexing[u].size):
                            uix = self.indexing[u][ix:ix + 1]
                            Kcc = self._cov(uix, uix, True)
                            Kcx_xxi_xc = data.w_cond_mean_K[u][ix:ix + 1, :] @ data.Kxc[u][:, ix:ix + 1]
                            try:
                                Rinvchol = _inv_trimatl(_chol_lower(_symmatu(Kcc - Kcx_xxi_xc)))
                                data.ccholprecdiag[u][ix] = Rinvchol[0, 0]
                                data.w_cond_prec_noref[u][ix] = Rinvchol.T @ Rinvchol
                                data.wcore[u] += float(w_x[ix] * data.w_cond_prec_noref[u][ix][0, 0] * w_x[ix])
                            except np.linalg.LinAlgError:
                                errtype = 3
                with np.errstate(all="ignore"):
                    data.logdetCi_comps[u] = float(np.sum(np.log(data.ccholprecdiag[u])))
                data.loglik_w_comps[u] = (self.indexing[u].size + 0.0) * HL2PI - 0.5 * data.wcore[u]
            if errtype > 0:
                self.last_errtype = errtype
                return False                                             # :971-982 (Q5)
        data.logdetCi = float(np.sum(data.logdetCi_comps[: self.n_blocks]))
        data.loglik_w = data.logdetCi + float(np.sum(data.loglik_w_comps[: self.n_blocks]))
        self.last_errtype = -1
        return True

    # -- spamtree_model.cpp:1011-1226
    def gibbs_sample_w(self, z, need_update=True):
        pd = self.param_data
        self.bigrnorm = np.asarray(z, dtype=np.float64).copy()
        errtype = -1
        for g in range(self.n_actual_groups - 1, -1, -1):
            for u in self.u_by_block_groups[g]:
                iu = self.indexing[u]
                if self.res_is_ref[g] == 1:
                    Smu_tot = np.zeros(iu.size)
                    Sigi_tot = pd.w_cond_prec[u].copy()
                    if self.parents[u].size > 0:
                        pd.AK_uP_all[u] = pd.w_cond_mean_K[u].T @ pd.w_cond_prec[u]
                    if self.children[u].size > 0:
                        Sigi_tot += np.sum(pd.Sigi_children[u], axis=2)
                    Sigi_tot[np.diag_indices_from(Sigi_tot)] += self.tausq_inv_long[iu]
                    try:
                        pd.Sigi_chol[u] = _inv_trimatl(_chol_lower(_symmatu(Sigi_tot)))
                    except np.linalg.LinAlgError:
                        errtype = 10
                    if self.parents[u].size > 0:
                        Smu_tot += pd.AK_uP_all[u].T @ self.w[self.parents_indexing[u]]
                    if self.children[u].size > 0:
                        Smu_tot += np.sum(pd.Smu_children[u], axis=1)
                    Smu_tot += self.tausq_inv_long[iu] * (self.y[iu] - self.XB[iu])
                    Sigi_chol = pd.Sigi_chol[u]
                    rnvec = self.bigrnorm[iu]
                    self.w[iu] = Sigi_chol.T @ (Sigi_chol @ Smu_tot + rnvec)
                else:
                    rnvec = self.bigrnorm[iu]
                    tsq_Zt_y_XB = self.tausq_inv_long[iu] * (self.y[iu] - self.XB[iu])
                    cond_mean_K_wpar = pd.w_cond_mean_K[u] @ self.w[self.parents_indexing[u]]
                    for ix in range(iu.size):
                        tsqi = self.tausq_inv_long[iu[ix]]
                        Sigi_tot = pd.w_cond_prec_noref[u][ix] + tsqi
                        Smu_tot = pd.w_cond_prec_noref[u][ix] * cond_mean_K_wpar[ix] + tsq_Zt_y_XB[ix]
                        try:
                            pd.Sigi_chol_noref[u][ix] = _inv_trimatl(_chol_lower(_symmatu(Sigi_tot)))
                        except np.linalg.LinAlgError:
                            errtype = 11
                        Sc = pd.Sigi_chol_noref[u][ix]
                        self.w[iu[ix]] = float((Sc.T @ Sc @ Smu_tot + Sc.T * rnvec[ix])[0, 0])
                        pd.AK_uP_all[u][:, ix] = pd.w_cond_mean_K[u][ix, :] * pd.w_cond_prec_noref[u][ix][0, 0]
                if self.parents[u].size > 0:
                    if need_update:
                        pd.AK_uP_u_all[u] = pd.AK_uP_all[u] @ pd.w_cond_mean_K[u]
                    w_par = self.w[self.parents_indexing[u]]
                    for p_, up in enumerate(self.parents[u]):
                        c_ix = self.this_is_jth_child[u][p_]
                        loc, oth = self.u_is_which_col_f[up][c_ix]
                        if need_update:
                            pd.Sigi_children[up][:, :, c_ix] = pd.AK_uP_u_all[u][np.ix_(loc, loc)]
                        pd.Smu_children[up][:, c_ix] = pd.AK_uP_all[u][loc, :] @ self.w[iu] - \
                            pd.AK_uP_u_all[u][np.ix_(loc, oth)] @ w_par[oth]
        self.last_sample_errtype = errtype
        if errtype > 0:
            raise RuntimeError("Error at gibbs_sample_w")                 # Rcpp::stop (:1215-1217)

    # -- spamtree_model.cpp:1234-1358 (sampling=true)
    def predict(self, theta_update=True):
        pd = self.param_data
        self.covpars.transform(pd.theta)
        for u in self.blocks_predicting:
            iu = self.indexing[u]
            if theta_update:
                pd.Kxc[u] = self._cov(self.parents_indexing[u], iu, False)
                u_par = self.parents[u][-1]
                if pd.has_updated[u_par] == 0:
                    if self.limited_tree:
                        Kxx = self._cov(self.indexing[u_par], self.indexing[u_par], True)
                        pd.Kxx_inv[u_par] = np.linalg.inv(Kxx)
                    else:
                        u_gp = self.parents[u_par][-1]
                        invchol_block_inplace_direct(pd.Kxx_invchol[u_par], pd.Kxx_invchol[u_gp],
                                                     pd.w_cond_mean_K[u_par], pd.Rcc_invchol[u_par])
                        pd.Kxx_inv[u_par] = pd.Kxx_invchol[u_par].T @ pd.Kxx_invchol[u_par]
                pd.w_cond_mean_K[u] = pd.Kxc[u].T @ pd.Kxx_inv[u_par]
            w_par = self.w[self.parents_indexing[u]]
            for ix in range(iu.size):
                uix = iu[ix:ix + 1]
                Kcc = self._cov(uix, uix, True)
                Ktemp = Kcc - pd.w_cond_mean_K[u][ix:ix + 1, :] @ pd.Kxc[u][:, ix:ix + 1]
                try:
                    Rchol = _chol_lower(_symmatu(Ktemp))
                except np.linalg.LinAlgError:
                    Rchol = np.zeros((1, 1))
                self.w[iu[ix]] = float(pd.w_cond_mean_K[u][ix, :] @ w_par + Rchol[0, 0] * self.bigrnorm[iu[ix]])

    # -- spamtree_model.cpp:1364-1391
    def gibbs_sample_beta(self, normals_by_q):
        for j in range(self.q):
            Si_chol = _chol_lower(_symmatu(self.tausq_inv[j] * self.XtX[j] + self.Vi))
            Sc = _inv_trimatl(Si_chol)
            ia = self.ix_by_q_a[j]
            if self.reference_quirks:
                w_used = self.w[ia]                                   # Q3: subset positions index the FULL w (:1375)
            else:
                w_used = self.w[self.na_ix_all][ia]
            Xprecy_j = self.Vim + self.tausq_inv[j] * self.X_available[ia].T @ (self.y_available[ia] - w_used)
            Bmu = Sc.T @ (Sc @ Xprecy_j)
            self.Bcoeff[:, j] = Bmu + Sc.T @ np.asarray(normals_by_q[j], dtype=np.float64)
            self.XB[self.ix_by_q[j]] = self.X[self.ix_by_q[j]] @ self.Bcoeff[:, j]

    # -- spamtree_model.cpp:1393-1417; gamma_draw(j, shape, scale) supplies R::rgamma
    def gibbs_sample_tausq(self, gamma_draw):
        for j in range(self.q):
            Zw_availab = self.w[self.na_ix_all]
            XB_availab = self.XB[self.na_ix_all]
            ia = self.ix_by_q_a[j]
            yrr = self.y_available[ia] - XB_availab[ia] - Zw_availab[ia]
            bcore = float(yrr @ yrr)
            aparam = 2.01 + ia.size / 2.0
            bparam = 1.0 / (1.0 + 0.5 * bcore)
            self.tausq_inv[j] = gamma_draw(j, aparam, bparam)
            self.tausq_inv_long[self.ix_by_q[j]] = self.tausq_inv[j]
        return None

    def beta_tausq_stats(self):
        """Sufficient statistics the HIP build reduces on device (K11): X'(y-w) (Q3 pairing) and sum (y-XB-w)^2."""
        xty = np.zeros((self.p, self.q))
        ssq = np.zeros(self.q)
        for j in range(self.q):
            ia = self.ix_by_q_a[j]
            w_used = self.w[ia] if self.reference_quirks else self.w[self.na_ix_all][ia]
            xty[:, j] = self.X_available[ia].T @ (self.y_available[ia] - w_used)
            yrr = self.y_available[ia] - self.XB[self.na_ix_all][ia] - self.w[self.na_ix_all][ia]
            ssq[j] = float(yrr @ yrr)
        return xty, ssq

    def theta_update(self, data, new_param):                          # :1420-1422
        data.theta = np.asarray(new_param, dtype=np.float64).copy()

    def accept_make_change(self):                                     # :1432-1435
        self.param_data, self.alter_data = self.alter_data, self.param_data


# ----------------------------------------------------------------------------------------------------------
# mh_adapt.{h,cpp}
# ----------------------------------------------------------------------------------------------------------
def logistic(x, l=0.0, u=1.0):
    return l + (u - l) / (1.0 + np.exp(-x))


def logit(x, l=0.0, u=1.0):
    return -np.log((u - l) / (x - l) - 1.0)


def par_huvtransf_fwd(par, bounds):                                   # mh_adapt.cpp:3-8
    par = np.asarray(par, dtype=np.float64)
    return np.array([logit(par[j], bounds[j, 0], bounds[j, 1]) for j in range(par.size)])


def par_huvtransf_back(par, bounds):                                  # mh_adapt.cpp:10-15
    par = np.asarray(par, dtype=np.float64)
    return np.array([logistic(par[j], bounds[j, 0], bounds[j, 1]) for j in range(par.size)])


def unif_bounds(par, bounds):                                         # mh_adapt.h:188-202 (clamps in place)
    out = False
    for i in range(par.size):
        if par[i] < bounds[i, 0]:
            out = True
            par[i] = bounds[i, 0] + 1e-10
        if par[i] > bounds[i, 1]:
            out = True
            par[i] = bounds[i, 1] - 1e-10
    return out


def calc_jacobian(new_param, param, bounds):                          # mh_adapt.h:210-239
    def npl(x, l, u):
        return -math.log(u - x) - math.log(x - l)
    jac = 0.0
    for j in range(param.size):
        jac += npl(param[j], bounds[j, 0], bounds[j, 1]) - npl(new_param[j], bounds[j, 0], bounds[j, 1])
    return jac


def do_I_accept(logaccept, u):                                        # mh_adapt.h:20-36 (u = R::runif)
    acceptj = 1.0
    if not np.isfinite(logaccept):
        acceptj = 0.0
    elif logaccept < 0:
        acceptj = math.exp(logaccept)
    return u < acceptj


class RAMAdapt:
    """mh_adapt.h:40-135 (Vihola 2012).  Note the member g0=50 shadows the file-level g0=500."""

    def __init__(self, npars, metropolis_sd):
        self.p = npars
        self.alpha_star = 0.234
        self.gamma = 0.5 + 1e-6
        self.Ip = np.eye(npars)
        self.g0 = 50
        self.S = np.asarray(metropolis_sd, dtype=np.float64).copy()
        self.paramsd = np.linalg.cholesky(self.S)
        self.prodparam = self.paramsd / (self.g0 + 1.0)
        self.started = False
        self.propos_count = 0.0
        self.accept_count = 0.0
        self.accept_ratio = 0.0
        self.history_length = 200
        self.acceptreject_history = np.zeros(self.history_length)
        self.c = 0
        self.flag_accepted = False

    def count_proposal(self):
        self.propos_count += 1
        self.c += 1
        self.flag_accepted = False

    def count_accepted(self):
        self.accept_count += 1
        self.acceptreject_history[self.c % self.history_length] = 1
        self.flag_accepted = True

    def update_ratios(self):
        self.accept_ratio = self.accept_count / self.propos_count
        if not self.flag_accepted:
            self.acceptreject_history[self.c % self.history_length] = 0

    def adapt(self, U, alpha, mc):
        U = np.asarray(U, dtype=np.float64)
        if mc < self.g0:
            self.prodparam = self.prodparam + np.outer(U, U) / (mc + 1.0)
        else:
            if not self.started:
                self.paramsd = self.prodparam
                self.started = True
            i = mc - self.g0
            eta = min(1.0, self.p * (i + 1.0) ** (-self.gamma))
            alpha = 1.0 if math.isnan(alpha) else min(1.0, alpha)     # std::min(1.0, NaN) == 1.0
            Sigma = self.Ip + eta * (alpha - self.alpha_star) * np.outer(U, U) / float(U @ U)
            self.S = self.paramsd @ Sigma @ self.paramsd.T
            self.paramsd = np.linalg.cholesky(self.S)


# ----------------------------------------------------------------------------------------------------------
# spamtree_fit.cpp:5-430
# ----------------------------------------------------------------------------------------------------------
def spamtree_mv_mcmc(y, X, Z, coords, mv_id, blocking, gix_block, res_is_ref, parents, children, limited_tree,
                     layer_names, layer_gibbs_group, indexing, set_unif_bounds_in, start_w, theta, beta, tausq,
                     mcmcsd, mcmc_keep=100, mcmc_burn=100, mcmc_thin=1, num_threads=1, use_alg="S",
                     adapting=False, main_verbose=False, verbose=False, debug=False, printall=False,
                     sample_beta=True, sample_tausq=True, sample_theta=True, sample_w=True, sample_predicts=True,
                     seed=2021, reference_distance=False, reference_quirks=True, trace=None):
    """One chain.  ``trace`` (optional dict) receives per-iteration loglik / acceptance records."""
    rng = StRng(seed)
    bounds = np.asarray(set_unif_bounds_in, dtype=np.float64)
    n_all = np.asarray(coords).shape[0]
    q = np.asarray(Z).shape[1]
    start_w_vec = np.zeros(n_all)                                      # :95 (start_w ignored)
    mtree = SpamTreeMV(y, X, Z, coords, mv_id, blocking, gix_block, res_is_ref, parents, children, limited_tree,
                       layer_names, layer_gibbs_group, indexing, start_w_vec, beta, theta, 1.0 / tausq,
                       reference_distance=reference_distance, reference_quirks=reference_quirks)
    mtree.get_loglik_comps_w(mtree.param_data)
    mtree.get_loglik_comps_w(mtree.alter_data)
    param = mtree.param_data.theta.copy()
    predict_param = param.copy()
    current_loglik = mtree.param_data.loglik_w
    p = mtree.p
    beta_mcmc = np.zeros((p, mcmc_keep, q))
    tausq_mcmc = np.zeros((q, mcmc_keep))
    theta_mcmc = np.zeros((param.size, mcmc_keep))
    w_mcmc: List[Optional[np.ndarray]] = [None] * mcmc_keep
    yhat_mcmc: List[Optional[np.ndarray]] = [None] * mcmc_keep
    mcmc = mcmc_thin * mcmc_keep + mcmc_burn
    msaved = 0
    adaptivemc = RAMAdapt(param.size, np.asarray(mcmcsd, dtype=np.float64))
    records = []
    for m in range(mcmc):
        mtree.predicting = False
        mx = m - mcmc_burn
        if mx >= 0 and mx % mcmc_thin == 0:
            mtree.predicting = True
        if sample_w:
            mtree.gibbs_sample_w(rng.sweep_normals(m, n_all), True)
            mtree.get_loglik_w(mtree.param_data)
            current_loglik = mtree.param_data.loglik_w
        rec = {"loglik_after_w": current_loglik}
        if sample_theta:
            adaptivemc.count_proposal()
            U_update = rng.theta_normals(m, param.size)
            new_param = par_huvtransf_back(par_huvtransf_fwd(param, bounds) + adaptivemc.paramsd @ U_update, bounds)
            out_unif_bounds = unif_bounds(new_param, bounds)
            mtree.theta_update(mtree.alter_data, new_param)
            acceptable = mtree.get_loglik_comps_w(mtree.alter_data)
            new_loglik = mtree.alter_data.loglik_w
            current_loglik = mtree.param_data.loglik_w
            if np.isnan(current_loglik):
                raise FloatingPointError("At nan loglik: error.")          # throw 1 (:234-237)
            jacobian = calc_jacobian(new_param, param, bounds)
            logaccept = new_loglik - current_loglik + jacobian
            accepted = bool(do_I_accept(logaccept, rng.mh_uniform(m))) and bool(acceptable)
            rec.update(new_loglik=new_loglik, logaccept=logaccept, accepted=accepted, acceptable=acceptable,
                       proposal=new_param.copy(), out_of_bounds=out_unif_bounds)
            if accepted:
                adaptivemc.count_accepted()
                current_loglik = new_loglik
                mtree.accept_make_change()
                param = new_param
            adaptivemc.update_ratios()
            if adapting:
                with np.errstate(all="ignore"):                              # IEEE semantics of acceptable*exp(.)
                    alpha_in = np.float64(1.0 if acceptable else 0.0) * np.exp(np.float64(logaccept))
                adaptivemc.adapt(U_update, float(alpha_in), m)
        need_update = bool(np.sum(np.abs(param - predict_param) > 1e-05))
        if mtree.predicting and sample_predicts and sample_w:
            mtree.predict(need_update)
            predict_param = param.copy()
        if sample_tausq:
            mtree.gibbs_sample_tausq(lambda j, a, b: rng.gamma(m, j, a, b))
        if sample_beta:
            mtree.gibbs_sample_beta([rng.beta_normals(m, j, p) for j in range(q)])
        records.append(rec)
        if mx >= 0 and mx % mcmc_thin == 0:
            tausq_mcmc[:, msaved] = 1.0 / mtree.tausq_inv
            beta_mcmc[:, msaved, :] = mtree.Bcoeff
            theta_mcmc[:, msaved] = mtree.param_data.theta
            w_mcmc[msaved] = mtree.w.copy()
            yhat_mcmc[msaved] = mtree.XB + mtree.w + mtree.tausq_inv_long ** (-0.5) * rng.yhat_normals(m, n_all)
            msaved += 1
    if trace is not None:
        trace["records"] = records
        trace["model"] = mtree
    return dict(w_mcmc=w_mcmc, yhat_mcmc=yhat_mcmc, beta_mcmc=beta_mcmc, tausq_mcmc=tausq_mcmc,
                theta_mcmc=theta_mcmc, paramsd=adaptivemc.paramsd, block_ct_obs=mtree.block_ct_obs,
                indexing=mtree.indexing, parents_indexing=mtree.parents_indexing)
