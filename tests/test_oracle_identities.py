"""Pins the NumPy oracle with dense brute-force identities -- the reference has no tests.

Everything here is CPU-only.  "Dense" quantities are built from the full covariance matrix with generic
linear algebra, never from the oracle's per-block caches.
"""
import math

import numpy as np
import pytest

from oracle import spamtree_oracle as so
from tests.util import make_problem, nice_theta, oracle_model


def dense_cov(pb, theta, rows=None):
    cp = so.CovarianceParams(2, pb["q"], -1)
    cp.transform(theta)
    rows = np.arange(pb["n"]) if rows is None else rows
    return so.Covariancef(pb["coords"], pb["mv_id"] - 1, rows, rows, cp, True)


def dense_precision(pb, theta):
    """Q = sum_u (I_u - H_u E_pa)' R_u^{-1} (I_u - H_u E_pa), R_u diagonal on non-reference levels."""
    n = pb["n"]
    K = dense_cov(pb, theta)
    Q = np.zeros((n, n))
    logdet = 0.0
    labels = np.unique(pb["block_groups"])
    for u in range(len(pb["indexing"])):
        iu = pb["indexing"][u]
        pa = pb["parents"][u]
        g = int(np.nonzero(labels == pb["block_groups"][u])[0][0])
        B = np.zeros((iu.size, n))
        B[np.arange(iu.size), iu] = 1.0
        if pa.size:
            pi = np.concatenate([pb["indexing"][a] for a in pa])
            H = np.linalg.solve(K[np.ix_(pi, pi)], K[np.ix_(pi, iu)]).T
            B[:, pi] -= H
            R = K[np.ix_(iu, iu)] - H @ K[np.ix_(pi, iu)]
        else:
            R = K[np.ix_(iu, iu)]
        if pb["res_is_ref"][g] == 0:
            R = np.diag(np.diag(R))
        Q += B.T @ np.linalg.solve(R, B)
        logdet += -np.linalg.slogdet(R)[1]
    return Q, logdet


def test_philox_known_answers():
    """Random123 kat_vectors for philox4x32-10."""
    kat = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
           ((0xffffffff,) * 4, (0xffffffff,) * 2, (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
           ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
            (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, exp in kat:
        out = so.philox4x32_10(*ctr, *key)
        assert tuple(int(x) for x in out) == exp


def test_normals_are_standard():
    z = so.StRng(2021).sweep_normals(3, 200000)
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1) < 0.01
    assert abs(np.mean(z ** 3)) < 0.03 and abs(np.mean(z ** 4) - 3) < 0.06
    g = np.array([so.StRng(7).gamma(it, 0, 50.5, 0.02) for it in range(3000)])
    assert abs(g.mean() - 50.5 * 0.02) < 0.01 and abs(g.var() - 50.5 * 0.02 ** 2) < 0.003


def test_cross_covariance_ag10_mpmath():
    """man/CrossCovarianceAG10.Rd:72-93 inputs (q=2) and a q=3 parameter set against 50-digit arithmetic."""
    import mpmath as mp
    mp.mp.dps = 50
    SS = 10
    xl = np.linspace(0.0, 1.0, SS)
    g = np.array([(a, b) for b in xl for a in xl])            # expand.grid: first factor fastest
    cx = np.vstack([g, g])
    mv = np.repeat([1, 2], SS * SS)
    ai1, ai2, phi_i, thetamv = [1, 1.5], [.1, .51], [1, 2], [5]
    D = np.array([[0, 1.0], [1.0, 0]])
    CC = so.CrossCovarianceAG10(cx, mv, cx, mv, ai1, ai2, phi_i, thetamv, D)
    assert CC.shape == (200, 200)
    rng = np.random.default_rng(1)
    for _ in range(200):
        i, j = rng.integers(0, 200, 2)
        h = mp.sqrt((mp.mpf(cx[i, 0]) - mp.mpf(cx[j, 0])) ** 2 + (mp.mpf(cx[i, 1]) - mp.mpf(cx[j, 1])) ** 2)
        vi, vj = mv[i] - 1, mv[j] - 1
        v = mp.mpf(D[vi, vj])
        if v == 0:
            ref = mp.mpf(ai1[vi]) ** 2 * mp.e ** (-5 * h) + mp.mpf(ai2[vi]) ** 2 * mp.e ** (-mp.mpf(phi_i[vi]) * h)
        else:
            ref = mp.mpf(ai1[vi]) * mp.mpf(ai1[vj]) * mp.e ** (-5 * h / mp.sqrt(v + 1)) / (v + 1)
        assert abs(CC[i, j] - float(ref)) <= 1e-14 * max(1.0, abs(float(ref)))
    # q = 3: psi = (a v + 1)^beta, C = exp(-c h / sqrt(psi)) / psi
    th = nice_theta(3)
    cp = so.CovarianceParams(2, 3, -1)
    cp.transform(th)
    pts = rng.uniform(size=(30, 2))
    mv3 = rng.integers(1, 4, 30)
    C3 = so.CrossCovarianceAG10(pts, mv3, pts, mv3, cp.ai1, cp.ai2, cp.phi_i, cp.thetamv, cp.Dmat)
    a, be, c = (mp.mpf(float(t)) for t in cp.thetamv)
    for i in range(30):
        for j in range(30):
            h = mp.sqrt((mp.mpf(pts[i, 0]) - mp.mpf(pts[j, 0])) ** 2 + (mp.mpf(pts[i, 1]) - mp.mpf(pts[j, 1])) ** 2)
            vi, vj = mv3[i] - 1, mv3[j] - 1
            v = mp.mpf(float(cp.Dmat[vi, vj]))
            if v == 0:
                ref = mp.mpf(float(cp.ai1[vi])) ** 2 * mp.e ** (-c * h) + \
                    mp.mpf(float(cp.ai2[vi])) ** 2 * mp.e ** (-mp.mpf(float(cp.phi_i[vi])) * h)
            else:
                psi = (a * v + 1) ** be
                ref = mp.mpf(float(cp.ai1[vi])) * mp.mpf(float(cp.ai1[vj])) * mp.e ** (-c * h / mp.sqrt(psi)) / psi
            assert abs(C3[i, j] - float(ref)) <= 2e-14 * max(1.0, abs(float(ref)))


def test_reference_distance_switch_bounds_q1_effect():
    """Q1: the cancellation form differs from the direct distance by <= ~3e-8 absolute in h."""
    rng = np.random.default_rng(0)
    x = rng.uniform(size=(300, 2))
    a = so.cexpcov(x, x, 1.0, 1.0, True, reference_distance=True)
    b = so.cexpcov(x, x, 1.0, 1.0, True, reference_distance=False)
    dh = np.abs(np.log(a) - np.log(b))
    assert dh.max() < 1e-7
    off = ~np.eye(300, dtype=bool)
    hmin = (-np.log(b[off])).min()
    assert dh[off].max() < 1e-15 / hmin + 1e-12                           # |d(h)| ~ ulp(|x|^2) / h off the diagonal


def test_one_level_tree_is_exact_gp():
    pb = make_problem(side=5, q=1, seed=3)
    assert len(pb["indexing"]) == 1
    rng = np.random.default_rng(0)
    w = rng.standard_normal(pb["n"])
    m = oracle_model(pb, w=w)
    assert m.get_loglik_comps_w(m.param_data)
    K = dense_cov(pb, pb["theta"])
    exact = -0.5 * pb["n"] * math.log(2 * math.pi) - 0.5 * np.linalg.slogdet(K)[1] - 0.5 * w @ np.linalg.solve(K, w)
    assert abs(m.param_data.loglik_w - exact) < 1e-9 * abs(exact)


@pytest.mark.parametrize("q,side,missing", [(1, 25, 0.0), (2, 16, 0.0), (3, 12, 0.0), (1, 24, 0.15)])
def test_loglik_equals_dense_dag_density(q, side, missing):
    pb = make_problem(side=side, q=q, seed=5, missing=missing)
    rng = np.random.default_rng(1)
    w = rng.standard_normal(pb["n"])
    m = oracle_model(pb, w=w)
    assert m.get_loglik_comps_w(m.param_data)
    obs_blocks = [u for u in range(m.n_blocks) if m.block_ct_obs[u] > 0]
    rows = np.sort(np.concatenate([pb["indexing"][u] for u in obs_blocks]))
    Q, logdet = dense_precision(pb, pb["theta"])
    # prediction blocks are not part of the density the sampler targets
    pred_rows = np.setdiff1d(np.arange(pb["n"]), rows)
    if pred_rows.size:
        keep = np.ones(pb["n"], dtype=bool)
        keep[pred_rows] = False
        n = pb["n"]
        K = dense_cov(pb, pb["theta"])
        Q = np.zeros((n, n)); logdet = 0.0
        labels = np.unique(pb["block_groups"])
        for u in obs_blocks:
            iu = pb["indexing"][u]; pa = pb["parents"][u]
            g = int(np.nonzero(labels == pb["block_groups"][u])[0][0])
            B = np.zeros((iu.size, n)); B[np.arange(iu.size), iu] = 1.0
            if pa.size:
                pi = np.concatenate([pb["indexing"][a] for a in pa])
                H = np.linalg.solve(K[np.ix_(pi, pi)], K[np.ix_(pi, iu)]).T
                B[:, pi] -= H
                R = K[np.ix_(iu, iu)] - H @ K[np.ix_(pi, iu)]
            else:
                R = K[np.ix_(iu, iu)]
            if pb["res_is_ref"][g] == 0:
                R = np.diag(np.diag(R))
            Q += B.T @ np.linalg.solve(R, B); logdet += -np.linalg.slogdet(R)[1]
    exact = -0.5 * rows.size * math.log(2 * math.pi) + 0.5 * logdet - 0.5 * w @ Q @ w
    assert abs(m.param_data.loglik_w - exact) < 1e-8 * abs(exact)
    # the cached-H evaluation (phase C) agrees with the factorisation pass (phase A)
    ll_a = m.param_data.loglik_w
    m.get_loglik_w(m.param_data)
    assert abs(m.param_data.loglik_w - ll_a) < 1e-10 * abs(ll_a)


def test_invchol_extension_is_inverse_cholesky():
    pb = make_problem(side=25, q=1, seed=2)
    m = oracle_model(pb)
    assert m.get_loglik_comps_w(m.param_data)
    K = dense_cov(pb, pb["theta"])
    checked = 0
    for u in range(m.n_blocks):
        if m.children[u].size > 0 and m.parents[u].size > 0:
            rows = np.concatenate([m.parents_indexing[u], m.indexing[u]])
            L = np.linalg.cholesky(K[np.ix_(rows, rows)])
            assert np.abs(m.param_data.Kxx_invchol[u] @ L - np.eye(rows.size)).max() < 1e-8
            checked += 1
    assert checked >= 4


def test_block_draw_is_exact_full_conditional():
    """Precision of every block's draw, and the mean wherever messages cannot be stale (Q2)."""
    pb = make_problem(side=25, q=1, seed=7, last_not_reference=False)
    rng = np.random.default_rng(3)
    w0 = rng.standard_normal(pb["n"])
    tausq = 0.2
    beta = np.array([0.3, -0.2, 0.1])
    m = oracle_model(pb, w=w0, tausq=tausq, beta=beta)
    assert m.get_loglik_comps_w(m.param_data)
    Q, _ = dense_precision(pb, pb["theta"])
    Qpost = Q + np.eye(pb["n"]) / tausq
    b = (pb["y"] - pb["X"] @ beta) / tausq
    z = np.zeros(pb["n"])
    w_before = m.w.copy()
    m.gibbs_sample_w(z)
    levels = np.unique(pb["block_groups"])
    w_running = w_before.copy()
    for g in range(levels.size - 1, -1, -1):
        for u in m.u_by_block_groups[g]:
            iu = pb["indexing"][u]
            Sc = m.param_data.Sigi_chol[u]
            assert np.abs(Sc.T @ Sc @ Qpost[np.ix_(iu, iu)] - np.eye(iu.size)).max() < 1e-8
            if g >= levels.size - 2:       # leaves and their parents: no stale intermediate ancestors
                rest = np.setdiff1d(np.arange(pb["n"]), iu)
                mean = np.linalg.solve(Qpost[np.ix_(iu, iu)], b[iu] - Qpost[np.ix_(iu, rest)] @ w_running[rest])
                assert np.abs(m.w[iu] - mean).max() < 1e-8 * max(1.0, np.abs(mean).max())
        for u in m.u_by_block_groups[g]:
            w_running[pb["indexing"][u]] = m.w[pb["indexing"][u]]


def test_chol_failure_protocol():
    pb = make_problem(side=25, q=1, seed=2)
    th = pb["theta"].copy()
    th[0] = -1.0                                            # negative sigma^2 -> not PD at the root
    m = oracle_model(pb, theta=th)
    assert m.get_loglik_comps_w(m.param_data) is False
    assert m.last_errtype == 1


def test_short_chain_runs_and_is_deterministic():
    pb = make_problem(side=20, q=1, seed=11, missing=0.1)
    kw = dict(mcmc_keep=3, mcmc_burn=4, mcmc_thin=1, adapting=True, seed=99)
    args = (pb["y"], pb["X"], pb["Z"], pb["coords"], pb["mv_id"], pb["blocking"], pb["gix_block"], pb["res_is_ref"],
            pb["parents"], pb["children"], False, pb["block_names"], pb["block_groups"], pb["indexing"],
            pb["bounds"], np.zeros((pb["n"], 1)), pb["theta"], np.zeros(pb["p"]), 0.1, 0.01 * np.eye(4))
    r1 = so.spamtree_mv_mcmc(*args, **kw)
    r2 = so.spamtree_mv_mcmc(*args, **kw)
    assert np.array_equal(r1["theta_mcmc"], r2["theta_mcmc"])
    assert np.array_equal(r1["w_mcmc"][-1], r2["w_mcmc"][-1])
    assert np.all(np.isfinite(r1["beta_mcmc"])) and np.all(r1["tausq_mcmc"] > 0)
    assert np.all(np.isfinite(r1["yhat_mcmc"][-1]))


def test_list_qtile_restatement_against_numpy_order_statistics():
    """oracle/list_summaries.py (list_mean.cpp:62-137): the rule is an interpolation between two ADJACENT order statistics
    around q * n (integer truncation included: not MATLAB's prctile); at the extremes the minimum / maximum."""
    from oracle.list_summaries import list_mean, list_qtile, prctile_stl
    rng = np.random.default_rng(0)
    for n in (1, 2, 5, 40, 101):
        v = rng.standard_normal(n)
        s = np.sort(v)
        for q in (0.0, 0.025, 0.3, 0.5, 0.75, 0.975, 1.0):
            got = prctile_stl(v, q * 100.0)
            assert s[0] - 1e-12 <= got <= s[-1] + 1e-12
            j = np.searchsorted(s, got, side="left")
            assert any(abs(got - s[k]) < 1e-12 for k in range(n)) or (s[j - 1] <= got <= s[j])   # between neighbours
        if n == 5:      # worked by hand from the source: r = 2.5, idx_lo = int(1.5) = 1, k = 3, weight (0.5 - (-0.5)) on the
            assert prctile_stl(v, 50.0) == s[1]          # lower one: the reference's "median" of five is the SECOND smallest
        assert prctile_stl(v, 0.0) == s[0] and prctile_stl(v, 100.0) == s[-1]
    x = [rng.standard_normal((6, 1)) for _ in range(9)]
    assert np.allclose(list_mean(x), sum(x) / 9)
    assert list_qtile(x, 0.5).shape == (6, 1)
