"""limited_tree = TRUE (/root/reference/src/tree_dep.cpp:133-186, /root/reference/src/spamtree_model.cpp:901-903,
1275-1278): every block has ONE parent, Kxx_inv(u) = inv_sympd(K_uu).

CPU part: the edge builder's contract and the oracle's limited branch against dense brute force (the OpenMP restatement
oracle/refcpu reproduces the limited golden fixture in tests/test_golden.py).
GPU part: the HIP path (st_options.reserved bit 1: marginal chain factors, k_marginal_invchol) against the oracle.
"""
import math

import numpy as np
import pytest

from tests.test_oracle_identities import dense_precision
from tests.util import make_problem, oracle_model

REL = 1e-9


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(1e-300, np.abs(b).max()))


def test_make_edges_limited_contract():
    """One parent = the last entry of the full tree's ancestor list; children = the non-empty blocks one level down."""
    full = make_problem(side=25, q=1, seed=4, missing=0.1)
    lim = make_problem(side=25, q=1, seed=4, missing=0.1, limited_tree=True)
    assert np.array_equal(full["blocking"], lim["blocking"])
    nb = len(full["parents"])
    lev = np.asarray(full["block_groups"])
    for u in range(nb):
        pf, pl = full["parents"][u], lim["parents"][u]
        assert pl.size == min(1, pf.size)
        if pf.size:
            assert pl[0] == pf[-1]
        cf, cl = full["children"][u], lim["children"][u]
        assert np.all(np.isin(cl, cf))
        nxt = np.unique(lev[lev > lev[u]]).min() if np.any(lev > lev[u]) else None
        expect = cf[lev[cf] == nxt] if cf.size else cf
        assert np.array_equal(np.sort(cl), np.sort(expect))


@pytest.mark.parametrize("q,side,missing", [(1, 25, 0.0), (2, 16, 0.0), (1, 24, 0.15)])
def test_limited_loglik_equals_dense_dag_density(q, side, missing):
    pb = make_problem(side=side, q=q, seed=5, missing=missing, limited_tree=True)
    rng = np.random.default_rng(1)
    w = rng.standard_normal(pb["n"])
    m = oracle_model(pb, w=w)
    assert m.limited_tree and m.get_loglik_comps_w(m.param_data)
    obs_blocks = [u for u in range(m.n_blocks) if m.block_ct_obs[u] > 0]
    rows = np.sort(np.concatenate([pb["indexing"][u] for u in obs_blocks]))
    sub = dict(pb)
    if rows.size < pb["n"]:                                   # prediction blocks are not part of the density
        sub["indexing"] = [ix if m.block_ct_obs[u] > 0 else ix[:0] for u, ix in enumerate(pb["indexing"])]
    Q, logdet = dense_precision(sub, pb["theta"])
    exact = -0.5 * rows.size * math.log(2 * math.pi) + 0.5 * logdet - 0.5 * w @ Q @ w
    assert abs(m.param_data.loglik_w - exact) < 1e-8 * abs(exact)
    ll_a = m.param_data.loglik_w
    m.get_loglik_w(m.param_data)
    assert abs(m.param_data.loglik_w - ll_a) < 1e-10 * abs(ll_a)


def test_limited_block_draw_is_exact_full_conditional():
    """With single parents no message can be stale (Q2 does not arise): every block's draw is the exact full conditional."""
    pb = make_problem(side=25, q=1, seed=7, last_not_reference=False, limited_tree=True)
    rng = np.random.default_rng(3)
    w0 = rng.standard_normal(pb["n"])
    tausq = 0.2
    beta = np.array([0.3, -0.2, 0.1])
    m = oracle_model(pb, w=w0, tausq=tausq, beta=beta)
    assert m.get_loglik_comps_w(m.param_data)
    Q, _ = dense_precision(pb, pb["theta"])
    Qpost = Q + np.eye(pb["n"]) / tausq
    b = (pb["y"] - pb["X"] @ beta) / tausq
    w_running = m.w.copy()
    m.gibbs_sample_w(np.zeros(pb["n"]))
    levels = np.unique(pb["block_groups"])
    for g in range(levels.size - 1, -1, -1):
        for u in m.u_by_block_groups[g]:
            iu = pb["indexing"][u]
            Sc = m.param_data.Sigi_chol[u]
            assert np.abs(Sc.T @ Sc @ Qpost[np.ix_(iu, iu)] - np.eye(iu.size)).max() < 1e-8
            rest = np.setdiff1d(np.arange(pb["n"]), iu)
            mean = np.linalg.solve(Qpost[np.ix_(iu, iu)], b[iu] - Qpost[np.ix_(iu, rest)] @ w_running[rest])
            assert np.abs(m.w[iu] - mean).max() < 1e-8 * max(1.0, np.abs(mean).max())
        for u in m.u_by_block_groups[g]:
            w_running[pb["indexing"][u]] = m.w[pb["indexing"][u]]


# ---------------------------------------------------------------------------------------------------------------
LIMITED_CASES = [dict(side=25, q=1), dict(side=40, q=1, missing=0.1), dict(side=16, q=2, missing=0.1),
                 dict(side=30, q=1, random_coords=True, missing=0.05), dict(side=12, q=3)]


def _hip(pb, **kw):
    from tests.test_gpu_parity import hip_model
    return hip_model(pb, **kw)


@pytest.mark.gpu
@pytest.mark.parametrize("force_generic", [False, True])
@pytest.mark.parametrize("case", LIMITED_CASES)
def test_limited_tree_parity(case, force_generic):
    """Phases A, B, C and P of the HIP path on make_edges_limited's tree against the oracle's limited branch."""
    pb = make_problem(seed=21, limited_tree=True, **case)
    rng = np.random.default_rng(5)
    w0 = rng.standard_normal(pb["n"])
    om = oracle_model(pb, w=w0, tausq=0.2)
    hm = _hip(pb, w=w0, tausq=0.2, force_generic=force_generic)
    assert om.get_loglik_comps_w(om.param_data) and hm.get_loglik_comps_w(0)
    assert abs(hm.loglik_w[0] - om.param_data.loglik_w) <= REL * abs(om.param_data.loglik_w)
    ld, ll = hm.comps(0)
    assert relerr(ld, om.param_data.logdetCi_comps) <= REL and relerr(ll, om.param_data.loglik_w_comps) <= REL
    for u in range(om.n_blocks):
        if om.block_ct_obs[u] == 0:
            continue
        H, Ri = hm.block(0, u)
        if om.parents[u].size:
            assert relerr(H, om.param_data.w_cond_mean_K[u]) <= 1e-8, u
        ref_ri = om.param_data.Rcc_invchol[u] if om.block_is_reference[u] else om.param_data.ccholprecdiag[u]
        assert relerr(Ri, ref_ri) <= REL, u
    for _ in range(3):
        z = rng.standard_normal(pb["n"])
        om.gibbs_sample_w(z); hm.deal_with_w(z)
        assert relerr(hm.get_w()[om.na_ix_all], om.w[om.na_ix_all]) <= REL
        om.get_loglik_w(om.param_data)
        assert abs(hm.get_loglik_w(0) - om.param_data.loglik_w) <= REL * abs(om.param_data.loglik_w)
    if len(om.blocks_predicting) > 0:
        om.predict(True); hm.predict(True)
        assert relerr(hm.get_w(), om.w) <= REL
    hm.close()


@pytest.mark.gpu
def test_limited_tree_quad_kernel(monkeypatch):
    """k_factor_quad on the limited tree (SPAMTREE_QUAD_MIN=1 makes these small levels eligible): siblings share their
    single parent's marginal factor; sibling leaf groups of different parents share nothing."""
    monkeypatch.setenv("SPAMTREE_QUAD_MIN", "1")
    monkeypatch.setenv("SPAMTREE_QUAD_UNITS", "4")
    test_limited_tree_parity(LIMITED_CASES[1], False)


@pytest.mark.gpu
def test_limited_tree_rejects_full_parent_lists_and_vice_versa():
    from spamtree_amd.model import SpamTreeError
    full = make_problem(side=25, q=1, seed=1)
    lim = make_problem(side=25, q=1, seed=1, limited_tree=True)
    bad1 = dict(full); bad1["limited_tree"] = True          # ancestor lists with the limited bit set
    bad2 = dict(lim); bad2["limited_tree"] = False          # single parents without it
    for bad in (bad1, bad2):
        with pytest.raises(SpamTreeError):
            _hip(bad)
