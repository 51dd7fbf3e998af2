"""GPU: the HIP path against (a) the reference's OWN distance formula at the stated tolerance, (b) dense linear algebra that
never touches the oracle's per-block code, (c) the reference's failure protocol (errtype 2, 3, 10, 11).

(a) /root/reference/src/covariance_functions.cpp:98-108 forms distances as |x|^2 + |y|^2 - 2 x.y; the HIP path
    computes sqrt(dx^2 + dy^2).  BASELINE.md states the parity tolerance against that formula: 1e-6 * max(1, phi / 30)
    relative.  The oracle evaluates the cancellation form in two flavours: plain double arithmetic (R's reference BLAS) and
    an emulated FMA BLAS (OpenBLAS / MKL), whose self-distances are non-zero for ~17 % of random points (up to 2.1e-8).
(b) exact GP on a one-level tree, log N(w; 0, Q^-1) with the dense DAG precision, exact Gaussian full conditionals: the
    same identities tests/test_oracle_identities.py runs against the oracle, here run directly against the HIP library.
(c) /root/reference/src/spamtree_model.cpp:919, 958, 971-982 (phase A, early return per level: Q5), :1056, 1135 (sweep).
"""
import math

import numpy as np
import pytest

from tests.test_gpu_parity import hip_model, relerr
from tests.test_oracle_identities import dense_cov, dense_precision
from tests.util import default_bounds, make_problem, nice_theta, oracle_model

pytestmark = pytest.mark.gpu

MH_START = np.full(4, 0.5 * (1e-3 + 1e3))          # midpoint of [1e-3, 1e3]: /root/reference/R/spamtree_fit.R:138


def stated_tol(phi):
    return 1e-6 * max(1.0, phi / 30.0)              # BASELINE.md, "Parity"


# ---------------------------------------------------------------------------------------------------------------------
# (a) the reference's distance formula
# ---------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("flavour", [True, "fma"])
@pytest.mark.parametrize("theta", [MH_START, nice_theta(1)], ids=["mh_start_phi500", "phi6"])
@pytest.mark.parametrize("random_coords", [False, True], ids=["grid", "random"])
def test_factors_and_draws_vs_reference_distance_formula(flavour, theta, random_coords):
    pb = make_problem(side=25, q=1, seed=11, missing=0.1, random_coords=random_coords)
    rng = np.random.default_rng(5)
    w0 = rng.standard_normal(pb["n"])
    om = oracle_model(pb, theta=theta, w=w0, tausq=0.2, reference_distance=flavour)
    hm = hip_model(pb, theta=theta, w=w0, tausq=0.2)
    assert om.get_loglik_comps_w(om.param_data) and hm.get_loglik_comps_w(0)
    tol = stated_tol(theta[3])
    if flavour == "fma":
        # An FMA BLAS moves the REFERENCE's own K_ii by up to phi * 2.2e-8 relative (self-distance 2.2e-8 instead of 0); a
        # conditional variance r_i = K_ii - H_i K_xi inherits that absolute change, i.e. a relative one amplified by
        # K_ii / r_i.  That factor is a property of the problem, not of either implementation: it scales the stated bound.
        cvar = np.concatenate([1.0 / om.param_data.ccholprecdiag[u] ** 2 for u in range(om.n_blocks)
                               if om.block_ct_obs[u] > 0 and not om.block_is_reference[u]])
        tol = max(tol, theta[3] * 2.2e-8 * theta[0] / cvar.min())
    assert abs(hm.loglik_w[0] - om.param_data.loglik_w) <= tol * abs(om.param_data.loglik_w)
    ld, ll = hm.comps(0)
    assert relerr(ld, om.param_data.logdetCi_comps) <= tol and relerr(ll, om.param_data.loglik_w_comps) <= tol
    for u in range(om.n_blocks):
        if om.block_ct_obs[u] == 0:
            continue
        H, Ri = hm.block(0, u)
        ref_ri = om.param_data.Rcc_invchol[u] if om.block_is_reference[u] else om.param_data.ccholprecdiag[u]
        assert relerr(Ri, ref_ri) <= tol, u
        if om.parents[u].size:
            assert relerr(H, om.param_data.w_cond_mean_K[u]) <= 10 * tol, u
    for it in range(3):
        z = rng.standard_normal(pb["n"])
        om.gibbs_sample_w(z)
        hm.deal_with_w(z)
    assert relerr(hm.get_w()[om.na_ix_all], om.w[om.na_ix_all]) <= tol
    om.get_loglik_w(om.param_data)
    assert abs(hm.get_loglik_w(0) - om.param_data.loglik_w) <= tol * abs(om.param_data.loglik_w)
    hm.close()


@pytest.mark.parametrize("flavour,random_coords", [(True, False), (True, True), ("fma", True)])
def test_chain_from_mh_start_vs_reference_distance_formula(flavour, random_coords):
    """60 adaptive iterations from the reference's start value theta = midpoint of the bounds (phi ~ 500, where the
    cancellation matters most), C++ host driver on the GPU against the oracle chain that uses the reference's formula."""
    from oracle import spamtree_oracle as so
    from spamtree_amd import fit
    from tests.test_gpu_chain import args_of
    pb = make_problem(side=25, q=1, seed=11, missing=0.1, random_coords=random_coords)
    a = list(args_of(pb, 4))
    a[16] = MH_START
    kw = dict(mcmc_keep=4, mcmc_burn=56, mcmc_thin=1, adapting=True, seed=99, main_verbose=False)
    ref = so.spamtree_mv_mcmc(*a, reference_distance=flavour, **kw)
    got = fit.spamtree_mv_mcmc(*a, **kw)
    assert "None" not in got
    phi_max = float(np.max(ref["theta_mcmc"][3]))
    tol = stated_tol(max(phi_max, MH_START[3]))
    assert relerr(got["theta_mcmc"], ref["theta_mcmc"]) <= tol
    assert relerr(got["tausq_mcmc"], ref["tausq_mcmc"]) <= tol
    assert relerr(got["beta_mcmc"], ref["beta_mcmc"]) <= tol
    for i in range(4):
        assert relerr(np.asarray(got["w_mcmc"][i]).reshape(-1), ref["w_mcmc"][i]) <= tol


# ---------------------------------------------------------------------------------------------------------------------
# (b) dense identities, no oracle involved
# ---------------------------------------------------------------------------------------------------------------------
def test_hip_one_level_tree_is_exact_gp():
    pb = make_problem(side=5, q=1, seed=3)
    assert len(pb["indexing"]) == 1
    w = np.random.default_rng(0).standard_normal(pb["n"])
    hm = hip_model(pb, w=w)
    assert hm.get_loglik_comps_w(0)
    K = dense_cov(pb, pb["theta"])
    exact = -0.5 * pb["n"] * math.log(2 * math.pi) - 0.5 * np.linalg.slogdet(K)[1] - 0.5 * w @ np.linalg.solve(K, w)
    assert abs(hm.loglik_w[0] - exact) < 1e-9 * abs(exact)
    hm.close()


@pytest.mark.parametrize("q,side,kw", [(1, 25, {}), (2, 16, {}), (3, 12, {}), (1, 30, dict(cell_size=9, K=(3, 2))),
                                       (3, 18, dict(cell_size=9))])
@pytest.mark.parametrize("generic", [False, True])
def test_hip_loglik_equals_dense_dag_density(q, side, kw, generic):
    """log p(w | theta) of phases A and C = log N(w; 0, Q^-1) with Q assembled from the full covariance matrix by generic
    dense algebra (tests/test_oracle_identities.py::dense_precision: np.linalg.solve on K, nothing block-cached)."""
    pb = make_problem(side=side, q=q, seed=5, **kw)
    w = np.random.default_rng(1).standard_normal(pb["n"])
    hm = hip_model(pb, w=w, force_generic=generic)
    assert hm.get_loglik_comps_w(0)
    Q, logdet = dense_precision(pb, pb["theta"])
    exact = -0.5 * pb["n"] * math.log(2 * math.pi) + 0.5 * logdet - 0.5 * w @ Q @ w
    assert abs(hm.loglik_w[0] - exact) < 1e-8 * abs(exact)
    assert abs(hm.get_loglik_w(0) - exact) < 1e-8 * abs(exact)
    hm.close()


@pytest.mark.parametrize("last_not_reference", [False, True])
def test_hip_block_draw_is_exact_full_conditional(last_not_reference):
    """With z = 0 a block's draw is its conditional mean; the deepest level is sampled first, so its response to z is its
    conditional Cholesky factor.  Both against the dense posterior precision Q + I / tausq (no oracle)."""
    pb = make_problem(side=25, q=1, seed=7, last_not_reference=last_not_reference)
    rng = np.random.default_rng(3)
    w0 = rng.standard_normal(pb["n"])
    tausq = 0.2
    beta = np.array([0.3, -0.2, 0.1])
    hm = hip_model(pb, w=w0, tausq=tausq, beta=beta)
    assert hm.get_loglik_comps_w(0)
    n = pb["n"]
    Q, _ = dense_precision(pb, pb["theta"])
    Qpost = Q + np.eye(n) / tausq
    b = (pb["y"] - pb["X"] @ beta) / tausq
    labels = np.unique(pb["block_groups"])
    by_level = [[u for u in range(len(pb["indexing"])) if pb["block_groups"][u] == g] for g in labels]
    # ---- means: leaves and their parents have no stale intermediate ancestors in their messages
    hm.deal_with_w(np.zeros(n))
    w1 = hm.get_w()
    w_running = w0.copy()
    for gi in range(labels.size - 1, labels.size - 3, -1):
        for u in by_level[gi]:
            iu = pb["indexing"][u]
            rest = np.setdiff1d(np.arange(n), iu)
            # (non-reference rows are conditionally independent given the ancestors: Qpost[iu, iu] is diagonal there)
            mean = np.linalg.solve(Qpost[np.ix_(iu, iu)], b[iu] - Qpost[np.ix_(iu, rest)] @ w_running[rest])
            assert np.abs(w1[iu] - mean).max() < 1e-8 * max(1.0, np.abs(mean).max()), (gi, u)
        for u in by_level[gi]:
            w_running[pb["indexing"][u]] = w1[pb["indexing"][u]]
    # ---- conditional covariance of the deepest level: w(z) - w(0) = Lc' z_u with Lc'Lc = Qpost[u,u]^-1
    deepest = by_level[-1]
    mmax = max(pb["indexing"][u].size for u in deepest)
    resp = {u: np.zeros((pb["indexing"][u].size, pb["indexing"][u].size)) for u in deepest}
    for k in range(mmax):
        z = np.zeros(n)
        for u in deepest:
            iu = pb["indexing"][u]
            if k < iu.size:
                z[iu[k]] = 1.0
        hm.set_w(w0)
        hm.deal_with_w(z)
        wk = hm.get_w()
        for u in deepest:
            iu = pb["indexing"][u]
            if k < iu.size:
                resp[u][:, k] = wk[iu] - w1[iu]
    for u in deepest:
        iu = pb["indexing"][u]
        S = Qpost[np.ix_(iu, iu)]
        if pb["res_is_ref"][labels.size - 1] == 0:
            S = np.diag(np.diag(S))
        cov = resp[u] @ resp[u].T
        assert np.abs(cov @ S - np.eye(iu.size)).max() < 1e-8, u
    hm.close()


# ---------------------------------------------------------------------------------------------------------------------
# (c) failure protocol
# ---------------------------------------------------------------------------------------------------------------------
def _relabelled_problem(level_from_bottom, side=16, n_relabel=12):
    """A univariate tree whose rows on ONE level (counted from the deepest observed level) are relabelled as a second
    outcome.  With a negative Dmat entry (outside the reference's bounds, but finite) the Apanasovich-Genton cross-covariance
    between the two outcomes exceeds what a valid model allows, so exactly that level's conditional variances go negative."""
    pb = make_problem(side=side, q=1, seed=3)
    labels = np.unique(pb["block_groups"])
    lev = labels[labels.size - 1 - level_from_bottom]
    rows = np.concatenate([pb["indexing"][u] for u in range(len(pb["indexing"])) if pb["block_groups"][u] == lev])
    pick = np.random.default_rng(0).choice(rows, n_relabel, replace=False)
    mv = pb["mv_id"].copy()
    mv[pick] = 2
    Z = np.zeros((pb["n"], 2))
    Z[np.arange(pb["n"]), mv - 1] = 1.0
    pb.update(mv_id=mv, Z=Z, q=2, bounds=default_bounds(2))
    th = nice_theta(2).copy()
    th[-1] = -0.9
    pb["theta"] = th
    return pb, int(np.nonzero(labels == lev)[0][0])


@pytest.mark.parametrize("level_from_bottom,code", [(1, 2), (0, 3)])
@pytest.mark.parametrize("generic", [False, True])
def test_phase_a_failure_codes_2_and_3(level_from_bottom, code, generic):
    """errtype 2 (reference child block, spamtree_model.cpp:919) and 3 (non-reference row, :958): code, the levels that were
    completed before the failing one (Q5: the reference returns after the failing LEVEL, :971-982), the untouched accepted
    slot, and recovery of the proposal slot."""
    pb, fail_level = _relabelled_problem(level_from_bottom)
    good = nice_theta(2)
    rng = np.random.default_rng(1)
    w0 = rng.standard_normal(pb["n"])
    om = oracle_model(pb, theta=good, w=w0, tausq=0.2)
    hm = hip_model(pb, theta=good, w=w0, tausq=0.2, force_generic=generic)
    assert om.get_loglik_comps_w(om.param_data) and hm.get_loglik_comps_w(0)
    ll0 = hm.loglik_w[0]
    om.theta_update(om.alter_data, pb["theta"])
    hm.theta_update(1, pb["theta"])
    assert om.get_loglik_comps_w(om.alter_data) is False and om.last_errtype == code
    assert hm.get_loglik_comps_w(1) is False and hm.last_errtype == code
    # levels above the failing one were completed by both: same caches
    ld, ll = hm.comps(1)
    labels = np.unique(pb["block_groups"])
    checked = 0
    for u in range(om.n_blocks):
        g = int(np.nonzero(labels == pb["block_groups"][u])[0][0])
        if g >= fail_level or om.block_ct_obs[u] == 0:
            continue
        assert abs(ld[u] - om.alter_data.logdetCi_comps[u]) <= 1e-9 * max(1.0, abs(ld[u]))
        assert abs(ll[u] - om.alter_data.loglik_w_comps[u]) <= 1e-9 * max(1.0, abs(ll[u]))
        _, Ri = hm.block(1, u)
        assert relerr(Ri, om.alter_data.Rcc_invchol[u]) <= 1e-9
        checked += 1
    assert checked >= 1
    # the accepted slot is untouched: its log-density and a sweep from it still agree with the oracle
    assert hm.get_loglik_w(0) == pytest.approx(ll0, rel=1e-12)
    z = rng.standard_normal(pb["n"])
    om.gibbs_sample_w(z)
    hm.deal_with_w(z)
    assert relerr(hm.get_w()[om.na_ix_all], om.w[om.na_ix_all]) <= 1e-9
    # the proposal slot recovers with a valid theta
    th2 = good * 1.03
    om.theta_update(om.alter_data, th2)
    hm.theta_update(1, th2)
    assert om.get_loglik_comps_w(om.alter_data) and hm.get_loglik_comps_w(1)
    assert abs(hm.loglik_w[1] - om.alter_data.loglik_w) <= 1e-9 * abs(om.alter_data.loglik_w)
    hm.close()


@pytest.mark.parametrize("case,code", [("all_rows", 10), ("all_rows_no_leaf_level", 10), ("leaf_rows_only", 11)])
@pytest.mark.parametrize("generic", [False, True])
def test_sweep_failure_codes_10_and_11(case, code, generic):
    """A negative tausq (finite, outside the prior's support) makes posterior precisions indefinite: the reference stops with
    "Error at gibbs_sample_w" (spamtree_model.cpp:1056 reference blocks -> 10, :1135 non-reference rows -> 11, :1215-1217).
    The code is the oracle's: the last one set, i.e. that of the shallowest failing level.  "leaf_rows_only": the second
    outcome lives on the non-reference level only and only ITS tausq is negative, so no reference block fails."""
    from spamtree_amd.model import SpamTreeError
    if case == "leaf_rows_only":
        pb, _ = _relabelled_problem(0)
        theta = nice_theta(2)
        bad = np.array([5.0, -1e6])
    else:
        pb = make_problem(side=25, q=1, seed=2, last_not_reference=(case == "all_rows"))
        theta = pb["theta"]
        bad = np.array([-1e3])
    good = np.full(pb["q"], 5.0)
    rng = np.random.default_rng(4)
    w0 = rng.standard_normal(pb["n"])
    om = oracle_model(pb, theta=theta, w=w0, tausq=0.2)
    hm = hip_model(pb, theta=theta, w=w0, tausq=0.2, force_generic=generic)
    assert om.get_loglik_comps_w(om.param_data) and hm.get_loglik_comps_w(0)

    def set_tausq_inv(t):
        om.tausq_inv = t.copy()
        om.tausq_inv_long = t[pb["mv_id"] - 1].astype(np.float64)
        hm.tausq_inv = t.copy()
        assert hm.lib.st_set_tausq_inv(hm.h, hm.tausq_inv.ctypes.data_as(hm.lib.st_set_tausq_inv.argtypes[1])) == 0

    z = rng.standard_normal(pb["n"])      # one good sweep first: the reference's per-row caches start as q x q zeros
    om.gibbs_sample_w(z)                  # (spamtree_model.cpp:486), which a failing first sweep would trip over
    hm.deal_with_w(z)
    set_tausq_inv(bad)
    z = rng.standard_normal(pb["n"])
    with np.errstate(all="ignore"), pytest.raises(RuntimeError):
        om.gibbs_sample_w(z)
    assert om.last_sample_errtype == code
    rc = hm.lib.st_sample_w(hm.h, z.ctypes.data_as(hm.lib.st_sample_w.argtypes[1]), 0, 0)
    assert rc == code
    with pytest.raises(SpamTreeError):
        hm.deal_with_w(z)
    # a valid tausq afterwards: the handle is not poisoned (the reference would have stopped the fit)
    set_tausq_inv(good)
    om.w = w0.copy()
    hm.set_w(w0)
    om.gibbs_sample_w(z)
    hm.deal_with_w(z)
    assert relerr(hm.get_w()[om.na_ix_all], om.w[om.na_ix_all]) <= 1e-9
    hm.close()


def test_swap_is_refused_while_top_levels_are_in_flight(monkeypatch):
    """st_swap between st_factor_begin and its st_factor would turn the arena being written into the accepted slot."""
    monkeypatch.setenv("SPAMTREE_QUAD_MIN", "1")      # small levels take k_factor_quad, so the levels above them run ahead
    pb = make_problem(side=40, q=1, seed=21, random_coords=True)
    hm = hip_model(pb, tausq=0.2)
    assert hm.get_loglik_comps_w(0)
    th = np.ascontiguousarray(pb["theta"] * 1.02)
    dp = th.ctypes.data_as(hm.lib.st_factor_begin.argtypes[2])
    if hm.lib.st_factor_ahead_levels(hm.h) == 0:
        hm.close()
        pytest.skip("tree does not qualify for the ahead-of-time top levels")
    assert hm.lib.st_factor_begin(hm.h, 1, dp, th.size) == 0
    assert hm.lib.st_swap(hm.h) == -1
    hm.theta_update(1, th)
    assert hm.get_loglik_comps_w(1)
    assert hm.lib.st_swap(hm.h) == 0
    hm.close()
