"""GPU parity: the HIP path (through the C-ABI) against the NumPy oracle on the same seeded inputs.

Tolerances (FP64): the two sides evaluate the same formulas in different association orders (the HIP build
works from the chain of inverse-Cholesky panels instead of the dense K_xx^{-1}); the stated bound is
REL = 1e-9 relative to the largest magnitude of the compared array for per-block caches, log-likelihoods and
draws of w after three sweeps.
"""
import numpy as np
import pytest

from tests.util import make_problem, oracle_model

pytestmark = pytest.mark.gpu
REL = 1e-9


def hip_model(pb, theta=None, beta=None, tausq=0.1, w=None, **kw):
    from spamtree_amd.model import SpamTreeMV
    theta = pb["theta"] if theta is None else theta
    beta = np.zeros(pb["p"]) if beta is None else beta
    w = np.zeros(pb["n"]) if w is None else w
    return SpamTreeMV(pb["y"], pb["X"], pb["Z"], pb["coords"], pb["mv_id"], pb["blocking"], pb["gix_block"],
                      pb["res_is_ref"], pb["parents"], pb["children"], pb.get("limited_tree", False), pb["block_names"],
                      pb["block_groups"], pb["indexing"], w, beta, theta, 1.0 / tausq, **kw)


def relerr(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    if a.size == 0:
        return 0.0
    return float(np.abs(a - b).max() / max(1e-300, np.abs(b).max()))


CASES = [
    dict(side=25, q=1, missing=0.0),
    dict(side=25, q=1, missing=0.12),
    dict(side=40, q=1, missing=0.0, random_coords=True),
    dict(side=16, q=2, missing=0.0),
    dict(side=14, q=3, missing=0.2),
    dict(side=25, q=1, missing=0.0, last_not_reference=False),
    # q = 3 with cell_size = 9 (config #5's shape): 27-row blocks -> the MFMA kernels evaluate the Apanasovich-Genton form
    dict(side=18, q=3, missing=0.25, cell_size=9),
    dict(side=20, q=2, missing=0.0, cell_size=16),
    # config #5 AS SPECIFIED (BASELINE.json configs[4]; NA census spamtree_model.cpp:303-313, quirk Q3 :1375): outcomes dropped
    # with probabilities 10 / 30 / 50 %, q = 3, cell_size = 9, the NA rows in their own prediction level
    dict(side=24, q=3, missing=(0.1, 0.3, 0.5), cell_size=9),
    # the default cell size with the same imbalanced pattern: wide blocks of unequal widths (siblings no longer share a row stride,
    # chains of odd lengths) on k_factor_lchain + k_factor_ref_finish and the generic sweep kernel
    dict(side=30, q=3, missing=(0.1, 0.3, 0.5)),
    # unusual trees: 3 x 2 branching, small cells, a finite depth with a nearest-neighbour leftover level
    dict(side=30, q=1, missing=0.1, cell_size=9, K=(3, 2)),
    dict(side=30, q=1, missing=0.0, tree_depth=2),
    dict(side=36, q=1, missing=0.05, cell_size=16, random_coords=True),
    # degenerate sizes: one block only (a one-level tree: the exact GP), a few tiny ragged blocks with prediction blocks,
    # two blocks of a trivariate problem, 4-row cells
    dict(side=5, q=1, missing=0.0),
    dict(side=7, q=1, missing=0.3),
    dict(side=4, q=3, missing=0.2),
    dict(side=9, q=2, missing=0.0, cell_size=4),
]


@pytest.mark.parametrize("case", CASES)
@pytest.mark.parametrize("generic", [False, True])
def test_factor_sample_loglik_predict_match_oracle(case, generic):
    pb = make_problem(seed=17, **case)
    rng = np.random.default_rng(5)
    w0 = rng.standard_normal(pb["n"])
    beta = np.array([0.3, -0.2, 0.1])
    om = oracle_model(pb, w=w0, beta=beta, tausq=0.2)
    hm = hip_model(pb, w=w0, beta=beta, tausq=0.2, force_generic=generic)
    if np.ndim(case["missing"]) > 0:      # the imbalanced pattern really is imbalanced, and there is something to predict
        n_obs = np.array([np.sum(np.isfinite(pb["y"][pb["mv_id"] == j + 1])) for j in range(pb["q"])])
        assert n_obs[0] > n_obs[1] > n_obs[2] and np.any(np.asarray(om.block_ct_obs) == 0)
    # ---- phase A on both slots
    assert om.get_loglik_comps_w(om.param_data)
    assert hm.get_loglik_comps_w(0)
    assert abs(hm.loglik_w[0] - om.param_data.loglik_w) <= REL * abs(om.param_data.loglik_w)
    ld, ll = hm.comps(0)
    assert relerr(ld, om.param_data.logdetCi_comps) <= REL
    assert relerr(ll, om.param_data.loglik_w_comps) <= REL
    for u in range(om.n_blocks):
        if om.block_ct_obs[u] == 0:
            continue
        H, Ri = hm.block(0, u)
        if om.parents[u].size:
            assert relerr(H, om.param_data.w_cond_mean_K[u]) <= 1e-8, u
        if om.block_is_reference[u]:
            assert relerr(Ri, om.param_data.Rcc_invchol[u]) <= REL, u
        else:
            assert relerr(Ri, om.param_data.ccholprecdiag[u]) <= REL, u
    # ---- three sweeps + loglik + stats
    for it in range(3):
        z = rng.standard_normal(pb["n"])
        om.gibbs_sample_w(z)
        hm.deal_with_w(z)
        assert relerr(hm.get_w()[om.na_ix_all], om.w[om.na_ix_all]) <= REL, it
        om.get_loglik_w(om.param_data)
        assert abs(hm.get_loglik_w(0) - om.param_data.loglik_w) <= REL * abs(om.param_data.loglik_w)
        xty, ssq = hm.stats()
        oxty, ossq = om.beta_tausq_stats()
        assert relerr(xty, oxty) <= REL and relerr(ssq, ossq) <= REL
    # ---- prediction at the NA blocks (reuses the last sweep's normals)
    om.predict(True)
    hm.predict(True)
    assert relerr(hm.get_w(), om.w) <= REL
    # ---- beta / tausq conjugate updates fed with the same draws
    nb = [rng.standard_normal(pb["p"]) for _ in range(pb["q"])]
    om.gibbs_sample_tausq(lambda j, a, b: a * b)
    hm.gibbs_sample_tausq(lambda j, a, b: a * b)
    assert relerr(hm.tausq_inv, om.tausq_inv) <= REL
    om.gibbs_sample_beta(nb)
    hm.gibbs_sample_beta(nb)
    assert relerr(hm.Bcoeff, om.Bcoeff) <= REL
    assert relerr(hm.get_XB(), om.XB) <= REL
    # ---- a proposal on the other slot, accept, sweep again
    th2 = pb["theta"] * (1.0 + 0.05 * rng.standard_normal(pb["theta"].size))
    om.theta_update(om.alter_data, th2)
    hm.theta_update(1, th2)
    assert om.get_loglik_comps_w(om.alter_data) and hm.get_loglik_comps_w(1)
    assert abs(hm.loglik_w[1] - om.alter_data.loglik_w) <= REL * abs(om.alter_data.loglik_w)
    om.accept_make_change()
    hm.accept_make_change()
    z = rng.standard_normal(pb["n"])
    om.gibbs_sample_w(z)
    hm.deal_with_w(z)
    assert relerr(hm.get_w()[om.na_ix_all], om.w[om.na_ix_all]) <= REL
    hm.close()


def test_cholesky_failure_codes():
    pb = make_problem(side=25, q=1, seed=2)
    th = pb["theta"].copy()
    th[0] = -1.0
    hm = hip_model(pb, theta=th)
    assert hm.get_loglik_comps_w(0) is False and hm.last_errtype == 1
    hm.theta_update(0, pb["theta"])
    assert hm.get_loglik_comps_w(0) is True
    hm.close()


def test_device_normals_match_oracle_stream():
    from oracle.spamtree_oracle import StRng
    pb = make_problem(side=25, q=1, seed=2)
    hm = hip_model(pb, tausq=1e-12)          # tausq_inv huge: w ~ y - XB + tiny noise is NOT what we test; see below
    hm.close()
    # direct check through yhat: XB = 0, w = 0, tausq_inv = 1  ->  yhat = normal(stream 5)
    hm = hip_model(pb, tausq=1.0)
    got = hm.yhat(None, seed=2021, it=7)
    exp = StRng(2021).yhat_normals(7, pb["n"])
    assert np.abs(got - exp).max() < 1e-13
    hm.close()


def test_generated_sweep_normals_are_the_documented_stream():
    from oracle.spamtree_oracle import StRng
    pb = make_problem(side=25, q=1, seed=4)
    om = oracle_model(pb, tausq=0.3)
    hm = hip_model(pb, tausq=0.3)
    assert om.get_loglik_comps_w(om.param_data) and hm.get_loglik_comps_w(0)
    om.gibbs_sample_w(StRng(77).sweep_normals(3, pb["n"]))
    hm.deal_with_w(None, seed=77, it=3)
    assert relerr(hm.get_w(), om.w) <= REL
    hm.close()


def test_gram_cache_gives_identical_sweeps():
    """caching the theta-only part of the messages per accepted theta must not change the draws.  The first
    sweep after a factorisation rebuilds the Gram parts in both models (same kernel: bit-identical); later sweeps of the
    caching model take k_sample_lean on reference levels, the other keeps k_sample_mfma: equal up to rounding."""
    pb = make_problem(side=40, q=1, seed=21, random_coords=True)
    rng = np.random.default_rng(1)
    a = hip_model(pb, tausq=0.2)
    b = hip_model(pb, tausq=0.2, cache_gram=False)
    assert a.get_loglik_comps_w(0) and b.get_loglik_comps_w(0)
    for it in range(3):
        z = rng.standard_normal(pb["n"])
        a.deal_with_w(z); b.deal_with_w(z)
        if it == 0:
            assert np.array_equal(a.get_w(), b.get_w())
        assert relerr(a.get_w(), b.get_w()) <= 1e-12
    th2 = pb["theta"] * 1.02
    for m in (a, b):
        m.theta_update(1, th2)
        assert m.get_loglik_comps_w(1)
        m.accept_make_change()
    for it in range(2):
        z = rng.standard_normal(pb["n"])
        a.deal_with_w(z); b.deal_with_w(z)
        assert relerr(a.get_w(), b.get_w()) <= 1e-12
    a.close(); b.close()


@pytest.mark.parametrize("case", [CASES[0], CASES[3], CASES[6], CASES[8]])
def test_lean_sample_kernel_matches_staged_kernel(case, monkeypatch):
    """k_sample_lean (sweeps with cached Gram parts, reference levels) against k_sample_mfma (SPAMTREE_SAMPLE_LEAN=0) and
    the oracle: w after three sweeps."""
    pb = make_problem(seed=51, **case)
    rng = np.random.default_rng(3)
    zs = [rng.standard_normal(pb["n"]) for _ in range(3)]
    ws = []
    for lean in ("1", "0"):
        monkeypatch.setenv("SPAMTREE_SAMPLE_LEAN", lean)
        hm = hip_model(pb, tausq=0.2)
        assert hm.get_loglik_comps_w(0)
        for z in zs:
            hm.deal_with_w(z)
        ws.append(hm.get_w().copy())
        hm.close()
    om = oracle_model(pb, tausq=0.2)
    assert om.get_loglik_comps_w(om.param_data)
    for z in zs:
        om.gibbs_sample_w(z)
    assert relerr(ws[0], ws[1]) <= 1e-11
    assert relerr(ws[0][om.na_ix_all], om.w[om.na_ix_all]) <= REL


@pytest.mark.parametrize("case", [CASES[0], CASES[1], CASES[3], CASES[6], CASES[8], CASES[5]])
def test_wave_sample_kernel_matches_lean_kernel_and_oracle(case, monkeypatch):
    """k_sample_wave (one reference block per wave; by default only on levels of >= 32 x CUs blocks, forced here) against
    k_sample_lean -- same arithmetic and summation orders: identical draws -- and the oracle, incl. a rebuild sweep (k_gram)."""
    pb = make_problem(seed=52, **case)
    rng = np.random.default_rng(4)
    zs = [rng.standard_normal(pb["n"]) for _ in range(3)]
    ws = []
    monkeypatch.setenv("SPAMTREE_SPLIT_GRAM", "2")            # k_gram on every level (default: big reference levels only)
    for wave in ("2", "0"):
        monkeypatch.setenv("SPAMTREE_SAMPLE_WAVE", wave)
        hm = hip_model(pb, tausq=0.2)
        assert hm.get_loglik_comps_w(0)
        for z in zs:
            hm.deal_with_w(z)
        ws.append(hm.get_w().copy())
        hm.close()
    om = oracle_model(pb, tausq=0.2)
    assert om.get_loglik_comps_w(om.param_data)
    for z in zs:
        om.gibbs_sample_w(z)
    assert np.array_equal(ws[0], ws[1])
    assert relerr(ws[0][om.na_ix_all], om.w[om.na_ix_all]) <= REL


@pytest.mark.parametrize("case", [CASES[0], CASES[1], CASES[6], CASES[7], CASES[9]])
def test_direct_gram_of_the_last_reference_level_is_bit_identical(case, monkeypatch):
    """Round 3: on a rebuild sweep the leaf level writes no Gram parts; its parents form them from the leaf groups' panels
    (k_gram_direct) instead of summing the children's records (k_gram).  Same arithmetic in the same order: the draws of
    sweeps after a factorisation and after an accepted theta must be IDENTICAL to the record route (SPAMTREE_GRAM_DIRECT=0).
    SPAMTREE_SPLIT_GRAM=2 puts every level on the split route, as the big levels of n = 1e6 are by default."""
    pb = make_problem(seed=53, **case)
    rng = np.random.default_rng(6)
    zs = [rng.standard_normal(pb["n"]) for _ in range(4)]
    monkeypatch.setenv("SPAMTREE_SPLIT_GRAM", "2")
    ws = []
    for direct in ("1", "0"):
        monkeypatch.setenv("SPAMTREE_GRAM_DIRECT", direct)
        hm = hip_model(pb, tausq=0.2)
        assert hm.get_loglik_comps_w(0)
        out = []
        for it, z in enumerate(zs):
            if it == 2:      # an accepted proposal: the records are rebuilt again
                hm.theta_update(1, pb["theta"] * 1.03)
                assert hm.get_loglik_comps_w(1)
                hm.accept_make_change()
            hm.deal_with_w(z)
            out.append(hm.get_w().copy())
        ws.append(out)
        hm.close()
    for a, b in zip(*ws):
        assert np.array_equal(a, b)
    om = oracle_model(pb, tausq=0.2)
    assert om.get_loglik_comps_w(om.param_data)
    om.gibbs_sample_w(zs[0]); om.gibbs_sample_w(zs[1])
    assert relerr(ws[0][1][om.na_ix_all], om.w[om.na_ix_all]) <= REL


def test_cross_covariance_ag10_export():
    """man/CrossCovarianceAG10.Rd:72-93 inputs (q = 2) and a q = 3 parameter set, device vs oracle (itself pinned by mpmath)."""
    from oracle import spamtree_oracle as so
    from spamtree_amd.covariance import CrossCovarianceAG10
    from spamtree_amd.model import SpamTreeError
    from tests.util import nice_theta
    xl = np.linspace(0.0, 1.0, 10)
    g = np.array([(a, b) for b in xl for a in xl])
    cx = np.vstack([g, g])
    mv = np.repeat([1, 2], 100)
    D = np.array([[0, 1.0], [1.0, 0]])
    got = CrossCovarianceAG10(cx, mv, cx, mv, [1, 1.5], [.1, .51], [1, 2], [5], D)
    ref = so.CrossCovarianceAG10(cx, mv, cx, mv, [1, 1.5], [.1, .51], [1, 2], [5], D)
    assert got.shape == (200, 200) and np.abs(got - ref).max() <= 1e-14 * np.abs(ref).max()
    cp = so.CovarianceParams(2, 3, -1)
    cp.transform(nice_theta(3))
    rng = np.random.default_rng(2)
    p1, p2 = rng.uniform(size=(70, 2)), rng.uniform(size=(55, 2))
    m1, m2 = rng.integers(1, 4, 70), rng.integers(1, 4, 55)
    got = CrossCovarianceAG10(p1, m1, p2, m2, cp.ai1, cp.ai2, cp.phi_i, cp.thetamv, cp.Dmat)
    ref = so.CrossCovarianceAG10(p1, m1, p2, m2, cp.ai1, cp.ai2, cp.phi_i, cp.thetamv, cp.Dmat)
    assert np.abs(got - ref).max() <= 1e-14 * np.abs(ref).max()
    with pytest.raises(SpamTreeError):                      # the reference stops: "Invalid Dmat for multivariate data"
        CrossCovarianceAG10(p1, np.ones(70), p2, np.ones(55), [1.0], [0.1], [1.0], [5.0], np.zeros((1, 1)))


@pytest.mark.parametrize("keep", [7, 40])
def test_posterior_means_and_quantiles_match_oracle_chain(keep):
    """Running means and per-row quantiles of the saved draws on the device against list_mean / list_qtile
    (/root/reference/src/list_mean.cpp:10-30, 62-137) applied to the ORACLE's draws of the same sweeps (same Philox streams):
    w and yhat = XB + w + tausq^(1/2) * normal (spamtree_fit.cpp:384)."""
    import ctypes as C
    from oracle.list_summaries import list_mean, list_qtile
    from oracle.spamtree_oracle import StRng
    from spamtree_amd.model import _dp
    pb = make_problem(side=25, q=1, seed=8, missing=0.1)
    beta = np.array([0.3, -0.2, 0.1])
    om = oracle_model(pb, tausq=0.2, beta=beta)
    hm = hip_model(pb, tausq=0.2, beta=beta)
    assert om.get_loglik_comps_w(om.param_data) and hm.get_loglik_comps_w(0)
    rng_o = StRng(3)
    ws, ys = [], []
    assert hm.lib.st_summary_reset(hm.h) == 0
    assert hm.lib.st_summary_reserve(hm.h, keep) == 0
    for it in range(keep):
        om.gibbs_sample_w(rng_o.sweep_normals(it, pb["n"]))
        om.predict(True)
        hm.deal_with_w(None, seed=3, it=it)
        hm.predict(True)
        ws.append(om.w.copy())
        ys.append(om.XB + om.w + np.sqrt(1.0 / om.tausq_inv_long) * rng_o.yhat_normals(it, pb["n"]))
        assert hm.lib.st_summary_accumulate(hm.h, 3, it) == 0
    wm, ym = np.zeros(pb["n"]), np.zeros(pb["n"])
    cnt = C.c_int64()
    assert hm.lib.st_summary_get(hm.h, _dp(wm), _dp(ym), C.byref(cnt)) == 0 and cnt.value == keep
    assert relerr(wm, list_mean(ws)) <= REL and relerr(ym, list_mean(ys)) <= REL
    for q in (0.025, 0.5, 0.975, 0.0, 1.0, 0.3):
        wq, yq = np.zeros(pb["n"]), np.zeros(pb["n"])
        assert hm.lib.st_summary_quantile(hm.h, q, _dp(wq), _dp(yq)) == 0
        assert relerr(wq, list_qtile(ws, q)) <= REL, q
        assert relerr(yq, list_qtile(ys, q)) <= REL, q
    assert hm.lib.st_summary_reserve(hm.h, 0) == 0
    assert hm.lib.st_summary_quantile(hm.h, 0.5, _dp(wm), None) < 0       # nothing stored any more
    hm.close()


@pytest.mark.parametrize("gen", ["1", "3", "3:2", "3:1"])
@pytest.mark.parametrize("case", [CASES[0], CASES[1], CASES[6], CASES[8]])
def test_factor_kernel_generations(case, gen, monkeypatch):
    """Both phase-A kernels for column-group levels (SPAMTREE_FACTOR_KERNEL, read at st_create: 1 = k_factor_mfma
    everywhere, 3 = k_factor_quad; SPAMTREE_QUAD_MIN=1 makes even these tiny levels eligible) give the oracle's factors."""
    gen, _, units = gen.partition(":")        # "3:2": k_factor_quad with at most 2 units per workgroup (default here: 4)
    monkeypatch.setenv("SPAMTREE_FACTOR_KERNEL", gen)
    monkeypatch.setenv("SPAMTREE_QUAD_MIN", "1")
    monkeypatch.setenv("SPAMTREE_QUAD_UNITS", units or "4")
    pb = make_problem(seed=31, **case)
    rng = np.random.default_rng(6)
    w0 = rng.standard_normal(pb["n"])
    om = oracle_model(pb, w=w0, tausq=0.2)
    hm = hip_model(pb, w=w0, tausq=0.2)
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
    z = rng.standard_normal(pb["n"])
    om.gibbs_sample_w(z); hm.deal_with_w(z)
    assert relerr(hm.get_w()[om.na_ix_all], om.w[om.na_ix_all]) <= REL
    hm.close()


@pytest.mark.parametrize("ref_route", ["0", "1"])
def test_big_block_kernel_two_passes(ref_route, monkeypatch):
    """The default multivariate tree (q = 3: 75-row blocks) deep enough for chains of 294 rows: factors against the oracle and
    against the generic kernels.  ref_route = "0": k_factor_bigmfma, both passes over the chain (chain tiles 0-16 and 17-..);
    "1" (the default): reference levels on k_factor_lchain + k_factor_ref_finish."""
    monkeypatch.setenv("SPAMTREE_LCHAIN_REF", ref_route)
    pb = make_problem(side=50, q=3, seed=3)
    rng = np.random.default_rng(2)
    w0 = rng.standard_normal(pb["n"])
    om = oracle_model(pb, w=w0, tausq=0.2)
    hm = hip_model(pb, w=w0, tausq=0.2)
    hg = hip_model(pb, w=w0, tausq=0.2, force_generic=True)
    assert om.get_loglik_comps_w(om.param_data) and hm.get_loglik_comps_w(0) and hg.get_loglik_comps_w(0)
    assert max(om.parents_indexing[u].size for u in range(om.n_blocks)) > 272
    assert abs(hm.loglik_w[0] - om.param_data.loglik_w) <= REL * abs(om.param_data.loglik_w)
    assert abs(hm.loglik_w[0] - hg.loglik_w[0]) <= REL * abs(hg.loglik_w[0])
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
    z = rng.standard_normal(pb["n"])
    om.gibbs_sample_w(z); hm.deal_with_w(z)
    assert relerr(hm.get_w()[om.na_ix_all], om.w[om.na_ix_all]) <= REL
    hm.close(); hg.close()
