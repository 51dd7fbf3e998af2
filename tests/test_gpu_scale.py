"""GPU tests beyond oracle size: mid-size parity against the OpenMP restatement (oracle/refcpu), and size-independent
properties at BASELINE.json's FULL sizes (configs #2, #3, #4 and #5 on one GPU)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REL = 1e-9


def build(side, q=1, cell_size=25, missing=None, **kw):
    from spamtree_amd.model import SpamTreeMV
    from spamtree_amd.synthetic import make_workload
    wl = make_workload(side, q=q, cell_size=cell_size, missing=missing)
    m = SpamTreeMV(wl["y"], wl["X"], wl["Z"], wl["coords"], wl["mv_id"], wl["blocking"], wl["gix_block"],
                   wl["res_is_ref"], wl["parents"], wl["children"], False, wl["block_names"], wl["block_groups"],
                   wl["indexing"], np.zeros(wl["n"]), np.array([-0.5, 0.2, 0.4]), wl["theta"], 1.0 / 0.15, **kw)
    return wl, m


def relerr(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(1e-300, np.abs(np.asarray(b)).max()))


@pytest.mark.parametrize("side,q,cell_size,missing", [(150, 1, 25, None), (48, 3, 25, None), (60, 3, 9, (0.1, 0.3, 0.5))])
def test_midsize_matches_refcpu(side, q, cell_size, missing):
    """6 levels, ancestor chains up to 125 rows (q=1) / 225 rows with 75-row blocks (q=3, generic kernels); config #5's shape
    (cell_size = 9, outcomes dropped with probabilities 10 / 30 / 50 %: 27-row blocks, ragged leaves, a prediction level)."""
    from oracle.refcpu import RefCpu
    wl, hm = build(side, q, cell_size, missing)
    rc = RefCpu(wl["y"], wl["X"], wl["coords"], wl["mv_id"], wl["res_is_ref"], wl["parents"], wl["children"],
                wl["block_names"], wl["block_groups"], wl["indexing"], threads=16)
    rng = np.random.default_rng(3)
    w0 = rng.standard_normal(wl["n"])
    hm.set_w(w0); rc.set_w(w0)
    rc.set_beta(np.tile(np.array([-0.5, 0.2, 0.4])[:, None], (1, q))); rc.set_tausq_inv(1.0 / 0.15)
    code, ll = rc.factor(0, wl["theta"])
    assert code == 0 and hm.get_loglik_comps_w(0)
    assert abs(hm.loglik_w[0] - ll) <= REL * abs(ll)
    ld, lc = hm.comps(0)
    rld, rlc = rc.comps(0)
    assert relerr(ld, rld) <= REL and relerr(lc, rlc) <= REL
    for it in range(2):
        z = rng.standard_normal(wl["n"])
        assert rc.sample_w(z) == 0
        hm.deal_with_w(z)
        obs = np.isfinite(wl["y"])        # rows of prediction blocks are not touched by the sweep
        assert relerr(hm.get_w()[obs], rc.get_w()[obs]) <= REL
        assert abs(hm.get_loglik_w(0) - rc.loglik_w(0)) <= REL * abs(rc.loglik_w(0))
    xty, ssq = hm.stats()
    rxty, rssq = rc.stats()
    assert relerr(xty, rxty) <= REL and relerr(ssq, rssq) <= REL
    rc.close(); hm.close()


def test_fast_and_generic_kernels_agree_midsize():
    wl, a = build(120)
    _, b = build(120, force_generic=True)
    rng = np.random.default_rng(5)
    w0 = rng.standard_normal(wl["n"])
    a.set_w(w0); b.set_w(w0)
    assert a.get_loglik_comps_w(0) and b.get_loglik_comps_w(0)
    assert abs(a.loglik_w[0] - b.loglik_w[0]) <= REL * abs(b.loglik_w[0])
    z = rng.standard_normal(wl["n"])
    a.deal_with_w(z); b.deal_with_w(z)
    assert relerr(a.get_w(), b.get_w()) <= REL
    a.close(); b.close()


FULL = [
    pytest.param(1000, 1, 25, None, id="config3_n1e6"),
    pytest.param(316, 1, 25, None, id="config2_n1e5"),
    pytest.param(577, 3, 25, None, id="config4_n1e6_q3"),
    pytest.param(1155, 3, 9, (0.1, 0.3, 0.5), id="config5_n4e6_q3_missing"),
]


@pytest.mark.parametrize("side,q,cell_size,missing", FULL)
def test_full_size_properties(side, q, cell_size, missing):
    """BASELINE.json's configurations at FULL size on one GPU: properties that need no oracle."""
    wl, hm = build(side, q, cell_size, missing)
    rng = np.random.default_rng(7)
    obs = np.isfinite(wl["y"])
    # (1) the factorisation pass and the cached-factor pass are two code paths for the same density
    hm.set_w(rng.standard_normal(wl["n"]) * 0.3)
    assert hm.get_loglik_comps_w(0)
    ll_a = hm.loglik_w[0]
    assert abs(hm.get_loglik_w(0) - ll_a) <= 1e-10 * abs(ll_a)
    # (2) both cache slots give bit-identical results for the same theta
    hm.theta_update(1, wl["theta"])
    assert hm.get_loglik_comps_w(1) and hm.loglik_w[1] == ll_a
    # (3) a sweep is bit-reproducible (no atomics in the message reduction) and the device stream is counter-based
    hm.deal_with_w(None, seed=11, it=3)   # the first sweep after a factorisation also rebuilds the Gram parts (other kernel)
    w_before = hm.get_w().copy()
    hm.deal_with_w(None, seed=11, it=4)
    w1 = hm.get_w().copy()
    hm.set_w(w_before)
    hm.deal_with_w(None, seed=11, it=4)
    assert np.array_equal(hm.get_w(), w1)
    # (3b) rows without an observation belong to prediction blocks: the sweep leaves them alone (spamtree_model.cpp:1024-1029
    # loops over u_by_block_groups = blocks with observations)
    if missing is not None:
        assert not np.all(obs) and np.array_equal(w1[~obs], w_before[~obs])
    # (4) with beta at its data-generating value, sweeps pull w towards y - XB at the observed rows
    hm.beta_update(np.tile(wl["beta_true"][:, None], (1, q)))
    r0 = np.where(obs, wl["y"], 0.0) - hm.get_XB()
    ssq_before = np.sum((r0 - w_before)[obs] ** 2)
    for it in range(3):
        hm.deal_with_w(None, seed=11, it=5 + it)
    w2 = hm.get_w()
    # (q = 1: a third of what it was after three sweeps; the trivariate workloads' prior is deliberately not the one that
    # generated the field -- opposite-sign a_i1 against positively related outcomes -- so they only have to improve clearly)
    assert np.all(np.isfinite(w2)) and np.sum((r0 - w2)[obs] ** 2) < (0.5 if q == 1 else 0.75) * ssq_before
    # (4b) the statistics of the conjugate updates see observed rows only (NA census, spamtree_model.cpp:303-313)
    xty, ssq = hm.stats()
    assert np.all(np.isfinite(xty)) and np.all(ssq > 0)
    for j in range(q):
        oj = obs & (wl["mv_id"] == j + 1)
        assert abs(ssq[j] - np.sum((r0 - w2)[oj] ** 2)) <= 1e-9 * ssq[j]
    # (5) rejecting a non-PD proposal leaves the accepted slot untouched
    bad = wl["theta"].copy()
    if q == 1:
        bad[0] = -1.0                       # sigma^2 < 0
    else:
        bad[0] = 0.0; bad[q] = 0.0          # outcome 1: a_i1 = a_i2 = 0 -> zero marginal variance -> a non-positive pivot
    hm.theta_update(1, bad)
    assert hm.get_loglik_comps_w(1) is False and hm.last_errtype in (1, 2, 3)
    assert abs(hm.get_loglik_w(0) - hm.loglik_w[0]) == 0.0
    # (6) prediction at the NA blocks fills finite values and touches nothing else
    if missing is not None:
        w3 = hm.get_w().copy()
        hm.predict(True)
        w4 = hm.get_w()
        assert np.all(np.isfinite(w4)) and np.array_equal(w4[obs], w3[obs]) and not np.array_equal(w4[~obs], w3[~obs])
    hm.close()
