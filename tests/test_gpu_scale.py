"""GPU tests beyond oracle size: mid-size parity against the OpenMP restatement (oracle/refcpu), and size-independent
properties at BASELINE.json's full size (n = 1e6, config #3)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
REL = 1e-9


def build(side, q=1, **kw):
    from spamtree_amd.model import SpamTreeMV
    from spamtree_amd.synthetic import make_workload
    wl = make_workload(side, q=q)
    m = SpamTreeMV(wl["y"], wl["X"], wl["Z"], wl["coords"], wl["mv_id"], wl["blocking"], wl["gix_block"],
                   wl["res_is_ref"], wl["parents"], wl["children"], False, wl["block_names"], wl["block_groups"],
                   wl["indexing"], np.zeros(wl["n"]), np.array([-0.5, 0.2, 0.4]), wl["theta"], 1.0 / 0.15, **kw)
    return wl, m


def relerr(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(1e-300, np.abs(np.asarray(b)).max()))


@pytest.mark.parametrize("side,q", [(150, 1), (48, 3)])
def test_midsize_matches_refcpu(side, q):
    """6 levels, ancestor chains up to 125 rows (q=1) / 225 rows with 75-row blocks (q=3, generic kernels)."""
    from oracle.refcpu import RefCpu
    wl, hm = build(side, q)
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
        assert relerr(hm.get_w(), rc.get_w()) <= REL
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


def test_full_size_properties():
    """n = 1e6: properties that need no oracle."""
    wl, hm = build(1000)
    rng = np.random.default_rng(7)
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
    # (4) with beta at its data-generating value, sweeps pull w towards y - XB (latent field variance 2.3, noise 0.1)
    hm.beta_update(np.tile(wl["beta_true"][:, None], (1, 1)))
    r0 = wl["y"] - hm.get_XB()
    ssq_before = np.sum((r0 - w_before) ** 2)
    for it in range(3):
        hm.deal_with_w(None, seed=11, it=5 + it)
    w2 = hm.get_w()
    assert np.all(np.isfinite(w2)) and np.sum((r0 - w2) ** 2) < 0.5 * ssq_before
    # (5) rejecting a non-PD proposal leaves the accepted slot untouched
    bad = wl["theta"].copy(); bad[0] = -1.0
    hm.theta_update(1, bad)
    assert hm.get_loglik_comps_w(1) is False and hm.last_errtype == 1
    assert abs(hm.get_loglik_w(0) - hm.loglik_w[0]) == 0.0
    hm.close()
