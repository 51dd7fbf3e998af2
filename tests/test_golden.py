"""Golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py from the NumPy oracle).

CPU: the oracle and the OpenMP restatement reproduce them (pins both against regressions).
GPU: the HIP path reproduces them through the C-ABI."""
import glob
import os

import numpy as np
import pytest

GOLD = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "*.npz")))
REL = 1e-9


def lists(ptr, idx):
    return [idx[ptr[i]:ptr[i + 1]] for i in range(ptr.size - 1)]


def limited(g):
    return bool(int(g["limited_tree"])) if "limited_tree" in g.files else False


def relerr(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(1e-300, np.abs(np.asarray(b)).max()))


def test_fixtures_exist():
    assert len(GOLD) >= 4


@pytest.mark.parametrize("path", GOLD)
def test_oracle_reproduces_golden(path):
    from oracle.spamtree_oracle import SpamTreeMV
    g = np.load(path)
    om = SpamTreeMV(g["y"], g["X"], g["Z"], g["coords"], g["mv_id"], g["blocking"], g["gix_block"], g["res_is_ref"],
                    lists(g["parents_ptr"], g["parents_idx"]), lists(g["children_ptr"], g["children_idx"]), limited(g),
                    g["block_names"], g["block_groups"], lists(g["indexing_ptr"], g["indexing_idx"]), g["w0"],
                    g["beta"], g["theta"], 1.0 / float(g["tausq"]))
    assert om.get_loglik_comps_w(om.param_data)
    assert abs(om.param_data.loglik_w - float(g["loglik_A"])) <= 1e-12 * abs(float(g["loglik_A"]))
    for it in range(3):
        om.gibbs_sample_w(g["z"][it])
        assert relerr(om.w, g["w_sweeps"][it]) <= 1e-12


@pytest.mark.parametrize("path", GOLD)
def test_refcpu_reproduces_golden(path):
    from oracle.refcpu import RefCpu
    g = np.load(path)
    q = int(np.unique(g["mv_id"]).size)
    rc = RefCpu(g["y"], g["X"], g["coords"], g["mv_id"], g["res_is_ref"], (g["parents_ptr"], g["parents_idx"]),
                (g["children_ptr"], g["children_idx"]), g["block_names"], g["block_groups"],
                (g["indexing_ptr"], g["indexing_idx"]), threads=2, limited_tree=limited(g))
    rc.set_w(g["w0"]); rc.set_beta(np.tile(g["beta"][:, None], (1, q))); rc.set_tausq_inv(1.0 / float(g["tausq"]))
    code, ll = rc.factor(0, g["theta"])
    assert code == 0 and abs(ll - float(g["loglik_A"])) <= REL * abs(ll)
    obs = np.isfinite(g["y"])
    for it in range(3):
        assert rc.sample_w(g["z"][it]) == 0
        assert relerr(rc.get_w()[obs], g["w_sweeps"][it][obs]) <= REL
        assert abs(rc.loglik_w(0) - g["loglik_w"][it]) <= REL * abs(g["loglik_w"][it])
    rc.close()


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLD)
def test_hip_reproduces_golden(path):
    from spamtree_amd.model import SpamTreeMV
    g = np.load(path)
    hm = SpamTreeMV(g["y"], g["X"], g["Z"], g["coords"], g["mv_id"], g["blocking"], g["gix_block"], g["res_is_ref"],
                    (g["parents_ptr"], g["parents_idx"]), (g["children_ptr"], g["children_idx"]), limited(g),
                    g["block_names"], g["block_groups"], (g["indexing_ptr"], g["indexing_idx"]), g["w0"], g["beta"],
                    g["theta"], 1.0 / float(g["tausq"]))
    assert hm.get_loglik_comps_w(0)
    assert abs(hm.loglik_w[0] - float(g["loglik_A"])) <= REL * abs(float(g["loglik_A"]))
    ld, ll = hm.comps(0)
    assert relerr(ld, g["logdet_comps"]) <= REL and relerr(ll, g["loglik_comps"]) <= REL
    for u in g["cache_blocks"]:
        H, Ri = hm.block(0, int(u))
        assert relerr(H, g[f"H_{u}"]) <= 1e-8 and relerr(Ri, g[f"Ri_{u}"]) <= REL
    obs = np.isfinite(g["y"])
    for it in range(3):
        hm.deal_with_w(g["z"][it])
        assert relerr(hm.get_w()[obs], g["w_sweeps"][it][obs]) <= REL
        assert abs(hm.get_loglik_w(0) - g["loglik_w"][it]) <= REL * abs(g["loglik_w"][it])
    xty, ssq = hm.stats()
    assert relerr(xty, g["xty"]) <= REL and relerr(ssq, g["ssq"]) <= REL
    hm.predict(True)
    assert relerr(hm.get_w(), g["w_predict"]) <= REL
    hm.close()
