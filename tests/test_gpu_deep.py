"""GPU parity on the long ancestor chains of BASELINE.json's configs #4 and #5, reached at sizes the NumPy oracle finishes
in about a minute: a thin strip split along one axis only (K = (2, 1)) gives a narrow, deep tree.

  * #4 (q = 3, default cell size: 75-row blocks): reference level 7 has P = 450 > 384 chain rows with m = 75 > 64 columns
    -> k_factor_bigmfma<5,3,24> takes its SECOND pass over the chain (spamtree_amd/csrc/factor_big.hpp, `pass == 1`);
    the leaf level below has P = 525 -> k_factor_lchain<136> (K in registers, the chain factor streamed twice; default), or
    with SPAMTREE_LCHAIN=0 k_factor_bigmfma<3,5,34> (m <= 48) / <4,5,34> (m <= 64) / k_factor_wide.
  * #5 (q = 3, cell_size = 9: 27-row blocks): levels with chains of 216 and 243 rows exceed k_factor_quad's 200-row register
    budget and take k_factor_mfma (chains 201-256).
Reference: /root/reference/src/spamtree_model.cpp:880-922 (reference child branch), :923-963 (non-reference rows).
Each test asserts through st_level_info that the branch it means to reach is the one the library dispatches.
"""
import numpy as np
import pytest

from tests.test_gpu_parity import REL, hip_model, relerr
from tests.util import make_problem, oracle_model, strip_coords

pytestmark = pytest.mark.gpu


def compare_all(pb, expect):
    rng = np.random.default_rng(2)
    w0 = rng.standard_normal(pb["n"])
    om = oracle_model(pb, w=w0, tausq=0.2)
    hm = hip_model(pb, w=w0, tausq=0.2)
    info = hm.level_info()
    expect(info)
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
    for it in range(3):
        z = rng.standard_normal(pb["n"])
        om.gibbs_sample_w(z)
        hm.deal_with_w(z)
        assert relerr(hm.get_w()[om.na_ix_all], om.w[om.na_ix_all]) <= REL, it
    om.get_loglik_w(om.param_data)
    assert abs(hm.get_loglik_w(0) - om.param_data.loglik_w) <= REL * abs(om.param_data.loglik_w)
    hm.close()


@pytest.mark.parametrize("wide,lchain", [("0", "0"), ("2", "0"), ("0", "1")])
@pytest.mark.parametrize("nx,leaf_inst", [(370, "<3,5,34>"), (400, "<4,5,34>")])
def test_config4_chains_bigmfma_second_pass(nx, leaf_inst, wide, lchain, monkeypatch):
    """wide = "2": the sibling-group kernel k_factor_wide forced onto every eligible level (by default only big non-reference
    levels take it) -- chains of 450 / 525 rows take 4 / 5 passes of eight chain tiles; wide = "0": k_factor_bigmfma, one
    block per workgroup, second pass beyond 384 rows."""
    monkeypatch.setenv("SPAMTREE_WIDE", wide)
    monkeypatch.setenv("SPAMTREE_LCHAIN", lchain)   # "1" (the default): the leaf level on k_factor_lchain
    monkeypatch.setenv("SPAMTREE_LCHAIN_REF", "0")  # the reference levels stay on the older kernels here (the default route: the test below)
    coords, mv = strip_coords(nx, 10, 3)
    pb = make_problem(coords=coords, mv_id=mv, q=3, seed=3, K=(2, 1), tree_depth=7)

    def expect(info):
        older = "k_factor_wide" if wide == "2" else "k_factor_bigmfma"
        assert len(info) == 8 and all(L["kernel"] == older for L in info[1:7])
        assert info[7]["kernel"] == ("k_factor_lchain" if lchain == "1" else older)
        # level 7 (index 6): the second pass needs more than 64 columns and more than 16 * 24 = 384 chain rows
        assert info[6]["max_m"] > 64 and info[6]["max_P"] == 450 and info[6]["max_P"] > 384
        assert info[7]["max_P"] == 525
        assert (32 < info[7]["max_m"] <= 48) if leaf_inst == "<3,5,34>" else (48 < info[7]["max_m"] <= 64)

    compare_all(pb, expect)


def test_config5_chains_201_to_256_on_k_factor_mfma():
    coords, mv = strip_coords(900, 6, 3)
    pb = make_problem(coords=coords, mv_id=mv, q=3, seed=4, K=(2, 1), cell_size=9, tree_depth=9)

    def expect(info):
        assert len(info) == 10
        assert info[8]["kernel"] == "k_factor_mfma" and info[8]["max_P"] == 216      # config #5's level 9
        assert info[9]["kernel"] == "k_factor_mfma" and info[9]["max_P"] == 243      # ... and its leaf level
        assert info[9]["max_P"] > 200 and info[8]["max_P"] > 200                      # beyond k_factor_quad's register budget

    compare_all(pb, expect)


@pytest.mark.parametrize("nx", [370, 400])
def test_config4_reference_levels_on_lchain_and_ref_finish(nx, monkeypatch):
    """Reference levels of a wide-block tree as k_factor_lchain (chain pass, columns treated as conditionally independent)
    + k_factor_ref_finish (Schur complement from the V scratch, blocked factorisation, -Ri T in place): the default route of
    every reference level behind a chain (SPAMTREE_LCHAIN_REF=0 / SPAMTREE_LCHAIN_REF_MIN: off / only levels of that many blocks)."""
    coords, mv = strip_coords(nx, 10, 3)
    pb = make_problem(coords=coords, mv_id=mv, q=3, seed=3, K=(2, 1), tree_depth=7)

    def expect(info):
        assert len(info) == 8 and all(L["kernel"] == "k_factor_lchain+ref_finish" for L in info[1:7])
        assert info[7]["kernel"] == "k_factor_lchain"
        assert info[6]["max_m"] > 64 and info[6]["max_P"] == 450

    compare_all(pb, expect)
