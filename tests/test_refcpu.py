"""CPU-only: the OpenMP restatement (oracle/refcpu.cpp, the timed CPU baseline) against the NumPy restatement."""
import numpy as np
import pytest

from oracle.refcpu import RefCpu
from tests.util import make_problem, oracle_model


@pytest.mark.parametrize("case", [dict(side=25, q=1), dict(side=24, q=1, missing=0.15), dict(side=14, q=3, missing=0.1),
                                  dict(side=16, q=2)])
def test_refcpu_matches_numpy_oracle(case):
    pb = make_problem(seed=23, **case)
    rng = np.random.default_rng(2)
    w0 = rng.standard_normal(pb["n"])
    beta = np.array([0.3, -0.2, 0.1])
    om = oracle_model(pb, w=w0, beta=beta, tausq=0.25)
    rc = RefCpu(pb["y"], pb["X"], pb["coords"], pb["mv_id"], pb["res_is_ref"], pb["parents"], pb["children"],
                pb["block_names"], pb["block_groups"], pb["indexing"], threads=4)
    rc.set_w(w0)
    rc.set_beta(np.tile(beta[:, None], (1, pb["q"])))
    rc.set_tausq_inv(4.0)
    assert om.get_loglik_comps_w(om.param_data)
    code, ll = rc.factor(0, pb["theta"])
    assert code == 0 and abs(ll - om.param_data.loglik_w) < 1e-10 * abs(ll)
    ld, lc = rc.comps(0)
    assert np.abs(ld - om.param_data.logdetCi_comps).max() < 1e-9
    for it in range(3):
        z = rng.standard_normal(pb["n"])
        om.gibbs_sample_w(z)
        assert rc.sample_w(z) == 0
        obs = om.na_ix_all
        assert np.abs(rc.get_w()[obs] - om.w[obs]).max() < 1e-10 * np.abs(om.w).max()
        om.get_loglik_w(om.param_data)
        assert abs(rc.loglik_w(0) - om.param_data.loglik_w) < 1e-10 * abs(om.param_data.loglik_w)
        xty, ssq = rc.stats()
        oxty, ossq = om.beta_tausq_stats()
        assert np.abs(xty - oxty).max() < 1e-10 * np.abs(oxty).max() and np.abs(ssq - ossq).max() < 1e-10 * ossq.max()
    if pb["q"] == 1:                      # negative sigma^2: the root Cholesky fails (errtype 1)
        th = pb["theta"].copy()
        th[0] = -1.0
        assert rc.factor(1, th)[0] == 1
    rc.close()
