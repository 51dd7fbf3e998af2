"""GPU: whole chains.  The C++ host driver (spamtree_amd/csrc/spamtree_fit.cpp through spamtree_amd/fit.py) and the
Python host driver (spamtree_amd/mcmc.py) against the oracle's restatement of spamtree_mv_mcmc with the same Philox
streams: theta / beta / tausq traces, saved w and yhat."""
import numpy as np
import pytest

from tests.util import make_problem

pytestmark = pytest.mark.gpu


def args_of(pb, k):
    return (pb["y"], pb["X"], pb["Z"], pb["coords"], pb["mv_id"], pb["blocking"], pb["gix_block"], pb["res_is_ref"],
            pb["parents"], pb["children"], pb.get("limited_tree", False), pb["block_names"], pb["block_groups"], pb["indexing"], pb["bounds"],
            np.zeros((pb["n"], 1)), pb["theta"], np.zeros(pb["p"]), 0.1, 0.01 * np.eye(k))


def relerr(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(1e-300, np.abs(np.asarray(b)).max()))


@pytest.mark.parametrize("case", [dict(side=25, q=1, seed=11, missing=0.1), dict(side=12, q=2, seed=12),
                                  dict(side=25, q=1, seed=13, missing=0.1, limited_tree=True)])
def test_cpp_and_python_drivers_match_oracle_chain(case):
    from oracle import spamtree_oracle as so
    from spamtree_amd import fit, mcmc
    pb = make_problem(**case)
    k = pb["theta"].size
    kw = dict(mcmc_keep=4, mcmc_burn=58, mcmc_thin=2, adapting=True, seed=99, main_verbose=False)   # crosses g0 = 50
    ref = so.spamtree_mv_mcmc(*args_of(pb, k), **kw)
    for drv in (fit.spamtree_mv_mcmc, mcmc.spamtree_mv_mcmc):
        got = drv(*args_of(pb, k), **kw)
        assert "None" not in got
        assert relerr(got["theta_mcmc"], ref["theta_mcmc"]) < 1e-8
        assert relerr(got["tausq_mcmc"], ref["tausq_mcmc"]) < 1e-8
        assert relerr(got["beta_mcmc"], ref["beta_mcmc"]) < 1e-8
        assert relerr(got["paramsd"], ref["paramsd"]) < 1e-7
        for i in range(4):
            assert relerr(np.asarray(got["w_mcmc"][i]).reshape(-1), ref["w_mcmc"][i]) < 1e-8
            assert relerr(np.asarray(got["yhat_mcmc"][i]).reshape(-1), ref["yhat_mcmc"][i]) < 1e-8


def test_cpp_chain_steps_and_reports_state():
    from spamtree_amd import fit
    pb = make_problem(side=25, q=1, seed=3)
    ch = fit.Chain(pb["y"], pb["X"], pb["Z"], pb["coords"], pb["mv_id"], pb["blocking"], pb["gix_block"], pb["res_is_ref"],
                   pb["parents"], pb["children"], pb.get("limited_tree", False), pb["block_names"], pb["block_groups"], pb["indexing"],
                   pb["bounds"], pb["theta"], np.zeros(pb["p"]), 0.1, 0.01 * np.eye(4), seed=5)
    ch.step(20)
    st = ch.state()
    assert st["iteration"] == 20 and np.all(np.isfinite(st["theta"])) and np.all(st["tausq_inv"] > 0)
    assert np.isfinite(st["loglik"]) and np.all(np.isfinite(ch.get_w()))
    ch.close()


def test_native_rccl_protocol_single_rank(monkeypatch):
    """The library's own RCCL path (st_comm_init: pack -> ncclAllReduce on the launch stream -> deterministic finish) with a
    one-rank communicator gives the chain of the plain single-GPU path bit for bit.  (More than one rank per GPU is not
    possible with RCCL; the multi-rank protocol itself is covered through gloo in tests/test_gpu_sharded.py.)"""
    from spamtree_amd import fit
    pb = make_problem(side=40, q=1, seed=3, missing=0.05)
    states = []
    monkeypatch.setenv("SPAMTREE_QUAD_MIN", "1")        # quad levels below, so that there are top levels to run ahead
    monkeypatch.setenv("SPAMTREE_QUAD_UNITS", "4")
    for uid in (None, fit.make_unique_id()):
        # the communicator run also starts phase A of the top levels ahead of time, as sharded runs do by default
        monkeypatch.setenv("SPAMTREE_ASYNC_TOP", "0" if uid is None else "1")
        ch = fit.Chain(pb["y"], pb["X"], pb["Z"], pb["coords"], pb["mv_id"], pb["blocking"], pb["gix_block"], pb["res_is_ref"],
                       pb["parents"], pb["children"], False, pb["block_names"], pb["block_groups"], pb["indexing"],
                       pb["bounds"], pb["theta"], np.zeros(pb["p"]), 0.1, 0.01 * np.eye(4), seed=5, unique_id=uid)
        ch.step(12)
        st = ch.state()
        states.append((st["theta"].copy(), st["tausq_inv"].copy(), float(st["loglik"]), ch.get_w().copy()))
        ch.close()
    a, b = states
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1]) and a[2] == b[2] and np.array_equal(a[3], b[3])


@pytest.mark.parametrize("async_top", ["1", "0"])
def test_cpp_driver_with_top_levels_ahead_of_time(async_top, monkeypatch):
    """st_factor_begin: the C++ driver starts phase A of the top levels for the proposal before the sweep, on a second
    stream (SPAMTREE_QUAD_MIN=1 gives this small tree quad levels below, hence top levels to run ahead).  Chains with and
    without it (SPAMTREE_ASYNC_TOP=0) equal the oracle's."""
    from oracle import spamtree_oracle as so
    from spamtree_amd import fit
    monkeypatch.setenv("SPAMTREE_QUAD_MIN", "1")
    monkeypatch.setenv("SPAMTREE_QUAD_UNITS", "4")
    monkeypatch.setenv("SPAMTREE_ASYNC_TOP", async_top)
    pb = make_problem(side=40, q=1, seed=21, missing=0.05)
    k = pb["theta"].size
    kw = dict(mcmc_keep=3, mcmc_burn=40, mcmc_thin=1, adapting=True, seed=7, main_verbose=False)
    ref = so.spamtree_mv_mcmc(*args_of(pb, k), **kw)
    got = fit.spamtree_mv_mcmc(*args_of(pb, k), **kw)
    assert relerr(got["theta_mcmc"], ref["theta_mcmc"]) < 1e-8
    assert relerr(got["tausq_mcmc"], ref["tausq_mcmc"]) < 1e-8 and relerr(got["beta_mcmc"], ref["beta_mcmc"]) < 1e-8
    for i in range(3):
        assert relerr(np.asarray(got["w_mcmc"][i]).reshape(-1), ref["w_mcmc"][i]) < 1e-8


def test_deferred_sweep_protocol_failure_and_misuse():
    """st_sample_w_loglik_begin / _end (the C++ driver's default).  A sweep that fails (negative tausq^-1: codes
    10 / 11, spamtree_model.cpp:1056, 1135) surfaces from _end AFTER the proposal's st_factor has run in between; a second
    _begin without _end is refused; the handle is usable afterwards and gives what the synchronous call gives."""
    import ctypes as C
    from tests.test_gpu_parity import hip_model
    pb = make_problem(side=25, q=1, seed=21)
    hm = hip_model(pb, tausq=0.2)
    lib, h = hm.lib, hm.h
    dp = C.POINTER(C.c_double)
    th = np.ascontiguousarray(pb["theta"], dtype=np.float64)
    assert hm.get_loglik_comps_w(0)
    z = np.random.default_rng(0).standard_normal(pb["n"])
    # reference: the synchronous pair (after one warm-up sweep: the first sweep after a factorisation also rebuilds the
    # records' Gram parts and takes other kernels, equal to 1e-12 but not bit for bit)
    ll_sync = C.c_double()
    assert lib.st_sample_w_loglik(h, z.ctypes.data_as(dp), 0, 0, 0, C.byref(ll_sync)) == 0
    hm.set_w(np.zeros(pb["n"]))
    assert lib.st_sample_w_loglik(h, z.ctypes.data_as(dp), 0, 0, 0, C.byref(ll_sync)) == 0
    w_sync = hm.get_w().copy()
    # deferred: the same sweep from the same state, the proposal's phase A in between
    hm.set_w(np.zeros(pb["n"]))
    assert lib.st_sample_w_loglik_begin(h, z.ctypes.data_as(dp), 0, 0, 0) == 0
    assert lib.st_sample_w_loglik_begin(h, z.ctypes.data_as(dp), 0, 0, 0) < 0                  # a second _begin without _end: usage error
    ll_f = C.c_double()
    assert lib.st_factor(h, 1, th.ctypes.data_as(dp), th.size, C.byref(ll_f)) == 0
    ll_def = C.c_double()
    assert lib.st_sample_w_loglik_end(h, C.byref(ll_def)) == 0
    assert np.array_equal(hm.get_w(), w_sync) and ll_def.value == ll_sync.value
    assert lib.st_sample_w_loglik_end(h, C.byref(ll_def)) < 0                                 # _end without _begin
    # a failing sweep: negative tausq^-1 makes the posterior precision of every block indefinite
    bad = np.array([-50.0])
    assert lib.st_set_tausq_inv(h, bad.ctypes.data_as(dp)) == 0
    assert lib.st_sample_w_loglik_begin(h, z.ctypes.data_as(dp), 0, 1, 0) == 0
    assert lib.st_factor(h, 1, th.ctypes.data_as(dp), th.size, C.byref(ll_f)) == 0
    assert lib.st_sample_w_loglik_end(h, C.byref(ll_def)) in (10, 11)
    # the handle is not poisoned: restore tausq and w, the synchronous sweep reproduces the reference
    good = np.array([1.0 / 0.2])
    assert lib.st_set_tausq_inv(h, good.ctypes.data_as(dp)) == 0
    hm.set_w(np.zeros(pb["n"]))
    ll3 = C.c_double()
    assert lib.st_sample_w_loglik(h, z.ctypes.data_as(dp), 0, 0, 0, C.byref(ll3)) == 0
    assert np.array_equal(hm.get_w(), w_sync) and ll3.value == ll_sync.value
    hm.close()


def test_chain_with_and_without_deferred_sync_are_identical():
    """SPAMTREE_DEFER_SYNC and SPAMTREE_EARLY_BETA (the tausq / beta draws made under the proposal's factorisation:
    st_factor_enqueue / st_factor_finish) are read once per process, so the settings run in subprocesses: same chain bit for bit."""
    import os
    import subprocess
    import sys
    code = ("import numpy as np, sys; sys.path.insert(0, %r)\n"
            "from tests.util import make_problem\nfrom spamtree_amd import fit\n"
            "pb = make_problem(side=25, q=1, seed=3, missing=0.1)\n"
            "ch = fit.Chain(pb['y'], pb['X'], pb['Z'], pb['coords'], pb['mv_id'], pb['blocking'], pb['gix_block'], pb['res_is_ref'],"
            " pb['parents'], pb['children'], False, pb['block_names'], pb['block_groups'], pb['indexing'], pb['bounds'], pb['theta'],"
            " np.zeros(pb['p']), 0.1, 0.01 * np.eye(4), seed=5)\n"
            "ch.step(40); st = ch.state()\n"
            "print(repr((st['theta'].tolist(), st['tausq_inv'].tolist(), st['loglik'], float(np.sum(ch.get_w())))))\n") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for defer, early in (("1", "1"), ("0", "1"), ("1", "0")):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SPAMTREE_DEFER_SYNC=defer, SPAMTREE_EARLY_BETA=early),
                           capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(r.stdout.strip().splitlines()[-1])
    assert outs[0] == outs[1] == outs[2]


def test_factor_in_two_halves_equals_st_factor():
    """st_factor_enqueue / st_factor_finish (include/spamtree_hip.h): the same return code and log-density as st_factor, with the
    setters -- which no longer synchronise the host -- called in between; misuse is refused."""
    import ctypes as C
    from tests.test_gpu_parity import hip_model
    pb = make_problem(side=25, q=1, seed=4, missing=0.1)
    rng = np.random.default_rng(1)
    w0 = rng.standard_normal(pb["n"])
    hm = hip_model(pb, w=w0, tausq=0.2)
    assert hm.get_loglik_comps_w(0)
    lib, h = hm.lib, hm.h
    dp = C.POINTER(C.c_double)
    th = np.ascontiguousarray(pb["theta"] * 1.1)
    ll0, ll1 = C.c_double(), C.c_double()
    assert lib.st_factor(h, 1, th.ctypes.data_as(dp), th.size, C.byref(ll0)) == 0
    assert lib.st_factor_finish(h, C.byref(ll1)) < 0                      # nothing enqueued
    assert lib.st_factor_enqueue(h, 1, th.ctypes.data_as(dp), th.size) == 0
    assert lib.st_factor_enqueue(h, 1, th.ctypes.data_as(dp), th.size) < 0   # a second one before the first is finished
    t2 = np.array([1.0 / 0.3]); b2 = np.full(pb["p"], 0.25)
    assert lib.st_set_tausq_inv(h, t2.ctypes.data_as(dp)) == 0 and lib.st_set_beta(h, b2.ctypes.data_as(dp)) == 0
    t2[:] = -1.0; b2[:] = 99.0                                               # the caller's buffers are free on return
    assert lib.st_factor_finish(h, C.byref(ll1)) == 0 and ll1.value == ll0.value
    bad = th.copy(); bad[0] = -1.0                                           # a failing proposal: the reference's errtype through _finish
    assert lib.st_factor_enqueue(h, 1, bad.ctypes.data_as(dp), bad.size) == 0
    assert lib.st_factor_finish(h, C.byref(ll1)) in (1, 2, 3)
    # the uploads arrived: a sweep with the new tausq / beta equals one on a fresh handle that was given them up front
    z = rng.standard_normal(pb["n"])
    hm.deal_with_w(z)
    hr = hip_model(pb, w=w0, tausq=0.3, beta=np.full(pb["p"], 0.25))
    assert hr.get_loglik_comps_w(0)
    hr.deal_with_w(z)
    assert np.array_equal(hm.get_w(), hr.get_w())
    hm.close(); hr.close()
