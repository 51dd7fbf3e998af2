"""GPU: the device parts of the tree builder (include/spamtree_tree.h; SURVEY.md section 8f-1 -- the quantile thresholds
`kthresholds`, /root/reference/src/tree_dep.cpp:16-27; one knot per fine cell, /root/reference/R/make_tree.R:84-92; the
same-margin nearest placed row for leftover and missing rows, make_tree.R:213-305, 317-413) give the host path's trees bit
for bit, and each primitive its brute-force answer."""
import ctypes as C

import numpy as np
import pytest

from spamtree_amd import _lib
from spamtree_amd import topology as tp

pytestmark = pytest.mark.gpu

FIELDS = ("sort_ix", "blocking", "gix_block", "res_is_ref", "parents_ptr", "parents_idx", "children_ptr", "children_idx",
          "block_names", "block_groups", "indexing_ptr", "indexing_idx", "parchi_map")


def test_device_sort_and_thresholds():
    lib = _lib.load()
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(100003), -rng.uniform(size=50), np.zeros(7), [-0.0]])
    out = np.empty_like(x)
    assert lib.st_tb_sort(x.ctypes.data_as(_lib.c_dp), x.size, 0, out.ctypes.data_as(_lib.c_dp)) == 0
    assert np.array_equal(out, np.sort(x))
    for k in (2, 7, 640):
        assert np.array_equal(tp.kthresholds(out, k, presorted=True), tp.kthresholds(x, k))


def test_device_cell_argmin_matches_host_rule():
    rng = np.random.default_rng(1)
    n, ncells = 200000, 30011
    code = rng.integers(0, ncells, size=n)
    key = rng.uniform(size=n).round(3)                       # many exact ties inside a cell
    ix = rng.permutation(n)
    assert np.array_equal(tp._cell_argmin(code, key, ix, ncells, 0), tp._cell_argmin(code, key, ix, ncells, None))


@pytest.mark.parametrize("grid", [True, False])
def test_device_nearest_matches_brute_force(grid):
    rng = np.random.default_rng(2)
    if grid:                                                  # a regular grid: exact distance ties everywhere
        g = np.stack(np.meshgrid(np.linspace(0, 1, 40), np.linspace(0, 1, 40), indexing="ij"), -1).reshape(-1, 2)
        pick = rng.permutation(1600)
        tc, qc = g[pick[:900]], g[pick[900:]]
    else:
        tc, qc = rng.uniform(size=(3000, 2)) * [1.0, 0.05], rng.uniform(size=(1500, 2)) * [1.0, 0.05]    # a thin strip
    tmv = rng.integers(0, 3, size=tc.shape[0]); tmv[tmv == 2] = 1          # margin 2 has no target: falls back to all margins
    qmv = rng.integers(0, 3, size=qc.shape[0])
    got = tp._nearest_rows(tc, tmv, qc, qmv, 3, True, 0)
    dx = qc[:, 0][:, None] - tc[:, 0][None, :]
    dy = qc[:, 1][:, None] - tc[:, 1][None, :]
    d2 = dx * dx + dy * dy
    ok = (tmv[None, :] == qmv[:, None]) | (qmv == 2)[:, None]
    d2 = np.where(ok, d2, np.inf)
    ref = np.argmax(d2 == d2.min(axis=1)[:, None], axis=1)                  # first (lowest) index attaining the minimum
    assert np.array_equal(got, ref)
    assert np.array_equal(tp._nearest_rows(tc, tmv, qc, qmv, 3, True, None), ref)      # the host path follows the same rule


def test_device_nearest_many_ties_and_collinear_targets():
    """(a) more exact ties than the host path's first candidate batch (12: four equidistant neighbours x three
    co-located margins under the all-margins fallback) -- both paths pick the lowest index; (b) collinear targets (a transect):
    the grid collapses to one row of cells and the ring walk must still stop early (the result is checked; the time bound is
    what a 4096-ring walk per query would blow)."""
    import time
    g = np.stack(np.meshgrid(np.arange(9.0), np.arange(9.0), indexing="ij"), -1).reshape(-1, 2)
    tc = np.tile(g, (3, 1)); tmv = np.repeat([0, 1, 2], g.shape[0])
    rng = np.random.default_rng(3)
    perm = rng.permutation(tc.shape[0]); tc, tmv = tc[perm], tmv[perm]
    qc = g[(g[:, 0] % 2 == 1) & (g[:, 1] % 2 == 1)] + 0.0
    keep = ~((tc[:, None, :] == qc[None, :, :]).all(-1).any(1))            # the query points themselves are not targets
    tc, tmv = tc[keep], tmv[keep]
    qmv = np.full(qc.shape[0], 3)                                         # margin 3 has no target
    d2 = ((qc[:, None, :] - tc[None, :, :]) ** 2).sum(-1)
    ref = np.argmax(d2 == d2.min(axis=1)[:, None], axis=1)
    assert (d2 == d2.min(axis=1)[:, None]).sum(axis=1).min() >= 12
    assert np.array_equal(tp._nearest_rows(tc, tmv, qc, qmv, 4, True, 0), ref)
    assert np.array_equal(tp._nearest_rows(tc, tmv, qc, qmv, 4, True, None), ref)
    # (b) a transect: 200 000 targets on the line y = 0.3, 50 000 queries off it
    tx = rng.uniform(size=200000); tc = np.stack([tx, np.full_like(tx, 0.3)], 1)
    qc = rng.uniform(size=(50000, 2))
    z = np.zeros(tc.shape[0], dtype=np.int64)
    t0 = time.time()
    got = tp._nearest_rows(tc, z, qc, np.zeros(qc.shape[0], dtype=np.int64), 1, True, 0)
    dt = time.time() - t0
    assert dt < 20.0, dt
    for lo in range(0, 600, 100):      # brute force on a sample (rounding makes many d2 ties here: dy^2 swamps small dx^2 differences)
        q = qc[lo:lo + 100]
        dx = q[:, 0][:, None] - tc[:, 0][None, :]; dy = q[:, 1][:, None] - tc[:, 1][None, :]
        d2 = dx * dx + dy * dy
        ref = np.argmax(d2 == d2.min(axis=1)[:, None], axis=1)
        assert np.array_equal(got[lo:lo + 100], ref)


@pytest.mark.parametrize("case", [dict(side=25, q=1), dict(side=14, q=3, missing=0.2), dict(side=30, q=1, missing=0.1, cell_size=9, K=(3, 2)),
                                  dict(side=30, q=1, tree_depth=2), dict(side=36, q=1, missing=0.05, cell_size=16, random=True),
                                  dict(side=18, q=3, missing=0.25, cell_size=9, mvbias=1.5), dict(side=120, q=2, missing=0.3, random=True),
                                  dict(side=20, q=2, missing=0.1, limited_tree=True)])
def test_device_tree_equals_host_tree(case):
    case = dict(case)
    side, q, missing, random = case.pop("side"), case.pop("q"), case.pop("missing", 0.0), case.pop("random", False)
    rng = np.random.default_rng(5)
    if random:
        base = rng.uniform(size=(side * side, 2))
        coords, mv = np.tile(base, (q, 1)), np.repeat(np.arange(1, q + 1), side * side)
    else:
        coords, mv = tp.grid_coords(side, q)
    y = rng.standard_normal(coords.shape[0])
    y[rng.uniform(size=y.size) < missing] = np.nan
    host = tp.prepare(y, coords, mv, **case)
    dev = tp.prepare(y, coords, mv, device=0, **case)
    for f in FIELDS:
        assert np.array_equal(getattr(host, f), getattr(dev, f)), f


def test_device_tree_feeds_the_hot_path():
    """A tree built with the device steps goes through st_create and phase A like any other."""
    from tests.test_gpu_parity import hip_model
    from tests.util import make_problem, oracle_model
    pb = make_problem(side=30, q=2, seed=4, missing=0.15, device=0)
    om, hm = oracle_model(pb), hip_model(pb)
    assert om.get_loglik_comps_w(om.param_data) and hm.get_loglik_comps_w(0)
    assert abs(hm.loglik_w[0] - om.param_data.loglik_w) <= 1e-9 * abs(om.param_data.loglik_w)
    hm.close()
