"""GPU: 2, 3 and 4 processes share ONE problem on device 0 (gloo exchange); every result must equal the single-process
run bit for bit (ownership by subtree, exchanges are sums with zeros)."""
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("side,q,quad_min,env", [(120, 1, None, {}), (48, 3, None, {}), (120, 1, "1", {}),
                                                 (48, 3, None, {"SPAMTREE_WIDE": "2"}),          # sibling-group kernel on every wide level
                                                 (40, 3, None, {"LCHAIN": "expected"}),           # 50-row leaves behind 225-row chains: k_factor_lchain
                                                                                                 # (slabs cut inside a rank's run) + the one-wave LDS solve
                                                 (120, 1, "1", {"SPAMTREE_SAMPLE_WAVE": "2"})])  # one block per wave in the sweep
def test_sharded_equals_single_process_bitwise(side, q, quad_min, env, tmp_path, monkeypatch):
    """quad_min = "1": even these small levels take k_factor_quad (SPAMTREE_QUAD_MIN, inherited by the spawned ranks),
    whose quads are cut at ownership boundaries and sized by the rank's share of the level (here forced: 4 units per
    workgroup in the single process, 2 with two ranks, 1 with three) -- the results must not depend on how they are cut."""
    if quad_min:
        monkeypatch.setenv("SPAMTREE_QUAD_MIN", quad_min)
    env = dict(env)
    expect_lchain = env.pop("LCHAIN", None) is not None
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    units = {1: "4", 2: "2", 3: "1", 4: None}     # None: the library's own choice for the rank's share of a level
    import torch.multiprocessing as mp
    from tests._sharded_worker import gpu_worker
    steps = 2
    for world in (1, 2, 3, 4):
        if units[world] is None:
            monkeypatch.delenv("SPAMTREE_QUAD_UNITS", raising=False)
        else:
            monkeypatch.setenv("SPAMTREE_QUAD_UNITS", units[world])
        mp.spawn(gpu_worker, args=(world, free_port(), side, q, str(tmp_path), steps), nprocs=world, join=True)
    ref = np.load(tmp_path / "res_1_0.npz")
    if expect_lchain:
        assert 6 in ref["kernels"].tolist() and 7 in ref["kernels"].tolist(), ref["kernels"]      # ST_KERNEL_LCHAIN / _LCHAIN_REF: the case reaches the kernels it is meant to cover
    for world in (2, 3, 4):
        rows = 0
        for rank in range(world):
            r = np.load(tmp_path / f"res_{world}_{rank}.npz")
            for k in ["ll_A", "ll_A2", "err", "ll_C0", "ll_C1"]:
                assert float(r[k]) == float(ref[k]), (world, rank, k)
            assert np.array_equal(r["w"], ref["w"]) and np.array_equal(r["xty"], ref["xty"]) and np.array_equal(r["ssq"], ref["ssq"])
            rows += int(r["owned_rows"])
        assert 0 < rows <= ref["w"].size


@pytest.mark.parametrize("side,q,quad_min", [(120, 1, None), (120, 1, "1"), (40, 2, None)])
def test_limited_tree_sharded_equals_single_process_bitwise(side, q, quad_min, tmp_path, monkeypatch):
    """limited_tree = TRUE (tree_dep.cpp:133-186: every block has ONE parent; spamtree_model.cpp:901-903, 1275-1278) on 2 and 3
    ranks: ownership follows the direct-parent chain up to the cut level, the cut-level records go to their single parents in
    the replicated top; every result equals the single-process run bit for bit (round 3; single GPU only before)."""
    if quad_min:
        monkeypatch.setenv("SPAMTREE_QUAD_MIN", quad_min)
    import torch.multiprocessing as mp
    from tests._sharded_worker import gpu_worker
    for world in (1, 2, 3):
        mp.spawn(gpu_worker, args=(world, free_port(), side, q, str(tmp_path), 2, True), nprocs=world, join=True)
    ref = np.load(tmp_path / "res_1_0.npz")
    for world in (2, 3):
        for rank in range(world):
            r = np.load(tmp_path / f"res_{world}_{rank}.npz")
            for k in ["ll_A", "ll_A2", "err", "ll_C0", "ll_C1"]:
                assert float(r[k]) == float(ref[k]), (world, rank, k)
            assert np.array_equal(r["w"], ref["w"]) and np.array_equal(r["xty"], ref["xty"]) and np.array_equal(r["ssq"], ref["ssq"])
