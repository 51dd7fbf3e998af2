"""Diagnostic (not part of the product): cost of a sweep whose message records' Gram parts are cached (theta unchanged)
against the first sweep after an accepted proposal (they are rebuilt), on one GPU.
Run on the GPU box:  python profiles/micro/rebuild_sweep.py [side] [q]      (SPAMTREE_GRAM_BIG=0: the generic kernel's own rebuild)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from spamtree_amd.model import SpamTreeMV  # noqa: E402
from spamtree_amd.synthetic import make_workload  # noqa: E402

side = int(sys.argv[1]) if len(sys.argv) > 1 else 577
q = int(sys.argv[2]) if len(sys.argv) > 2 else 3
wl = make_workload(side, q=q)
hm = SpamTreeMV(wl["y"], wl["X"], wl["Z"], wl["coords"], wl["mv_id"], wl["blocking"], wl["gix_block"], wl["res_is_ref"],
                wl["parents"], wl["children"], False, wl["block_names"], wl["block_groups"], wl["indexing"],
                np.zeros(wl["n"]), np.zeros(wl["p"]), wl["theta"], 10.0, device=0)
rng = np.random.default_rng(0)
assert hm.get_loglik_comps_w(0)
z = rng.standard_normal(wl["n"])


_it = [0]


def sweep():
    """one sweep with device-generated normals (Philox stream of the iteration, as the C++ driver runs it): the round-2 version
    uploaded 8 MB of host normals per sweep, +0.5 ms of PCIe time that is not part of a chain's sweep"""
    hm.synchronize()
    _it[0] += 1
    t0 = time.perf_counter()
    if os.environ.get("REBUILD_SWEEP_HOST_Z") == "1":
        hm.deal_with_w(z)
    else:
        hm.deal_with_w(None, seed=7, it=_it[0])
    hm.synchronize()
    return (time.perf_counter() - t0) * 1e3


first = sweep()                       # Gram parts built for the first time
cached = [sweep() for _ in range(5)]
rebuild = []
for k in range(3):
    th = wl["theta"] * (1.0 + 0.01 * (k + 1))
    hm.theta_update(1, th)
    assert hm.get_loglik_comps_w(1)
    hm.accept_make_change()           # the proposal's panels become the accepted ones: the cached Gram parts are stale
    rebuild.append(sweep())
    cached.append(sweep())
print(f"side {side} q {q}: first sweep {first:.2f} ms; cached sweeps {np.median(cached):.2f} ms (median of {len(cached)}); "
      f"sweeps after an accepted theta {np.median(rebuild):.2f} ms (median of {len(rebuild)}), GRAM_BIG={os.environ.get('SPAMTREE_GRAM_BIG', '1')}")
hm.close()
