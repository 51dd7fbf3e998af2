"""Diagnostic (not part of the product): time of phase P (st_predict: draws at the blocks without observations,
/root/reference/src/spamtree_model.cpp:1234-1358) next to phase A of the leaf level.  python profiles/micro/predict_time.py [side] [q] [cell] [missing]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from spamtree_amd.model import SpamTreeMV  # noqa: E402
from spamtree_amd.synthetic import make_workload  # noqa: E402

side = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
q = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cell = int(sys.argv[3]) if len(sys.argv) > 3 else 25
missing = tuple(float(x) for x in sys.argv[4].split(",")) if len(sys.argv) > 4 else (0.1,) * q
wl = make_workload(side, q=q, cell_size=cell, missing=missing, device=0)
hm = SpamTreeMV(wl["y"], wl["X"], wl["Z"], wl["coords"], wl["mv_id"], wl["blocking"], wl["gix_block"], wl["res_is_ref"],
                wl["parents"], wl["children"], False, wl["block_names"], wl["block_groups"], wl["indexing"],
                np.zeros(wl["n"]), np.zeros(wl["p"]), wl["theta"], 10.0, device=0)
assert hm.get_loglik_comps_w(0)
hm.deal_with_w(None, seed=3, it=1)
n_na = int(np.sum(~np.isfinite(wl["y"])))
hm.profile(1)
hm.get_loglik_comps_w(1)
hm.synchronize(); lv = hm.profile_levels()[0]; hm.profile_get()
for changed in (True, False):
    ts = []
    for _ in range(5):
        hm.synchronize(); t0 = time.perf_counter()
        hm.predict(changed)
        hm.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    ev = hm.profile_get()["predict"]
    print(f"side {side} q {q} cell {cell} missing {missing}: n = {wl['n']}, {n_na} rows to predict; predict(theta_changed={changed}): "
          f"{np.median(ts):.3f} ms host-timed, {ev[0] / max(1, ev[1]):.3f} ms per launch by events ({ev[1]} launches); phase A by level {np.round(lv, 3)}")
hm.close()
