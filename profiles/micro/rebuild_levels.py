"""Diagnostic (not part of the product): per-level time of phase B for sweeps with cached Gram parts and for the first sweep
after an accepted theta (records rebuilt), HIP events around every launch.  python profiles/micro/rebuild_levels.py [side] [q]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from spamtree_amd.model import SpamTreeMV  # noqa: E402
from spamtree_amd.synthetic import make_workload  # noqa: E402

side = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
q = int(sys.argv[2]) if len(sys.argv) > 2 else 1
wl = make_workload(side, q=q)
hm = SpamTreeMV(wl["y"], wl["X"], wl["Z"], wl["coords"], wl["mv_id"], wl["blocking"], wl["gix_block"], wl["res_is_ref"],
                wl["parents"], wl["children"], False, wl["block_names"], wl["block_groups"], wl["indexing"],
                np.zeros(wl["n"]), np.zeros(wl["p"]), wl["theta"], 10.0, device=0)
rng = np.random.default_rng(0)
assert hm.get_loglik_comps_w(0)
z = rng.standard_normal(wl["n"])
hm.deal_with_w(z)
hm.profile(1)


def levels():
    import ctypes as C
    nl = C.c_int32(); ms = np.zeros(128); by = np.zeros(128)
    hm.lib.st_profile_levels(hm.h, C.byref(nl), ms.ctypes.data_as(C.POINTER(C.c_double)), by.ctypes.data_as(C.POINTER(C.c_double)), 128)
    k = nl.value
    return ms[k:2 * k].copy()


hm.synchronize(); hm.profile_get(); levels()
for _ in range(5):
    hm.deal_with_w(z)
hm.synchronize()
tot = hm.profile_get()["sample"]
cached = levels()
print("cached sweep : total %.3f ms  by level (mean per launch)" % (tot[0] / 5), np.round(cached, 4))
reb = []
tots = []
for k in range(3):
    hm.profile(0)
    hm.theta_update(1, wl["theta"] * (1.0 + 0.01 * (k + 1)))
    assert hm.get_loglik_comps_w(1)
    hm.accept_make_change()
    hm.profile(1); hm.profile_get(); levels()
    hm.deal_with_w(z)
    hm.synchronize()
    tots.append(hm.profile_get()["sample"][0])
    reb.append(levels())
print("rebuild sweep: total %.3f ms  by level (mean per launch; levels with a separate Gram launch count two launches)" % np.median(tots), np.round(np.median(np.array(reb), axis=0), 4))
hm.close()
