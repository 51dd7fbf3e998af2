"""Diagnostic (not part of the product): where the generic phase-B kernel k_sample<true> spends its time at config #4's shape,
per tree level.  Build with -DFM_STAMPS into profiles/micro/libspamtree_hip_stamps.so (see stamps.py); run: python profiles/micro/stamps_sample_big.py [side]"""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from spamtree_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.join(ROOT, "profiles", "micro", "libspamtree_hip_stamps.so")
from spamtree_amd.model import SpamTreeMV  # noqa: E402
from spamtree_amd.synthetic import make_workload  # noqa: E402

NAMES = {7: "wait for the workgroup (loop top)", 0: "topology + w_pa + N w_pa", 1: "S and Smu (cached Ri'Ri + children)", 2: "Cholesky solve (one wave, LDS)",
         3: "w out + ev", 4: "segment sums", 5: "column pass + records"}
side = int(sys.argv[1]) if len(sys.argv) > 1 else 577
wl = make_workload(side, q=3)
hm = SpamTreeMV(wl["y"], wl["X"], wl["Z"], wl["coords"], wl["mv_id"], wl["blocking"], wl["gix_block"], wl["res_is_ref"],
                wl["parents"], wl["children"], False, wl["block_names"], wl["block_groups"], wl["indexing"],
                np.zeros(wl["n"]), np.zeros(wl["p"]), wl["theta"], 10.0, device=0)
lib = hm.lib
lib.st_debug_stamps_sample.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
lib.st_debug_stamp_level_sample.argtypes = [C.c_int]
hm.get_loglik_comps_w(0)
rng = np.random.default_rng(0)
hm.deal_with_w(rng.standard_normal(wl["n"]))       # builds the cached Gram parts
nlev = len(np.unique(np.asarray(wl["block_groups"])))
for lev in (2, nlev - 3, nlev - 2, nlev - 1):
    lib.st_debug_stamp_level_sample(lev)
    buf = (C.c_ulonglong * 16)()
    lib.st_debug_stamps_sample(buf, 1)
    for _ in range(2):
        hm.deal_with_w(rng.standard_normal(wl["n"]))
    lib.st_debug_stamps_sample(buf, 0)
    v = np.array(list(buf), dtype=np.float64)
    tot = v.sum()
    if tot == 0:
        continue
    print(f"level {lev}: total ticks {tot:.3e}")
    for i in np.argsort(-v):
        if v[i] > 0:
            print(f"   {NAMES.get(int(i), str(i)):40s} {100 * v[i] / tot:5.1f} %")
hm.close()
