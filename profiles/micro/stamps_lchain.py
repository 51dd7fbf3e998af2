"""Diagnostic (not part of the product): where k_factor_lchain's time goes at config #4's shape (leaf level).
Build with -DFM_STAMPS into profiles/micro/libspamtree_hip_stamps.so (see stamps.py); run: python profiles/micro/stamps_lchain.py [side]"""
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

NAMES = {0: "topology + coords + LDS clear", 1: "covariance -> registers", 2: "phase 1: DMA wait + pad", 3: "phase 1: barrier", 4: "phase 1: DMA issue",
         5: "phase 1: V tile (MFMA)", 6: "r_j", 7: "phase 2: DMA wait + diag fix", 8: "phase 2: barrier", 9: "phase 2: DMA issue",
         10: "phase 2: T tile (MFMA) + stores", 11: "scalars",
         12: "LOADER phase 1: landing wait", 13: "LOADER phase 1: padding", 14: "LOADER phase 1: barrier wait", 15: "LOADER phase 1: requests (+ loop)"}
side = int(sys.argv[1]) if len(sys.argv) > 1 else 577
wl = make_workload(side, q=3)
hm = SpamTreeMV(wl["y"], wl["X"], wl["Z"], wl["coords"], wl["mv_id"], wl["blocking"], wl["gix_block"], wl["res_is_ref"],
                wl["parents"], wl["children"], False, wl["block_names"], wl["block_groups"], wl["indexing"],
                np.zeros(wl["n"]), np.zeros(wl["p"]), wl["theta"], 10.0, device=0)
lib = hm.lib
lib.st_debug_stamps_wide.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
lib.st_debug_stamp_level_wide.argtypes = [C.c_int]
hm.get_loglik_comps_w(0)
nlev = len(np.unique(np.asarray(wl["block_groups"])))
lev = nlev - 1
lib.st_debug_stamp_level_wide(lev)
buf = (C.c_ulonglong * 16)()
lib.st_debug_stamps_wide(buf, 1)
for _ in range(2):
    hm.get_loglik_comps_w(1)
lib.st_debug_stamps_wide(buf, 0)
v = np.array(list(buf), dtype=np.float64)
tot = v.sum()
print(f"level {lev}: total ticks {tot:.3e}")
for i in np.argsort(-v):
    if v[i] > 0:
        print(f"   {NAMES.get(int(i), str(i)):34s} {100 * v[i] / tot:5.1f} %")
hm.close()
