"""Diagnostic (not part of the product): where k_factor_ref_finish's time goes at config #4's shape (reference levels 5-6).
Build with -DFM_STAMPS into profiles/micro/libspamtree_hip_stamps.so (see stamps.py); run: python profiles/micro/stamps_ref_finish.py [side]"""
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

NAMES = {0: "topology + coords + 1 / r", 2: "Schur: V chunks -> LDS, MFMA", 3: "cholesky + inverse",
         4: "N = -Ri T in place (MFMA) + N w_pa", 5: "Ri out + scalars"}
side = int(sys.argv[1]) if len(sys.argv) > 1 else 577
wl = make_workload(side, q=3)
hm = SpamTreeMV(wl["y"], wl["X"], wl["Z"], wl["coords"], wl["mv_id"], wl["blocking"], wl["gix_block"], wl["res_is_ref"],
                wl["parents"], wl["children"], False, wl["block_names"], wl["block_groups"], wl["indexing"],
                np.zeros(wl["n"]), np.zeros(wl["p"]), wl["theta"], 10.0, device=0)
lib = hm.lib
lib.st_debug_stamps_wide.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
lib.st_debug_stamp_level_wide.argtypes = [C.c_int]
hm.get_loglik_comps_w(0)
info = hm.level_info()
for lev, L in enumerate(info):
    if L["kernel"] != "k_factor_lchain+ref_finish":
        continue
    lib.st_debug_stamp_level_wide(lev)
    buf = (C.c_ulonglong * 16)()
    lib.st_debug_stamps_wide(buf, 1)
    for _ in range(2):
        hm.get_loglik_comps_w(1)
    lib.st_debug_stamps_wide(buf, 0)
    v = np.array(list(buf), dtype=np.float64)
    tot = v.sum()
    if tot == 0:
        continue
    print(f"level {lev} ({L['n_blocks']} blocks, P = {L['max_P']}): total ticks {tot:.3e}")
    for i in np.argsort(-v):
        if v[i] > 0:
            print(f"   {NAMES.get(int(i), str(i)):40s} {100 * v[i] / tot:5.1f} %")
hm.close()
