"""Diagnostic (not part of the product): where k_factor_quad's time goes, per tree level.
Build the stamped library first:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DFM_STAMPS -I include
  -I spamtree_amd/csrc -o profiles/micro/libspamtree_hip_stamps.so spamtree_amd/csrc/*.cpp spamtree_amd/csrc/*.hip -lrccl   (per-family accessors: st_debug_stamps = k_factor_quad, _mfma, _wide, _sample, _generic)
Run on the GPU box:  python profiles/micro/stamps.py [side]"""
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

NAMES = {0: "topology+coords", 1: "covariance", 2: "private ancestor step", 3: "panel barrier wait", 4: "panel compute (MFMA)",
         5: "DMA wait + pad", 6: "DMA issue", 7: "cholesky", 8: "N = -Ri T + store", 9: "final barrier", 10: "hv", 11: "R / leaf outputs",
         12: "Ri out + e2", 13: "scalars"}

side = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
q = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cell = int(sys.argv[3]) if len(sys.argv) > 3 else 25
missing = tuple(float(x) for x in sys.argv[4].split(",")) if len(sys.argv) > 4 else None
wl = make_workload(side, q=q, cell_size=cell, missing=missing, device=0)
hm = SpamTreeMV(wl["y"], wl["X"], wl["Z"], wl["coords"], wl["mv_id"], wl["blocking"], wl["gix_block"], wl["res_is_ref"],
                wl["parents"], wl["children"], False, wl["block_names"], wl["block_groups"], wl["indexing"],
                np.zeros(wl["n"]), np.zeros(wl["p"]), wl["theta"], 10.0, device=0)
lib = hm.lib
lib.st_debug_stamps.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
lib.st_debug_stamp_level.argtypes = [C.c_int]
hm.get_loglik_comps_w(0)
nlev = len(np.unique(np.asarray(wl["block_groups"])))
for lev in range(max(0, nlev - 4), nlev):
    lib.st_debug_stamp_level(lev)
    buf = (C.c_ulonglong * 16)()
    lib.st_debug_stamps(buf, 1)
    for _ in range(3):
        hm.get_loglik_comps_w(1)
    lib.st_debug_stamps(buf, 0)
    v = np.array(list(buf), dtype=np.float64)
    tot = v.sum()
    if tot == 0:
        continue
    print(f"level {lev}: total ticks {tot:.3e}")
    for i in np.argsort(-v):
        if v[i] > 0:
            print(f"   {NAMES.get(int(i), str(i)):28s} {100 * v[i] / tot:5.1f} %")
hm.close()
