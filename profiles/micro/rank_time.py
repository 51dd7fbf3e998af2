"""Diagnostic (not part of the product): what ONE rank of an N-GPU run computes per iteration, timed on a single GPU
without the collectives (the exchanged buffers simply keep the other ranks' entries at zero): the fixed cost that bounds
strong scaling.  Run on the GPU box:  python profiles/micro/rank_time.py [side] [q] [cell_size] [missing, e.g. 0.1,0.3,0.5]
(config #3: 1000; #4: 577 3; #5: 1155 3 9 0.1,0.3,0.5).  SINGLE-GPU ESTIMATE of the per-rank compute: no multi-GPU run is behind it."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from spamtree_amd.model import SpamTreeMV, _dp, _f64  # noqa: E402
from spamtree_amd.synthetic import make_workload  # noqa: E402

side = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
q = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cell = int(sys.argv[3]) if len(sys.argv) > 3 else 25
missing = tuple(float(x) for x in sys.argv[4].split(",")) if len(sys.argv) > 4 else None
wl = make_workload(side, q=q, cell_size=cell, missing=missing, device=0)
print(f"workload: side {side} q {q} cell_size {cell} missing {missing}: n = {wl['n']}")
rng = np.random.default_rng(0)
for world in (1, 2, 4, 8):
    hm = SpamTreeMV(wl["y"], wl["X"], wl["Z"], wl["coords"], wl["mv_id"], wl["blocking"], wl["gix_block"], wl["res_is_ref"],
                    wl["parents"], wl["children"], False, wl["block_names"], wl["block_groups"], wl["indexing"],
                    np.zeros(wl["n"]), np.zeros(wl["p"]), wl["theta"], 10.0, device=0, rank=0, world=world)
    lib, h = hm.lib, hm.h
    th = _f64(wl["theta"])
    ptr, n, ll = C.c_void_p(), C.c_int64(), C.c_double()

    def factor(slot):
        lib.st_factor_local(h, slot, _dp(th), th.size)
        lib.st_mg_pack_comps(h, slot, C.byref(ptr), C.byref(n))
        lib.st_mg_finish(h, C.byref(ll))

    def sample(it):
        lib.st_sample_w_local(h, None, 7, it)
        lib.st_mg_top_region(h, C.byref(ptr), C.byref(n))
        lib.st_sample_w_top(h)
        lib.st_mg_pack_w(h, C.byref(ptr), C.byref(n))
        lib.st_mg_unpack_w(h)

    def loglik(slot):
        lib.st_loglik_local(h, slot)
        lib.st_mg_pack_comps(h, slot, C.byref(ptr), C.byref(n))
        lib.st_mg_finish(h, C.byref(ll))

    if world == 1:
        def factor(slot):  # noqa: F811
            lib.st_factor(h, slot, _dp(th), th.size, C.byref(ll))

        def sample(it):  # noqa: F811
            lib.st_sample_w(h, None, 7, it)

        def loglik(slot):  # noqa: F811
            lib.st_loglik_w(h, slot, C.byref(ll))
    factor(0)
    lib.st_profile_enable(h, 1)
    for _ in range(5):
        factor(1)
    lvl = hm.profile_levels()[0]
    lib.st_profile_enable(h, 0)
    print(f"world {world}: phase A by level (ms): {[round(float(x), 3) for x in lvl]}")
    res = {}
    for name, fn, arg in (("A", factor, 1), ("B", sample, 3), ("C", loglik, 0)):
        for _ in range(3):
            fn(arg)
        lib.st_synchronize(h)
        t0 = time.perf_counter()
        for i in range(20):
            fn(arg)
        lib.st_synchronize(h)
        res[name] = (time.perf_counter() - t0) / 20 * 1e3
    # whole iterations (B, C, A back to back), without and with phase A of the top levels started ahead (st_factor_begin)
    lib.st_factor_begin.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double), C.c_int]
    for name, ahead in (("iter", False), ("iter_ahead", True)):
        for rep in range(23):
            if rep == 3:
                lib.st_synchronize(h)
                t0 = time.perf_counter()
            if ahead:
                lib.st_factor_begin(h, 1, _dp(th), th.size)
            sample(rep); loglik(0); factor(1)
        lib.st_synchronize(h)
        res[name] = (time.perf_counter() - t0) / 20 * 1e3
    print(f"world {world}: whole iteration {res['iter']:.3f} ms; with the top levels ahead of time {res['iter_ahead']:.3f} ms")
    tot = res["A"] + res["B"] + res["C"]
    print(f"world {world}: rank 0 per iteration  A {res['A']:.3f}  B {res['B']:.3f}  C {res['C']:.3f}  sum {tot:.3f} ms"
          f"  -> ideal strong-scaling speed-up without collectives: {0 if world == 1 else base / tot:.2f}x" if world > 1 else
          f"world 1: A {res['A']:.3f}  B {res['B']:.3f}  C {res['C']:.3f}  sum {tot:.3f} ms")
    if world == 1:
        base = tot
    hm.close()
