#!/usr/bin/env python3
"""Generates the committed golden vectors from the NumPy oracle (oracle/spamtree_oracle.py).

The reference ships no fixtures and cannot be run here, so these vectors pin THIS
repository's oracle against regressions and give the GPU path a device-independent target; they are not outputs
of the reference.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from tests.util import make_problem, oracle_model  # noqa: E402

CASES = {
    "q1_n625": dict(side=25, q=1, seed=101, missing=0.1),
    "q1_n1600_random": dict(side=40, q=1, seed=102, random_coords=True),
    "q3_n588": dict(side=14, q=3, seed=103, missing=0.15),
    "q1_n900_limited": dict(side=30, q=1, seed=104, missing=0.1, limited_tree=True),   # limited_tree = TRUE (tree_dep.cpp:133-186)
}


def csr(lists):
    ptr = np.zeros(len(lists) + 1, dtype=np.int64)
    ptr[1:] = np.cumsum([len(x) for x in lists])
    idx = np.concatenate([np.asarray(x, dtype=np.int64) for x in lists]) if ptr[-1] else np.zeros(0, dtype=np.int64)
    return ptr, idx


def main():
    out_dir = os.path.dirname(os.path.abspath(__file__))
    only = set(sys.argv[1:])                     # python tests/golden/make_golden.py [case ...]: default all
    for name, kw in CASES.items():
        if only and name not in only:
            continue
        pb = make_problem(**kw)
        rng = np.random.default_rng(kw["seed"] + 1000)
        w0 = rng.standard_normal(pb["n"])
        beta = np.array([0.3, -0.2, 0.1])
        tausq = 0.2
        om = oracle_model(pb, w=w0, beta=beta, tausq=tausq)
        assert om.get_loglik_comps_w(om.param_data)
        ip, ii = csr(pb["indexing"]); pp, pi = csr(pb["parents"]); cp, ci = csr(pb["children"])
        d = dict(y=pb["y"], X=pb["X"], Z=pb["Z"], coords=pb["coords"], mv_id=pb["mv_id"], blocking=pb["blocking"],
                 gix_block=pb["gix_block"], res_is_ref=pb["res_is_ref"], block_names=pb["block_names"],
                 block_groups=pb["block_groups"], indexing_ptr=ip, indexing_idx=ii, parents_ptr=pp, parents_idx=pi,
                 children_ptr=cp, children_idx=ci, theta=pb["theta"], beta=beta, tausq=np.array(tausq), w0=w0,
                 limited_tree=np.array(int(bool(kw.get("limited_tree", False)))),
                 loglik_A=np.array(om.param_data.loglik_w), logdet_comps=om.param_data.logdetCi_comps.copy(),
                 loglik_comps=om.param_data.loglik_w_comps.copy())
        # a few per-block caches
        blocks = [u for u in range(om.n_blocks) if om.block_ct_obs[u] > 0 and om.parents[u].size][:6]
        d["cache_blocks"] = np.array(blocks, dtype=np.int64)
        for u in blocks:
            d[f"H_{u}"] = om.param_data.w_cond_mean_K[u]
            d[f"Ri_{u}"] = om.param_data.Rcc_invchol[u] if om.block_is_reference[u] else om.param_data.ccholprecdiag[u]
        zs, ws, lls = [], [], []
        for _ in range(3):
            z = rng.standard_normal(pb["n"])
            om.gibbs_sample_w(z)
            om.get_loglik_w(om.param_data)
            zs.append(z); ws.append(om.w.copy()); lls.append(om.param_data.loglik_w)
        xty, ssq = om.beta_tausq_stats()
        om.predict(True)
        d.update(z=np.array(zs), w_sweeps=np.array(ws), loglik_w=np.array(lls), xty=xty, ssq=ssq, w_predict=om.w.copy())
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), **d)
        print(name, "n", pb["n"], "blocks", om.n_blocks, "loglik", float(d["loglik_A"]))


if __name__ == "__main__":
    main()
