"""CPU baseline at several sizes: oracle/refcpu (the OpenMP restatement of the reference algorithm as
written) timed on the GPU box's host cores at n ~ 4e4, 1e5, 2.5e5, 5e5 (and 1e6 when its caches fit the host memory budget),
so that the cost-law extrapolation bench.py's `cpu_baseline` uses can be checked against measurements.

Run on the GPU box:  python profiles/cpu_scaling.py [--max-gb 200]   ->  gpurun_out/cpu_scaling.json  (copy to profiles/<round>/)
Test infrastructure only (uses oracle/): never part of the product path."""
import argparse
import json
import os
import resource
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import host_cpu, reference_cost  # noqa: E402
from oracle.refcpu import RefCpu  # noqa: E402
from spamtree_amd.synthetic import make_workload  # noqa: E402


def rss_gb():
    return resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024.0 ** 2


def time_one(side, threads, budget_s, min_iters=2):
    wl = make_workload(side)
    r0 = rss_gb()
    t_init = time.perf_counter()
    rc = RefCpu(wl["y"], wl["X"], wl["coords"], wl["mv_id"], wl["res_is_ref"], wl["parents"], wl["children"], wl["block_names"],
                wl["block_groups"], wl["indexing"], threads=threads)
    rc.set_tausq_inv(10.0)
    rc.set_beta(np.zeros((wl["p"], 1)))
    rng = np.random.default_rng(1)
    rc.factor(0, wl["theta"])
    rc.factor(1, wl["theta"])                 # touches the second cache copy too (page faults are not timed)
    t_init = time.perf_counter() - t_init
    its, t0 = 0, time.perf_counter()
    while True:
        rc.sample_w(rng.standard_normal(wl["n"]))
        rc.loglik_w(0)
        rc.factor(1, wl["theta"] * (1 + 0.01 * rng.standard_normal(wl["theta"].size)))
        rc.stats()
        its += 1
        dt = time.perf_counter() - t0
        if (dt > budget_s and its >= min_iters) or its >= 200:
            break
    peak = rss_gb()
    rc.close()
    return dict(side=side, n=int(wl["n"]), iterations=its, seconds=round(dt, 3), it_per_s=its / dt, init_s=round(t_init, 1),
                rss_gb=round(peak, 2), rss_delta_gb=round(peak - r0, 2), reference_cost_flops=reference_cost(wl))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--max-gb", type=float, default=200.0, help="skip a size whose extrapolated resident set exceeds this")
    ap.add_argument("--budget", type=float, default=12.0, help="seconds of timed iterations per size")
    ap.add_argument("--sides", type=str, default="200,316,500,707,1000")
    args = ap.parse_args()
    threads = int(os.environ.get("SPAMTREE_CPU_THREADS", min(os.cpu_count() or 1, 16)))
    out = {"host": host_cpu(), "threads": threads, "kind": "port (oracle/refcpu, g++ -O2 -fopenmp, own potrf/trtri/gemm kernels: no BLAS in the image)",
           "points": [], "skipped": []}
    for side in [int(x) for x in args.sides.split(",")]:
        if out["points"]:
            last = out["points"][-1]
            est = last["rss_delta_gb"] * (side * side) / last["n"] * 1.15 + 4.0
            if est > args.max_gb:
                out["skipped"].append({"side": side, "n": side * side, "estimated_rss_gb": round(est, 1), "limit_gb": args.max_gb})
                continue
        p = time_one(side, threads, args.budget)
        out["points"].append(p)
        print(json.dumps(p), flush=True)
    pts = out["points"]
    if len(pts) >= 2:
        ln, ls = np.log([p["n"] for p in pts]), np.log([1.0 / p["it_per_s"] for p in pts])
        out["fitted_exponent_seconds_vs_n"] = float(np.polyfit(ln, ls, 1)[0])
        lc = np.log([p["reference_cost_flops"] for p in pts])
        out["fitted_exponent_seconds_vs_cost_law"] = float(np.polyfit(lc, ls, 1)[0])     # 1.0 = the law predicts the scaling
        base = min(pts, key=lambda p: abs(p["n"] - 99856))
        for p in pts:
            pred = base["it_per_s"] * base["reference_cost_flops"] / p["reference_cost_flops"]
            p["cost_law_prediction_from_n1e5_it_per_s"] = pred
            p["measured_over_predicted"] = p["it_per_s"] / pred
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "cpu_scaling.json"), "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "points"}))


if __name__ == "__main__":
    main()
