"""Condenses the raw rocprofv3 output of profiles/collect.sh (gpurun_out/<tag>_*) into the committed summaries
profiles/r01/<tag>_bench.json, <tag>_kernel_stats.csv and <tag>_pmc_hbm.json.

HBM bytes per kernel family = FETCH_SIZE x 2 (the gfx950 correction of MI355X_MICROARCH.md: 128-byte requests are
tallied at 64 bytes) + WRITE_SIZE, both reported by rocprofv3 in KB, from separate counter passes."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "c_quad"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out")
dst = os.path.join(root, "profiles", "r01")
os.makedirs(dst, exist_ok=True)


def family(name):
    for f in ("k_factor_quad", "k_factor_mfma", "k_factor", "k_sample_mfma", "k_sample_lean", "k_sample_leaf", "k_sample", "k_loglik_grp", "k_loglik", "k_sum2", "k_stats",
              "k_xb", "k_normals"):
        if f in name:
            return f
    return name[:40]


b = os.path.join(src, f"{tag}_bench.json")
if os.path.exists(b):
    line = [x for x in open(b).read().strip().splitlines() if x.startswith("{")][-1]
    json.dump(json.loads(line), open(os.path.join(dst, f"{tag}_bench.json"), "w"), indent=1)

st = sorted(glob.glob(os.path.join(src, f"{tag}_trace", "*", "*kernel_stats.csv")), key=os.path.getmtime)   # newest run
if st:
    shutil.copy(st[-1], os.path.join(dst, f"{tag}_kernel_stats.csv"))

out = {"units": "bytes", "note": "FETCH_SIZE doubled (gfx950), WRITE_SIZE as reported; per launch = mean over launches"}
per = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    fs = sorted(glob.glob(os.path.join(src, f"{tag}_pmc_{c}", "*", "*counter_collection.csv")), key=os.path.getmtime)
    if not fs:
        continue
    by = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[-1])):
        if r.get("Counter_Name") == c:
            by[family(r["Kernel_Name"])].append(float(r["Counter_Value"]) * 1024.0)
    per[c] = {k: {"launches": len(v), "mean_bytes": sum(v) / len(v), "total_bytes": sum(v)} for k, v in by.items()}
out["raw"] = per
summ = {}
if "FETCH_SIZE" in per and "WRITE_SIZE" in per:
    for k in per["FETCH_SIZE"]:
        f, w = per["FETCH_SIZE"][k], per["WRITE_SIZE"].get(k, {"mean_bytes": 0.0, "launches": 0})
        summ[k] = {"launches_in_run": f["launches"], "fetch_bytes_per_launch_x2": 2.0 * f["mean_bytes"],
                   "write_bytes_per_launch": w["mean_bytes"], "hbm_bytes_per_launch": 2.0 * f["mean_bytes"] + w["mean_bytes"]}
    # phase A = every k_factor* launch of one factorisation (one launch per tree level)
    fa = [k for k in summ if k.startswith("k_factor")]
    tot = sum(2.0 * per["FETCH_SIZE"][k]["total_bytes"] + per["WRITE_SIZE"].get(k, {"total_bytes": 0.0})["total_bytes"] for k in fa)
    n = sum(per["FETCH_SIZE"][k]["launches"] for k in fa)
    summ["phase_A"] = {"kernels": fa, "launches_in_run": n, "hbm_bytes_per_launch": tot / max(n, 1)}
out["summary"] = summ
json.dump(out, open(os.path.join(dst, f"{tag}_pmc_hbm.json"), "w"), indent=1)
print(json.dumps(summ, indent=1))
