"""Condenses the raw rocprofv3 output of profiles/collect.sh (gpurun_out/<tag>_*) into the committed summaries
profiles/<round>/<tag>_bench.json (ROUND env, default r03), <tag>_kernel_stats.csv and <tag>_pmc_hbm.json.

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
dst = os.path.join(root, "profiles", os.environ.get("ROUND", "r03"))
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
if summ:
    # bench.py reads the counter summary named HERE (file + the commit whose working tree was measured), not the last one by name
    import subprocess
    head = subprocess.run(["git", "-C", root, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip()
    dirty = bool(subprocess.run(["git", "-C", root, "status", "--porcelain", "--", "spamtree_amd", "include"], capture_output=True, text=True).stdout.strip())
    json.dump({"file": os.path.relpath(os.path.join(dst, f"{tag}_pmc_hbm.json"), root), "commit": head, "uncommitted_source_changes": dirty,
               "workload": "config #3 (bench.py defaults), one GPU"}, open(os.path.join(root, "profiles", "pmc_source.json"), "w"), indent=1)

# ---- SQ counter pass (collect.sh, own --pmc run): per kernel INSTANTIATION (the quad kernel's template arguments tell the
# levels apart), means per launch; mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (launch duration x 2.4 GHz x 1024 SIMDs)
fs = sorted(glob.glob(os.path.join(src, f"{tag}_pmc_SQ", "*", "*counter_collection.csv")), key=os.path.getmtime)
if fs:
    import re
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(dict)
    for r in csv.DictReader(open(fs[-1])):
        name = r["Kernel_Name"]
        m = re.match(r"void (k_\w+)(<[^>]*>)?", name)
        key = (m.group(1) + (m.group(2) or "")) if m else name[:48]
        acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[key][r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-6
    sq = {}
    for key, cs in acc.items():
        if not key.startswith("k_"):
            continue
        d = {c: sum(v) / len(v) for c, v in cs.items()}
        ms = sum(dur[key].values()) / max(1, len(dur[key]))
        e = {"launches": len(dur[key]), "mean_ms_under_pmc": round(ms, 4)}
        e.update({c: round(v, 1) for c, v in d.items()})
        wc = d.get("SQ_WAVE_CYCLES", 0.0)
        if ms > 0:
            e["mfma_busy_frac"] = round(d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (ms * 1e-3 * 2.4e9 * 1024), 3)
        if wc > 0:
            e["wave_parked_frac"] = round(d.get("SQ_WAIT_ANY", 0.0) / wc, 3)
            e["wave_issue_stalled_frac"] = round(d.get("SQ_WAIT_INST_ANY", 0.0) / wc, 3)
            e["wave_issuing_frac"] = round(d.get("SQ_ACTIVE_INST_ANY", 0.0) / wc, 3)
        sq[key] = e
    json.dump({"note": "rocprofv3 --pmc SQ_* --kernel-trace (own pass); means per launch; wave fractions are shares of SQ_WAVE_CYCLES "
                       "(parked at s_waitcnt / barrier, issue-stalled, issuing)", "kernels": sq},
              open(os.path.join(dst, f"{tag}_pmc_sq.json"), "w"), indent=1)
