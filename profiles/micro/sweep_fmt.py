import json
import sys

d = json.loads(sys.stdin.read())
r = d["roofline"]
print(round(d["value"], 1), "it/s  frac", round(r["frac"], 3), " A by level:", r["by_level_ms"], " phases:", r["phase_ms_per_iter"],
      " B by level:", r.get("sample_by_level_ms"))
