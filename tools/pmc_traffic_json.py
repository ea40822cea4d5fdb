#!/usr/bin/env python3
"""Combine a FETCH_SIZE and a WRITE_SIZE rocprofv3 --pmc pass (two rocpd databases) into the per-kernel HBM-traffic JSON
bench.py reads: hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950 FETCH_SIZE half-count correction, see
/opt/skills/guides/MI355X_MICROARCH.md and profiles/r01_pmc_fetch_write.md)."""
import collections, json, re, sqlite3, sys


def short(n):
    return re.sub(r"\(.*$", "", re.sub(r"^void ", "", n).replace("(anonymous namespace)::", ""))[:90]


def means(path, counter):
    cur = sqlite3.connect(path).cursor()
    agg = collections.defaultdict(lambda: [0, 0.0])
    for name, cname, val in cur.execute("select name, counter_name, counter_value from pmc_events"):
        if cname == counter:
            a = agg[short(name)]; a[0] += 1; a[1] += val
    counts.update({k: v[0] for k, v in agg.items()})
    return {k: v[1] / v[0] for k, v in agg.items()}


import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from desed_task_amd.build import source_rev
counts = {}
fetch, write = means(sys.argv[1], "FETCH_SIZE"), means(sys.argv[2], "WRITE_SIZE")
steps = int(sys.argv[3]) if len(sys.argv) > 3 else None        # training steps the profiled command ran (all of them eager)
out = {"note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over bench.py --no-graph; HBM bytes = "
               "(2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 FETCH_SIZE half-count correction)",
       "lib_rev": source_rev(), "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    f, w = fetch.get(k, 0.0), write.get(k, 0.0)
    if f + w < 64:
        continue
    out["kernels"][k] = {"fetch_kib_raw": round(f, 1), "write_kib": round(w, 1), "hbm_bytes": int((2 * f + w) * 1024),
                         "dispatches": counts.get(k, 0)}
if steps:
    # whole-step HBM traffic: sum over every kernel of (bytes per launch x launches) / steps of the profiled run
    out["steps"] = steps
    out["hbm_bytes_per_step"] = int(sum(r["hbm_bytes"] * r["dispatches"] for r in out["kernels"].values()) / steps)
json.dump(out, sys.stdout, indent=1)
