#!/usr/bin/env python3
"""Per-kernel means of every counter in one or more rocprofv3 rocpd databases, one row per kernel, plus the ratios that say what a
kernel's waves wait for: WAIT_ANY / WAVE_CYCLES (waves stalled on any counter), WAIT_INST_ANY / WAVE_CYCLES (waiting for an issue
slot), WAIT_INST_LDS / WAVE_CYCLES, LDS_BANK_CONFLICT / LDS_IDX_ACTIVE.   python tools/pmc_wait_summary.py a.db [b.db ...]"""
import re, sqlite3, sys, collections
def short(n):
    return re.sub(r"\(.*$", "", re.sub(r"^void ", "", n).replace("(anonymous namespace)::", ""))[:72]
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for path in sys.argv[1:]:
    cur = sqlite3.connect(path).cursor()
    for name, cname, val in cur.execute("select name, counter_name, counter_value from pmc_events"):
        a = agg[short(name)][cname]; a[0] += 1; a[1] += val
cols = sorted({c for k in agg for c in agg[k]})
def mean(k, c):
    n, t = agg[k].get(c, (0, 0.0)); return t / n if n else float("nan")
def ratio(k, a, b):
    x, y = mean(k, a), mean(k, b)
    return x / y if y == y and y > 0 and x == x else float("nan")
print("| kernel | wait_any/wave | wait_inst/wave | wait_lds/wave | lds_conflict/lds_active | lds_fifo_full/wave | " + " | ".join(cols) + " |")
print("|---|---|---|---|---|---|" + "---|" * len(cols))
for k in sorted(agg, key=lambda k: -mean(k, "SQ_WAVE_CYCLES") if mean(k, "SQ_WAVE_CYCLES") == mean(k, "SQ_WAVE_CYCLES") else 0):
    if not (mean(k, "SQ_WAVE_CYCLES") > 1e5):
        continue
    print("| `%s` | %.2f | %.2f | %.2f | %.2f | %.3f | " % (k, ratio(k, "SQ_WAIT_ANY", "SQ_WAVE_CYCLES"), ratio(k, "SQ_WAIT_INST_ANY", "SQ_WAVE_CYCLES"),
          ratio(k, "SQ_WAIT_INST_LDS", "SQ_WAVE_CYCLES"), ratio(k, "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"),
          ratio(k, "SQ_LDS_CMD_FIFO_FULL", "SQ_WAVE_CYCLES")) + " | ".join("%.3g" % mean(k, c) for c in cols) + " |")
