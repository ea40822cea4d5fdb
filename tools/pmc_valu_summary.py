#!/usr/bin/env python3
"""Per-kernel VALU / LDS issue occupancy from one rocprofv3 --pmc pass (rocpd database):
    pmc_valu_summary.py <results.db>
Counters expected: SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE.  The SQ counters are summed
over the hardware instances of a dispatch, GRBM_GUI_ACTIVE (wall cycles) is the maximum.  Columns (means over a kernel's dispatches):
  valu_util = 4 * SQ_INSTS_VALU / (gui_active * 1024)   fraction of the chip's VALU issue capacity in use: a wave64 VALU instruction
              occupies its SIMD for 4 cycles (more for transcendentals, 64-bit integer and 32-bit multiplies: a lower estimate),
              1024 SIMDs.  SQ_ACTIVE_INST_VALU / SQ_ACTIVE_INST_LDS are printed raw."""
import collections, re, sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
per = collections.defaultdict(lambda: collections.defaultdict(list))
names = {}
for did, name, c, v in cur.execute("select dispatch_id, name, counter_name, counter_value from pmc_events"):
    per[did][c].append(v)
    names[did] = re.sub(r"\(.*$", "", re.sub(r"^void ", "", name))[:80]
agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0, 0.0])
for did, d in per.items():
    if not names[did].strip():
        continue
    a = agg[names[did]]
    a[0] += 1
    a[1] += sum(d.get("SQ_ACTIVE_INST_VALU", [0.0]))
    a[2] += sum(d.get("SQ_ACTIVE_INST_LDS", [0.0]))
    a[3] += sum(d.get("SQ_INSTS_VALU", [0.0]))
    a[4] += max(d.get("GRBM_GUI_ACTIVE", [0.0]))
print("| kernel | dispatches | gui_active (cycles) | SQ_INSTS_VALU | valu_util | SQ_ACTIVE_INST_VALU | SQ_ACTIVE_INST_LDS |")
print("|---|---|---|---|---|---|---|")
for k, (n, av, al, iv, g) in sorted(agg.items(), key=lambda kv: -kv[1][4]):
    if g <= 0 or g / n < 5000:
        continue
    print("| `%s` | %d | %.0f | %.0f | %.3f | %.0f | %.0f |" % (k, n, g / n, iv / n, 4 * iv / (g * 1024), av / n, al / n))
