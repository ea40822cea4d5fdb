#!/usr/bin/env python3
"""Per-kernel MFMA-pipe occupancy from one rocprofv3 --pmc pass (rocpd database):
    pmc_ratio_summary.py <results.db> [skip_first_frac]
Counters expected: SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE.  rocprofv3 reports one row per hardware instance
(XCD / shader engine) per dispatch: the SQ counters are SUMMED over the instances of a dispatch, GRBM_GUI_ACTIVE (wall cycles
of the dispatch, one copy per XCD) is taken as the maximum.  Columns per kernel (means over its dispatches):
  mfma_busy      SQ_VALU_MFMA_BUSY_CYCLES, per-SIMD cycles the MFMA pipe was busy (32 per v_mfma_f32_32x32x16_bf16,
                 64 per v_mfma_f32_32x32x2_f32, 32 per 16x16x4_f32), summed over the chip's 1024 SIMDs
  gui_active     wall cycles of the dispatch
  mfma_util      mfma_busy / (gui_active * 1024 SIMDs) = fraction of the chip's MFMA issue capacity in use while the kernel ran
  mfma/sq_busy   mfma_busy / SQ_BUSY_CYCLES (SQ_BUSY_CYCLES counts per shader engine, so this is a ratio, not a fraction)"""
import collections, re, sqlite3, sys
db = sys.argv[1]
skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
cur = sqlite3.connect(db).cursor()
per = collections.defaultdict(lambda: collections.defaultdict(list))
names, starts = {}, {}
for did, name, start, c, v in cur.execute("select dispatch_id, name, start, counter_name, counter_value from pmc_events"):
    per[did][c].append(v)
    names[did] = re.sub(r"\(.*$", "", re.sub(r"^void ", "", name).replace("(anonymous namespace)::", ""))[:90]
    starts[did] = start
if not per:
    print("no pmc events"); sys.exit(0)
t0, t1 = min(starts.values()), max(starts.values())
cut = t0 + (t1 - t0) * skip
agg = collections.defaultdict(lambda: [0, 0.0, 0.0, 0.0])
for did, d in per.items():
    if starts[did] < cut or not names[did].strip():
        continue
    a = agg[names[did]]
    a[0] += 1
    a[1] += sum(d.get("SQ_VALU_MFMA_BUSY_CYCLES", [0.0]))
    a[2] += sum(d.get("SQ_BUSY_CYCLES", [0.0]))
    a[3] += max(d.get("GRBM_GUI_ACTIVE", [0.0]))
rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
print("| kernel | dispatches | mfma_busy (SIMD-cycles) | gui_active (cycles) | mfma_util | SQ_BUSY_CYCLES | mfma/sq_busy |")
print("|---|---|---|---|---|---|---|")
for k, (n, m, sq, g) in rows:
    if m <= 0:
        continue
    print("| `%s` | %d | %.0f | %.0f | %.3f | %.0f | %.2f |" % (k, n, m / n, g / n, m / (g * 1024.0) if g else 0.0, sq / n, m / sq if sq else 0.0))
