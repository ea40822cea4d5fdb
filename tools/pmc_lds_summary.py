#!/usr/bin/env python3
"""Per-kernel LDS bank-conflict share from a rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE pass (rocpd database):
mean per dispatch of both counters and their ratio (conflict cycles / LDS-array active cycles)."""
import collections, re, sqlite3, sys
cur = sqlite3.connect(sys.argv[1]).cursor()
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for name, c, v in cur.execute("select name, counter_name, counter_value from pmc_events"):
    k = re.sub(r"\(.*$", "", re.sub(r"^void ", "", name))[:80]
    a = agg[k][c]; a[0] += 1; a[1] += v
rows = []
for k, d in agg.items():
    bc, ia = d["SQ_LDS_BANK_CONFLICT"], d["SQ_LDS_IDX_ACTIVE"]
    if ia[0] and ia[1] > 0 and k.strip():
        rows.append((bc[1] / max(bc[0], 1), ia[1] / ia[0], k))
rows.sort(reverse=True)
print("| kernel | SQ_LDS_BANK_CONFLICT (mean / dispatch) | SQ_LDS_IDX_ACTIVE | conflict share |")
print("|---|---|---|---|")
for bc, ia, k in rows:
    print("| `%s` | %.0f | %.0f | %.2f |" % (k, bc, ia, bc / ia))
