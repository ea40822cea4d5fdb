#!/usr/bin/env python3
"""Per-kernel average of one PMC counter from a rocprofv3 rocpd database (value units as reported; FETCH_SIZE / WRITE_SIZE are KiB)."""
import re, sqlite3, sys, collections
def short(n):
    return re.sub(r"\(.*$", "", re.sub(r"^void ", "", n).replace("(anonymous namespace)::", ""))[:90]
db = sqlite3.connect(sys.argv[1]); cur = db.cursor()
agg = collections.defaultdict(lambda: [0, 0.0])
for name, cname, val in cur.execute("select name, counter_name, counter_value from pmc_events"):
    a = agg[(short(name), cname)]; a[0] += 1; a[1] += val
print("| kernel | counter | dispatches | mean value (KiB) | mean MB |")
print("|---|---|---|---|---|")
for (k, c), (n, tot) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if tot / n < 64: continue
    print("| `%s` | %s | %d | %.1f | %.2f |" % (k, c, n, tot / n, tot / n * 1024 / 1e6))
