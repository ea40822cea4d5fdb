#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd sqlite database (--kernel-trace) into a per-kernel table (markdown)."""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.replace("(anonymous namespace)::", "")
    name = re.sub(r"\(.*$", "", name)
    return name[:110]


def main(path, skip_first_frac=0.0):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    rows = cur.execute("select name, start, end from kernels order by start").fetchall()
    if not rows:
        print("no kernels"); return
    t0, t1 = rows[0][1], rows[-1][2]
    cut = t0 + (t1 - t0) * skip_first_frac
    agg = {}
    for name, s, e in rows:
        if s < cut:
            continue
        k = short(name)
        a = agg.setdefault(k, [0, 0.0, 1e30, 0.0])
        d = (e - s) / 1e3
        a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    print("| kernel | calls | total us | avg us | min us | max us | % |")
    print("|---|---|---|---|---|---|---|")
    for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("| `%s` | %d | %.1f | %.2f | %.2f | %.2f | %.1f |" % (k, a[0], a[1], a[1] / a[0], a[2], a[3], 100 * a[1] / tot))
    print("\ntotal kernel time %.1f us over %d dispatches; wall span %.1f us" % (tot, sum(a[0] for a in agg.values()), (t1 - cut) / 1e3))


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.0)
