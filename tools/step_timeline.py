#!/usr/bin/env python3
"""Timeline of ONE training step from a rocprofv3 --kernel-trace rocpd database: every kernel between the last two adam_kernel
launches with its start offset, duration and stream, plus gaps (time no kernel runs) and overlap (time > 1 kernel runs).
    python tools/step_timeline.py results.db [min_us_to_print] [step counted from the end: bench.py runs 6 eager steps last]"""
import re, sqlite3, sys


def short(name):
    name = re.sub(r"^void ", "", name).replace("(anonymous namespace)::", "")
    return re.sub(r"\(.*$", "", name)[:70]


def main(path, min_us=0.0, back=8):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select name, start, end, stream_id, queue_id from kernels order by start").fetchall()
    adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[0]]
    if len(adam) < 3:
        print("need >= 3 steps"); return
    if back == 0:
        # auto: among the last 16 steps, the replayed one with the shortest span (a tracer-induced host stall inside a step -- the
        # host is several times slower under rocprofv3 -- stretches its span but not its kernels)
        best = None
        for k in range(2, min(17, len(adam))):
            a, b_ = adam[-k - 1], adam[-k]
            if b_ - a >= 100:
                span_ = rows[b_][2] - rows[a + 1][1]
                if best is None or span_ < best[0]:
                    best = (span_, k)
        back = best[1] if best else 8
    lo, hi = adam[-back - 1] + 1, adam[-back] + 1
    step = rows[lo:hi]
    t0 = rows[adam[-back - 1]][2]
    print("NOTE: a KERNEL INVENTORY under the tracer, not the timeline of the untraced replay -- rocprofv3 re-maps the branches of the replayed "
          "graph onto hardware queues (DESIGN.md 12.6: `head_bwd` starts 500 us after the loss here, 4 us in the untraced step).  The in-graph "
          "wall-clock stamps the step-level decisions were taken from: `bench.py --ts-probe` -> profiles/*_ts_probe.json.\n")
    print("step = %d kernels, %.1f us from the previous Adam's end to this Adam's end" % (len(step), (step[-1][2] - t0) / 1e3))
    print("| start us | dur us | stream/queue | kernel |"); print("|---|---|---|---|")
    for name, s, e, st, q in step:
        if (e - s) / 1e3 >= min_us:
            print("| %8.1f | %7.1f | %s/%s | `%s` |" % ((s - t0) / 1e3, (e - s) / 1e3, st, q, short(name)))
    ev = sorted([(s, 1) for _, s, e, _, _ in step] + [(e, -1) for _, s, e, _, _ in step])
    busy = over = 0; depth = 0; prev = ev[0][0]
    for t, d in ev:
        if depth >= 1: busy += t - prev
        if depth >= 2: over += t - prev
        depth += d; prev = t
    span = step[-1][2] - step[0][1]
    print("\nspan %.1f us, busy %.1f us (idle %.1f), >= 2 kernels in flight %.1f us, sum of durations %.1f us"
          % (span / 1e3, busy / 1e3, (span - busy) / 1e3, over / 1e3, sum(e - s for _, s, e, _, _ in step) / 1e3))


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.0, int(sys.argv[3]) if len(sys.argv) > 3 else 8)
