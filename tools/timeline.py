#!/usr/bin/env python3
"""Critical-path view of one training step from a rocprofv3 --kernel-trace CSV.

usage: tools/timeline.py <kernel_trace.csv> [step_index_from_end=2] [out.txt]

A step is delimited by consecutive adam_kernel launches.  Prints, for the chosen step: wall time, the time with
0 / 1 / 2 / >=3 kernels in flight, per-queue busy time, the largest gaps with nothing running, and a coarse
phase table (forward / backward split at head_bwd_kernel) with the kernel families active in each.
"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"^void ", "", name).replace("wun::", "")
    m = re.match(r"([a-z0-9_]+)(<[^>]*>)?", name)
    return (m.group(1) + (m.group(2) or "")) if m else name[:40]


def main():
    path = sys.argv[1]
    back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    out = open(sys.argv[3], "w") if len(sys.argv) > 3 else sys.stdout
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")))
    rows.sort()
    adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
    if len(adam) < back + 1:
        print("not enough steps in the trace", file=out)
        return
    lo, hi = adam[-back - 1] + 1, adam[-back] + 1
    step = rows[lo:hi]
    t0 = rows[adam[-back - 1]][1]           # end of the previous step's optimizer
    t1 = step[-1][1]
    p = lambda *a: print(*a, file=out)
    p("step: %d kernels, wall %.3f ms (previous adam end -> this adam end)" % (len(step), (t1 - t0) / 1e6))
    # concurrency profile
    ev = []
    for s, e, _, _ in step:
        ev.append((s, 1)); ev.append((e, -1))
    ev.sort()
    depth, last, hist = 0, t0, defaultdict(int)
    gaps = []
    for t, d in ev:
        if t > last:
            hist[min(depth, 3)] += t - last
            if depth == 0:
                gaps.append((t - last, last))
        depth += d
        last = t
    for k in range(4):
        p("  %s kernels in flight: %.3f ms" % (">=3" if k == 3 else str(k), hist[k] / 1e6))
    p("  idle gaps: %d, total %.3f ms; largest:" % (len(gaps), sum(g for g, _ in gaps) / 1e6))
    starts = sorted((s, n) for s, _, n, _ in step)
    for g, at in sorted(gaps, reverse=True)[:8]:
        nxt = next((short(n) for s, n in starts if s >= at + g), "?")
        p("    %.1f us at +%.3f ms before %s" % (g / 1e3, (at - t0) / 1e6, nxt))
    busy = defaultdict(int)
    for s, e, _, q in step:
        busy[q] += e - s
    for q, b in sorted(busy.items(), key=lambda kv: -kv[1]):
        p("  queue %s busy %.3f ms" % (q, b / 1e6))
    # dependent-chain stalls: gaps between consecutive kernels of the busiest queue (barrier / event packets)
    mainq = max(busy.items(), key=lambda kv: kv[1])[0]
    mk = sorted((s, e) for s, e, _, q in step if q == mainq)
    g = [b[0] - a[1] for a, b in zip(mk, mk[1:]) if b[0] - a[1] > 1000]
    p("  queue %s: %d gaps > 1 us between consecutive kernels, total %.3f ms (median %.1f us)" %
      (mainq, len(g), sum(g) / 1e6, sorted(g)[len(g) // 2] / 1e3 if g else 0.0))
    # phases
    hb = next((s for s, _, n, _ in step if "head_bwd_kernel" in n), t1)
    for label, a, b in (("forward", t0, hb), ("backward+adam", hb, t1)):
        fam = defaultdict(lambda: [0, 0])
        for s, e, n, _ in step:
            if a <= s < b:
                f = fam[short(n).split("<")[0]]
                f[0] += 1; f[1] += e - s
        p("%s: %.3f ms" % (label, (b - a) / 1e6))
        for k, (c, d) in sorted(fam.items(), key=lambda kv: -kv[1][1]):
            p("    %-34s %4d launches  %.3f ms" % (k, c, d / 1e6))
    # the tail: what runs after the last conv kernel of the step
    convs = [e for _, e, n, _ in step if "conv_mfma_kernel" in n or "conv_bf16_kernel" in n]
    if convs:
        lc = max(convs)
        p("tail after the last conv kernel: %.3f ms" % ((t1 - lc) / 1e6))
        for s, e, n, q in step:
            if e > lc:
                p("    +%.3f .. +%.3f ms  q%s  %s" % ((s - t0) / 1e6, (e - t0) / 1e6, q, short(n)))


if __name__ == "__main__":
    main()
