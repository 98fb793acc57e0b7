#!/usr/bin/env python3
"""Per-kernel listing of one step from a rocprofv3 --kernel-trace CSV (start / end / duration us, queue, kernel, workgroups).
usage: tools/trace_step.py <kernel_trace.csv> [steps_from_end=2] [from_us] [to_us]"""
import csv, re, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0"),
                 r.get("Grid_Size_X", r.get("Grid_Size", "")), r.get("Workgroup_Size_X", "")))
rows.sort()
back = int(sys.argv[2]) if len(sys.argv) > 2 else 2
lo_us = float(sys.argv[3]) if len(sys.argv) > 3 else -1e30
hi_us = float(sys.argv[4]) if len(sys.argv) > 4 else 1e30
adam = [i for i, r in enumerate(rows) if "adam_kernel" in r[2]]
lo, hi = adam[-back - 1] + 1, adam[-back] + 1
t0 = rows[adam[-back - 1]][1]
def short(name):
    name = re.sub(r"^void ", "", name).replace("wun::", "")
    m = re.match(r"([a-z0-9_]+)(<[^>]*>)?", name)
    s = (m.group(1) + (m.group(2) or "")) if m else name[:40]
    return s.replace("conv_mfma_kernel", "conv").replace("wgrad_mfma_kernel", "wgrad").replace(", true", "").replace(", false", "")
qs = sorted(set(r[3] for r in rows[lo:hi]))
for s, e, n, q, g, w in rows[lo:hi]:
    a, b = (s - t0) / 1e3, (e - t0) / 1e3
    if b < lo_us or a > hi_us: continue
    print("%8.1f %8.1f %6.1f q%d %-34s wgs %s" % (a, b, b - a, qs.index(q), short(n), int(g) // max(1, int(w or 1)) if g else ""))
print("step wall %.1f us, %d kernels" % ((rows[hi - 1][1] - t0) / 1e3, hi - lo))
