#!/usr/bin/env python3
"""Reads the workgroup traces written by tools/diag_r3.py and prints: effective shader clock, workgroup phase
durations (prologue / chunk loop / epilogue), workgroups per CU over time and per-CU busy structure.
usage: python tools/diag_r3_report.py gpurun_out/diag/trace_<name>.npy"""
import sys
import numpy as np

for path in sys.argv[1:]:
    tr = np.load(path).astype(np.int64)
    t0, t1, t2, t3 = tr[:, 0], tr[:, 1], tr[:, 2], tr[:, 3]
    hw = tr[:, 4] & 0xFFFFFFFF
    xcc = (tr[:, 4] >> 32) & 15
    rt0, rt3 = tr[:, 5], tr[:, 6]
    cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; simd = (hw >> 4) & 3; wv = hw & 15
    cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    n = len(tr)
    # clock: shader cycles per 100 MHz tick over the whole launch (per XCC the counters may differ -> use per-WG pairs)
    dcy = (t3 - t0).astype(float); drt = (rt3 - rt0).astype(float)
    ok = drt > 50
    mhz = np.median(dcy[ok] / drt[ok]) * 100.0
    print("== %s: %d workgroups, %d distinct CUs, %d XCCs" % (path, n, len(np.unique(cuid)), len(np.unique(xcc))))
    print("   shader clock (median over workgroups): %.0f MHz" % mhz)
    us = lambda cyc: cyc / mhz
    span_rt = (rt3.max() - rt0.min()) / 100.0
    print("   launch span by the 100 MHz clock: %.1f us" % span_rt)
    extra = ()
    if tr.shape[1] >= 16:
        t7, t8, t9 = tr[:, 7], tr[:, 8], tr[:, 9]
        extra = (("  prologue: index math", t7 - t0), ("  prologue: issue first loads", t8 - t7),
                 ("  prologue: wait + LDS store", t9 - t8), ("  prologue: barrier", t1 - t9))
    for nm, d in extra + (("prologue (entry -> first chunk staged)", t1 - t0), ("chunk loop", t2 - t1), ("epilogue", t3 - t2),
                  ("whole workgroup", t3 - t0)):
        d = d.astype(float)
        print("   %-40s median %7.2f us  p10 %7.2f  p90 %7.2f  max %7.2f" % (
            nm, us(np.median(d)), us(np.percentile(d, 10)), us(np.percentile(d, 90)), us(d.max())))
    # per-CU structure in the realtime domain (common clock across XCCs)
    start = (rt0 - rt0.min()) / 100.0
    end = (rt3 - rt0.min()) / 100.0
    per_cu = {}
    for i in range(n):
        per_cu.setdefault(int(cuid[i]), []).append((start[i], end[i]))
    cnts = [len(v) for v in per_cu.values()]
    print("   workgroups per CU: min %d median %d max %d" % (min(cnts), int(np.median(cnts)), max(cnts)))
    # concurrency profile: resident workgroups per CU averaged over time
    T = end.max()
    grid = np.linspace(0, T, 400)
    res = np.zeros_like(grid)
    for s_, e_ in zip(start, end):
        res += (grid >= s_) & (grid < e_)
    res /= max(1, len(per_cu))
    print("   mean resident workgroups per CU at 0/10/25/50/75/90/100%% of the launch: %s" % " ".join(
        "%.2f" % res[int(f * 399)] for f in (0.0, 0.1, 0.25, 0.5, 0.75, 0.9, 0.995)))
    ends_sorted = np.sort(end)
    print("   last workgroup ends at %.1f us; 90%% of workgroups done at %.1f us; first ends at %.1f us" % (
        ends_sorted[-1], ends_sorted[int(0.9 * n)], ends_sorted[0]))
    # one CU's timeline
    k = sorted(per_cu)[len(per_cu) // 2]
    print("   timeline of CU %d: %s" % (k, " ".join("[%.0f-%.0f]" % se_ for se_ in sorted(per_cu[k]))))
