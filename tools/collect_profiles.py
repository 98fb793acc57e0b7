#!/usr/bin/env python3
"""Copy the artefacts of `tools/profile_round.sh <tag>` from gpurun_out/ (scratch) into profiles/ (tracked), putting a
header line on the two rocprofv3 kernel-stats CSVs that says which command they profile and how many training steps
they contain (counted from the adam_kernel calls).   usage: python tools/collect_profiles.py round5"""
import csv
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, dst = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")


def steps_of(path):
    rows = list(csv.DictReader(open(path)))
    return sum(int(r["Calls"]) for r in rows if "adam_kernel" in r["Name"])


def settle_of(path):
    try:
        return json.loads(open(path).read().strip().split("\n")[-1])["config"].get("settle_steps", 0)
    except Exception:
        return 0


for name, bench, what in (("kernel_stats.csv", "prof_bench.json", "three streams, as in the timed run"),
                          ("kernel_stats_single_stream.csv", "prof1_bench.json",
                           "every launch on ONE stream (WUN_SINGLE_STREAM=1): per-kernel durations free of overlap")):
    p = os.path.join(src, "%s_%s" % (tag, name))
    n, settle = steps_of(p), settle_of(os.path.join(src, "%s_%s" % (tag, bench)))
    head = ("# rocprofv3 --kernel-trace --stats of `python bench.py --steps 20 --warmup 5 --no-cpu-baseline` "
            "(tools/profile_round.sh %s); pinned tilings profiles/%s_tune_table.txt; %s; %d training steps in the file = "
            "5 warm-up + %d settle + 20 timed + 2 event-bracketed roofline steps; no tuning pass (the table is imported)\n"
            % (tag, tag, what, n, settle))
    with open(os.path.join(dst, "%s_%s" % (tag, name)), "w") as f:
        f.write(head)
        f.write(open(p).read())
for pat in ("bench.json", "bench.err", "prof_bench.json", "prof1_bench.json", "timeline.txt", "pmc_traffic.json",
            "pmc_mfma_util.txt", "cfg_*.json"):
    for p in glob.glob(os.path.join(src, "%s_%s" % (tag, pat))):
        shutil.copy(p, dst)
if os.path.exists(os.path.join(src, "parity_observed.json")):
    shutil.copy(os.path.join(src, "parity_observed.json"), os.path.join(dst, "%s_parity_observed.json" % tag))
print("copied", tag)
