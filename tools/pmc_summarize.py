#!/usr/bin/env python3
"""Aggregate a rocprofv3 --pmc counter_collection CSV per kernel for the LAST step only
(dispatches from the last btc_to_ncw_kernel on), so autotuning / warm-up launches are excluded.
usage: python tools/pmc_summarize.py <counter_collection.csv> <out.json>"""
import collections
import csv
import json
import re
import sys


def canon(name):
    """conv_mfma_kernel<4, 3, 4, 1, 8, true, false>(...) -> conv_mfma_kernel<4, 3, 4, 1, 8>; batch-folded
    instantiations get a ', fold' marker.  Same rule as bench.py's pmc_traffic()."""
    nm = re.sub(r"\(.*", "", name.replace("void wun::", "").replace("wun::", ""))
    m = re.match(r"conv_mfma_kernel<(\d+, \d+, \d+, \d+, \d+), (?:true|false)(?:, (true|false|fold))?>", nm)
    if m:
        return "conv_mfma_kernel<%s%s>" % (m.group(1), ", fold" if m.group(2) in ("true", "fold") else "")
    return nm


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    first = max(int(r["Dispatch_Id"]) for r in rows if "btc_to_ncw" in r["Kernel_Name"])
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    cnt = collections.defaultdict(set)
    dur = collections.defaultdict(float)
    seen = set()
    for r in rows:
        d = int(r["Dispatch_Id"])
        if d < first:
            continue
        nm = canon(r["Kernel_Name"])
        agg[nm][r["Counter_Name"]] += float(r["Counter_Value"])
        cnt[nm].add(d)
        if d not in seen:
            seen.add(d)
            dur[nm] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    out = {k: dict(v, launches=len(cnt[k]), duration_ns=dur[k]) for k, v in agg.items()}
    json.dump(out, open(sys.argv[2], "w"), indent=1)
    print("kernels:", len(out), "dispatches in last step:", len(seen))


if __name__ == "__main__":
    main()
