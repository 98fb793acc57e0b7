#!/bin/bash
# FETCH_SIZE (KiB) of one operator launch under a given library build: tools/probes/fetch_probe.sh <lib.so> <op_bench args...>
lib=$1; shift
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/fp_$$
WUN_LIB=$lib rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/fp_$$ -o p -- python $R/tools/op_bench.py "$@" 3 > /dev/null 2>&1
python - <<PY
import csv, glob, collections
f = glob.glob("/tmp/fp_$$/**/*counter_collection.csv", recursive=True)[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "mfma_kernel" in r["Kernel_Name"]:
        agg[r["Kernel_Name"].split("(")[0][-40:]].append(float(r["Counter_Value"]))
for k, v in agg.items():
    print("$lib", k, "launches", len(v), "FETCH_SIZE x2 = %.1f MB/launch" % (2 * sum(v) / len(v) * 1024 / 1e6))
PY
