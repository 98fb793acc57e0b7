#!/bin/bash
# Step time of the headline workload with the N>1 gradient all-reduce machinery running on a 1-rank RCCL group
# (bench.py --force-allreduce): the collective kernels launch per bucket on the comm stream and contend with the
# backward pass exactly as at N>1, only the wire time is missing.  Output: gpurun_out/<round>_force_allreduce.txt
set -u
R=${1:-round4}
OUT=gpurun_out/${R}_force_allreduce.txt
mkdir -p gpurun_out
: > $OUT
run() {  # label, extra args
  local label=$1; shift
  local line
  line=$(python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-roofline "$@" 2>gpurun_out/${R}_far.err | grep "^{\"metric\"" | tail -1)
  python - "$label" "$line" >> $OUT <<'PY'
import json, sys
d = json.loads(sys.argv[2])
print("%-34s ms_per_step %.3f  median %.3f  value %.3e %s" % (sys.argv[1], d["ms_per_step"], d["ms_median"], d["value"], d["unit"]))
PY
  grep -h "forced all-reduce" gpurun_out/${R}_far.err >> $OUT
}
run "no all-reduce (plain N=1 step)"
run "forced, 1 bucket"            --force-allreduce --bucket-mib 4096
run "forced, 16 MiB buckets"      --force-allreduce --bucket-mib 16
run "forced, 5 MiB buckets"       --force-allreduce --bucket-mib 5
run "no all-reduce (repeat)"
cat $OUT
