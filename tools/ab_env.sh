#!/bin/bash
# Same-box A/B of environment switches on the headline step, arms alternating (box-to-box variance is 3-10 %: only
# arms measured on one box, interleaved, compare).
# usage: tools/ab_env.sh rounds "<envA>" "<envB>" ...      (use "X=" for an empty arm)
#   TUNE=1 tools/ab_env.sh ...   every arm autotunes into its own table first (for switches that change launches);
#                                default: all arms run the pinned table of profiles/
R=$1; shift
mkdir -p gpurun_out
for r in $(seq 1 $R); do
  i=0
  for E in "$@"; do
    i=$((i+1))
    CACHE=""
    if [ "${TUNE:-0}" = "1" ]; then CACHE="WUN_TUNE_CACHE=$PWD/gpurun_out/ab_arm$i.txt"; fi
    ms=$(env $E $CACHE python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | grep '^{"metric"' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f (median %.3f) %s' % (d['ms_per_step'], d['ms_median'], d['config']['tilings']))")
    echo "round $r arm $i [$E] $ms"
  done
done
