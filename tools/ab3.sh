#!/bin/bash
# Same-box A/B/C... of environment switches on the headline step, arms alternating; each arm has its own tuning cache.
# usage: tools/ab3.sh rounds "<envA>" "<envB>" ["<envC>" ...]
R=$1; shift
mkdir -p gpurun_out
for r in $(seq 1 $R); do
  i=0
  for E in "$@"; do
    i=$((i+1))
    ms=$(env $E WUN_TUNE_CACHE=$PWD/gpurun_out/ab3_arm$i.txt python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f (median %.3f) %s' % (d['ms_per_step'], d['ms_median'], d['config']['tilings']))")
    echo "round $r arm $i [$E] $ms"
  done
done
