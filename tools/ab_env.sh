#!/bin/bash
# Same-box A/B of environment switches on the headline step, alternating arms; each arm has its own tuning cache.
# usage: tools/ab_env.sh "<envA>" "<envB>" [rounds]      e.g. tools/ab_env.sh "X=1" "WUN_NO_EARLY_WINDOW=1" 3
A="$1"; B="$2"; R=${3:-3}
mkdir -p gpurun_out
for r in $(seq 1 $R); do
  for arm in A B; do
    if [ $arm = A ]; then E="$A"; else E="$B"; fi
    ms=$(env $E WUN_TUNE_CACHE=$PWD/gpurun_out/ab_env_$arm.txt python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f (median %.3f) %s' % (d['ms_per_step'], d['ms_median'], d['config']['tilings']))")
    echo "round $r arm $arm [$E] $ms"
  done
done
