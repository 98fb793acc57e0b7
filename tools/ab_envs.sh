#!/bin/bash
# Same-box A/B of environment switches on the headline step with the PINNED tilings, alternating arms.
# usage: tools/ab_envs.sh rounds "<envA>" "<envB>" ["<envC>" ...]    ("X=1" = a no-op arm)
R=$1; shift
for r in $(seq 1 $R); do
  for E in "$@"; do
    ms=$(env $E python bench.py --steps 40 --warmup 8 --no-cpu-baseline --no-roofline ${BENCH_ARGS:-} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f (median %.3f) %s' % (d['ms_per_step'], d['ms_median'], d['config']['tilings']))")
    echo "round $r [$E] $ms"
  done
done
