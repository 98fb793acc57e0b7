#!/bin/bash
# Same-box A/B of library builds on the headline step: alternates the builds (DVFS / box spread cancels).
# usage: tools/ab_bench.sh "libwun.so libwun_x.so ..." [rounds] [extra bench.py args]
LIBS=${1:-"libwun.so"}; R=${2:-3}; shift 2
for r in $(seq 1 $R); do
  for L in $LIBS; do
    ms=$(WUN_LIB=$L python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f' % d['ms_per_step'])")
    echo "round $r $L $ms ms/step"
  done
done
