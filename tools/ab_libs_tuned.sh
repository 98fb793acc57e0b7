#!/bin/bash
# Same-box A/B of library builds, each with its OWN freshly tuned table (the builds differ in what the tuner can choose).
# usage: tools/ab_libs_tuned.sh "libA.so libB.so" [rounds]
LIBS=${1:-"libwun.so"}; R=${2:-2}
mkdir -p gpurun_out
for L in $LIBS; do rm -f gpurun_out/ab_tuned_$L.txt; done
for r in $(seq 1 $R); do
  for L in $LIBS; do
    ms=$(WUN_LIB=$L WUN_TUNE_CACHE=$PWD/gpurun_out/ab_tuned_$L.txt python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.3f (median %.3f) %s' % (d['ms_per_step'], d['ms_median'], d['config']['tilings']))")
    echo "round $r $L $ms"
  done
done
