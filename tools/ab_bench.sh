#!/bin/bash
# Same-box A/B of two trees (ON THE GPU BOX): interleaved bench.py runs of the current tree and of a reference tree
# copied under .ab_<name>/ (ignored by git, travels with gpurun).  Box-to-box spread is several per cent, so step times of
# different builds only compare within one call.   usage: tools/ab_bench.sh <ref-dir> <rounds> [bench args...]
R=$PWD; REF=$1; N=${2:-2}; shift 2
mkdir -p gpurun_out/ab
for i in $(seq 1 $N); do
    (cd $R/$REF && python bench.py --no-cpu-baseline --no-roofline "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('ref  round $i ms_median %.4f ms_per_step %.4f' % (d['ms_median'], d['ms_per_step']))") | tee -a gpurun_out/ab/ab.txt
    (cd $R && python bench.py --no-cpu-baseline --no-roofline "$@" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('new  round $i ms_median %.4f ms_per_step %.4f' % (d['ms_median'], d['ms_per_step']))") | tee -a gpurun_out/ab/ab.txt
done
