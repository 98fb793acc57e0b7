#!/bin/bash
# PMC passes (HBM traffic, MFMA busy) of ONE named config / dtype of bench.py, ON THE GPU BOX -- the per-config sibling of the
# PMC block of tools/profile_round.sh (same counters, same separate passes, same summariser).
# usage: tools/pmc_config.sh <tag> [bench.py args, e.g. --config baseline_stereo --dtype bf16]   -> gpurun_out/<tag>_pmc_*
set -u
tag=$1; shift
R=$PWD
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_${tag}_$c
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_${tag}_$c -o p -- python $R/bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2>&1
    python $R/tools/pmc_summarize.py $(find /tmp/pmc_${tag}_$c -name "*counter_collection.csv" | head -1) $R/gpurun_out/${tag}_pmc_$c.json
done
rm -rf /tmp/pmc_${tag}_mfma
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv \
    -d /tmp/pmc_${tag}_mfma -o p -- python $R/bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2>&1
python $R/tools/pmc_summarize.py $(find /tmp/pmc_${tag}_mfma -name "*counter_collection.csv" | head -1) $R/gpurun_out/${tag}_pmc_mfma.json
cd $R
python tools/pmc_report.py gpurun_out/${tag}_pmc_FETCH_SIZE.json gpurun_out/${tag}_pmc_WRITE_SIZE.json gpurun_out/${tag}_pmc_mfma.json \
    gpurun_out/${tag}_pmc_traffic.json gpurun_out/${tag}_pmc_mfma_util.txt "$tag: bench.py $*"
python -c "import json; d=json.load(open('gpurun_out/${tag}_pmc_traffic.json')); print('$tag step_hbm_bytes %.3f GB' % (d['step_hbm_bytes']/1e9))"
