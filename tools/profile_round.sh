#!/bin/bash
# Run ON THE GPU BOX (via gpurun) from the repo root: final bench line, rocprofv3 kernel stats of the
# same command with the tuning table reused, and the PMC passes.  Outputs under gpurun_out/<tag>_*.
# usage: tools/profile_round.sh <tag>
set -u
tag=${1:-round}
R=$PWD
mkdir -p gpurun_out
# tilings: bench.py imports the committed pinned table of the config by itself (read-only); PINNED=0 -> fresh tuning
PIN=$R/profiles/round6_tune_table.txt
if [ "${PINNED:-1}" != 1 ]; then
    export WUN_TUNE_CACHE=$R/gpurun_out/${tag}_tune_table.txt
    [ "${KEEP_TUNE:-0}" = 1 ] || rm -f $WUN_TUNE_CACHE
    PIN=$WUN_TUNE_CACHE
fi
if [ "${ONLYCFGS:-0}" != 1 ]; then
python bench.py > gpurun_out/${tag}_bench.json 2> gpurun_out/${tag}_bench.err
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_${tag}
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_${tag} -o ks -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline \
    > $R/gpurun_out/${tag}_prof_bench.json 2> $R/gpurun_out/${tag}_prof_bench.err
cp $(find /tmp/prof_${tag} -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${tag}_kernel_stats.csv
cp $(find /tmp/prof_${tag} -name "*kernel_trace.csv" | head -1) $R/gpurun_out/${tag}_kernel_trace.csv
python $R/tools/timeline.py $R/gpurun_out/${tag}_kernel_trace.csv 6 $R/gpurun_out/${tag}_timeline.txt
# same command with every launch on one stream: per-kernel durations free of overlap with the other
# streams' kernels -- the figure bench.py's roofline (HIP events, single-stream profiled steps) must agree with
rm -rf /tmp/prof1_${tag}
WUN_SINGLE_STREAM=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof1_${tag} -o ks -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline \
    > $R/gpurun_out/${tag}_prof1_bench.json 2> $R/gpurun_out/${tag}_prof1_bench.err
cp $(find /tmp/prof1_${tag} -name "*kernel_stats.csv" | head -1) $R/gpurun_out/${tag}_kernel_stats_single_stream.csv
for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$c
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2>&1
    python $R/tools/pmc_summarize.py $(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1) $R/gpurun_out/${tag}_pmc_$c.json
done
rm -rf /tmp/pmc_mfma
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv \
    -d /tmp/pmc_mfma -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > /dev/null 2>&1
python $R/tools/pmc_summarize.py $(find /tmp/pmc_mfma -name "*counter_collection.csv" | head -1) $R/gpurun_out/${tag}_pmc_mfma.json
cd $R
python tools/pmc_report.py gpurun_out/${tag}_pmc_FETCH_SIZE.json gpurun_out/${tag}_pmc_WRITE_SIZE.json gpurun_out/${tag}_pmc_mfma.json \
    gpurun_out/${tag}_pmc_traffic.json gpurun_out/${tag}_pmc_mfma_util.txt "$tag" $PIN
tail -1 gpurun_out/${tag}_bench.json | cut -c1-400
fi
cd $R
# the other BASELINE.json configs, fp32 and the bf16 speed mode (named-config benches; no CPU baseline)
if [ "${CFGS:-1}" = 1 ]; then
    unset WUN_TUNE_CACHE
    for c in baseline baseline_stereo full full_multi_instrument; do
        # configs[0] (SURVEY 8d's PR1 CPU-baseline config) and configs[2] carry `cpu_baseline` and `parity` like the headline
        # line (~1 s / ~3 s of CPU work each); the others skip the CPU leg
        nocpu=--no-cpu-baseline
        case $c in baseline|baseline_stereo) nocpu= ;; esac
        python bench.py --config $c $nocpu > gpurun_out/${tag}_cfg_${c}_f32.json 2> gpurun_out/${tag}_cfg_${c}_f32.err
        python bench.py --config $c --dtype bf16 --no-cpu-baseline > gpurun_out/${tag}_cfg_${c}_bf16.json 2> gpurun_out/${tag}_cfg_${c}_bf16.err
    done
    python bench.py --dtype bf16 --no-cpu-baseline > gpurun_out/${tag}_cfg_m1_context_bf16.json 2> gpurun_out/${tag}_cfg_m1_context_bf16.err
    WUN_NO_TUNE=1 python bench.py --config deep_l16_f48 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${tag}_cfg_deep_f32.json 2> gpurun_out/${tag}_cfg_deep_f32.err
    python bench.py --config deep_l16_f48 --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/${tag}_cfg_deep_bf16.json 2> gpurun_out/${tag}_cfg_deep_bf16.err   # (autotuned: the bf16 tile menu is small)
    for f in gpurun_out/${tag}_cfg_*.json; do echo "$f: $(cut -c1-200 $f)"; done
fi
