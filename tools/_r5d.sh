mkdir -p gpurun_out/r5d; rm -f gpurun_out/ab/ab.txt
echo "== baseline_stereo bf16" >> gpurun_out/ab/ab.txt; tools/ab_bench.sh .ab_r4 2 --config baseline_stereo --dtype bf16 >/dev/null 2>&1
echo "== m1_context bf16" >> gpurun_out/ab/ab.txt; tools/ab_bench.sh .ab_r4 1 --dtype bf16 >/dev/null 2>&1
echo "== full bf16" >> gpurun_out/ab/ab.txt; tools/ab_bench.sh .ab_r4 1 --config full --dtype bf16 >/dev/null 2>&1
echo "== deep bf16 (heuristic tilings)" >> gpurun_out/ab/ab.txt; WUN_NO_TUNE=1 tools/ab_bench.sh .ab_r4 1 --config deep_l16_f48 --dtype bf16 --steps 5 --warmup 2 >/dev/null 2>&1
cat gpurun_out/ab/ab.txt
python bench.py --config baseline_stereo --dtype bf16 --no-cpu-baseline > gpurun_out/r5d/cfg_baseline_stereo_bf16.json 2> gpurun_out/r5d/cfg_baseline_stereo_bf16.err
WUN_NO_TUNE=1 python bench.py --config deep_l16_f48 --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r5d/cfg_deep_bf16.json 2> gpurun_out/r5d/cfg_deep_bf16.err
tail -3 gpurun_out/r5d/cfg_deep_bf16.err
