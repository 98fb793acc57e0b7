mkdir -p gpurun_out/r5e gpurun_out/ab; rm -f gpurun_out/ab/ab.txt
timeout 900 python -m pytest tests/test_gpu_bf16.py -q -x > gpurun_out/r5e/pytest_bf16.log 2>&1; echo "rc=$?" > gpurun_out/r5e/rc.txt
tail -5 gpurun_out/r5e/pytest_bf16.log
echo "== baseline_stereo bf16" >> gpurun_out/ab/ab.txt; tools/ab_bench.sh .ab_r4 2 --config baseline_stereo --dtype bf16 >/dev/null 2>&1
echo "== deep bf16 (heuristic tilings)" >> gpurun_out/ab/ab.txt; WUN_NO_TUNE=1 tools/ab_bench.sh .ab_r4 1 --config deep_l16_f48 --dtype bf16 --steps 5 --warmup 2 >/dev/null 2>&1
cat gpurun_out/ab/ab.txt
python bench.py --config baseline_stereo --dtype bf16 --no-cpu-baseline > gpurun_out/r5e/cfg_baseline_stereo_bf16.json 2> gpurun_out/r5e/cfg_baseline_stereo_bf16.err
WUN_NO_TUNE=1 python bench.py --config deep_l16_f48 --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r5e/cfg_deep_bf16.json 2> gpurun_out/r5e/cfg_deep_bf16.err
WUN_TUNE_CACHE=$PWD/gpurun_out/r5e/deep_bf16_table.txt timeout 600 python bench.py --config deep_l16_f48 --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r5e/cfg_deep_bf16_tuned.json 2> gpurun_out/r5e/cfg_deep_bf16_tuned.err
cut -c1-300 gpurun_out/r5e/cfg_deep_bf16_tuned.json
