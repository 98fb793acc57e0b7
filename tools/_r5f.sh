mkdir -p gpurun_out/r5f
timeout 900 python -m pytest tests/test_data_parallel_gpu.py tests/test_gpu_bf16.py -q -x -k "bench or small or m4 or deep_variant_l16_f48 and heuristic or two_ranks" > gpurun_out/r5f/pytest.log 2>&1; echo "rc=$?" > gpurun_out/r5f/rc.txt
tail -4 gpurun_out/r5f/pytest.log
for i in 1 2; do
python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('plain              ms_median %.4f' % d['ms_median'])"
python bench.py --force-allreduce --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('forced normal/normal ms_median %.4f exposed %.4f nocomm %.4f buckets %d' % (d['ms_median'], d['comm']['exposed_ms'], d['ms_per_step_no_comm'], d['comm']['buckets']))"
WUN_SIDE_PRIO=low WUN_COMM_PRIO=high python bench.py --force-allreduce --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('forced low/high      ms_median %.4f exposed %.4f nocomm %.4f' % (d['ms_median'], d['comm']['exposed_ms'], d['ms_per_step_no_comm']))"
WUN_COMM_PRIO=high python bench.py --force-allreduce --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('forced normal/high   ms_median %.4f exposed %.4f nocomm %.4f' % (d['ms_median'], d['comm']['exposed_ms'], d['ms_per_step_no_comm']))"
done 2>&1 | tee gpurun_out/r5f/force_allreduce.txt
WUN_NO_TUNE=1 python bench.py --config deep_l16_f48 --dtype bf16 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('deep bf16 ms', d['ms_per_step'], {k:round(v,2) for k,v in d['roofline']['family_ms_per_step'].items()})"
