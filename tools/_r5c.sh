mkdir -p gpurun_out/r5c
timeout 900 python -m pytest tests/test_gpu_bf16.py -q -x > gpurun_out/r5c/pytest_bf16.log 2>&1; echo "rc=$?" > gpurun_out/r5c/rc.txt
tail -25 gpurun_out/r5c/pytest_bf16.log
