mkdir -p gpurun_out
( for e in "X=1" "X=2" "X=3" "WUN_BF16_HEAD_OVERLAP=1"; do echo "==== bf16 $e"; env $e python tools/repro_probe.py bf16 2>&1 | grep -v amdgpu.ids | cut -c1-160; done
  for e in "X=1" "X=2" "X=3"; do echo "==== f32 $e"; env $e python tools/repro_probe.py f32 2>&1 | grep -v amdgpu.ids | cut -c1-160; done ) > gpurun_out/repro_probe.log 2>&1
cat gpurun_out/repro_probe.log | tail -80
(time timeout 900 python -m pytest tests/test_gpu_bf16.py -q -k "reproducible or full_size or train_step_small" 2>&1 | tail -15) > gpurun_out/bf16_emul_test3.log 2>&1
tail -8 gpurun_out/bf16_emul_test3.log
