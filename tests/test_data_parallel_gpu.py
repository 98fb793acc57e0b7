"""Data-parallel training on the real kernels: two ranks (torch.distributed.run, one process per
rank) == one process on the concatenated batch.  The test box has one GPU, so both ranks share it and
the exchange runs over gloo (WUN_DIST_BACKEND=gloo; RCCL refuses duplicate devices) -- everything
else (bucket events recorded by the backward pass, overlapped per-bucket all-reduce on the
communication stream, 1/N folded into Adam, parameter broadcast) is the production path."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


@pytest.mark.parametrize("overlap", [True, False])
def test_two_ranks_equal_one_process_on_the_global_batch(tmp_path, overlap):
    import dp_worker
    from wave_u_net_amd import training
    steps = 3
    out = os.path.join(str(tmp_path), "dp.npz")
    env = dict(os.environ, WUN_DIST_BACKEND="gloo", WUN_NO_TUNE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if not overlap:
        env["WUN_NO_OVERLAP"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "dp_worker.py"), out, str(steps)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    dp = np.load(out)
    assert int(dp["world"]) == 2 and int(dp["overlap"]) == int(overlap)

    os.environ["WUN_NO_TUNE"] = "1"
    try:
        cfg = dict(dp_worker.make_cfg(), batch_size=6)
        tr = training.Trainer(cfg)
        mix, targets = dp_worker.global_batch(cfg, tr.t_in, tr.t_out, 6)
        mix, targets = mix.to(tr.device), targets.to(tr.device)
        losses = [float(tr.step(mix, targets).item()) for _ in range(steps)]
        torch.cuda.synchronize()
    finally:
        os.environ.pop("WUN_NO_TUNE", None)
    ref = tr.sep.params.cpu().numpy()
    # rank 0's loss is the mean over ITS half of the batch; parameters see the global gradient
    assert np.abs(dp["params"] - ref).max() <= 2e-5 * max(1.0, np.abs(ref).max())
    assert np.all(np.isfinite(dp["losses"])) and dp["losses"][-1] < dp["losses"][0] and losses[-1] < losses[0]


def test_two_rank_resume_keeps_replicas_identical(tmp_path):
    """train() -> checkpoint -> train(load_model=...) on two ranks (the flow of validation.optimise):
    rank 0 loads, parameters / Adam slots / global_step are broadcast, the checkpoint path reaches every
    rank, and rank 0's tuning table is the one every rank imports."""
    out = os.path.join(str(tmp_path), "resume")
    env = dict(os.environ, WUN_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WUN_NO_TUNE", None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(ROOT, "tests", "dp_worker.py"), out, "resume"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    a, b = np.load(out + ".rank0.npz"), np.load(out + ".rank1.npz")
    assert int(a["step"]) == 4 and int(b["step"]) == 4
    for k in ("params", "m", "v"):
        assert np.array_equal(a[k], b[k]), k                 # bit-identical replicas (same tilings on both ranks)
    assert str(a["table"]) == str(b["table"]) and str(a["table"]).startswith("wun-tune 2 ")


def test_bench_launches_itself_two_ranks_on_one_gpu():
    """`python bench.py --gpus 2` with no launcher around it (the driver's command shape): bench.py re-executes itself
    under torch.distributed.run, both ranks share this box's GPU (exchange over gloo), rank 0 prints ONE JSON line
    with n_gpus = 2 and the per-step statistics.  Tiny config so the run takes seconds."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(WUN_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "full", "--batch", "3",
           "--set", "num_layers=4", "--set", "num_initial_filters=8", "--set", "num_frames=72",
           "--steps", "4", "--warmup", "2", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    d = lines[0]
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 4 and d["config"]["global_batch"] == 6
    assert d["config"]["parallelism"] == "dp2"
    assert d["value"] > 0 and d["ms_p10"] <= d["ms_median"] <= d["ms_p90"]
    assert np.isfinite(d["config"]["final_loss"])
    # the line says WHY the scaling is what it is: communication the overlap did not hide, and the same step without
    # the all-reduce
    c = d["comm"]
    assert c["buckets"] >= 1 and c["bytes"] == sum(c["bucket_bytes"]) and c["bytes"] > 0
    assert c["exposed_ms"] >= 0.0 and c["overlapped"] is True and c["backend"] == "gloo"
    assert 0.0 < d["ms_per_step_no_comm"] <= 1.5 * d["ms_per_step"] + 5.0


@pytest.mark.gpu
def test_bench_two_ranks_headline_config_reduced_width():
    """The driver's multi-GPU command on the headline configuration's wiring (M1 + context, reduced to 4 levels / 8
    filters so two ranks fit one GPU in seconds): `python bench.py --gpus 2 --config m1_context ...` end to end --
    self-launch under torch.distributed.run, tuning-table broadcast, bucketed gradient exchange, one JSON line."""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(WUN_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--config", "m1_context",
           "--set", "num_layers=4", "--set", "num_initial_filters=8", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [json.loads(ln) for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    d = lines[0]
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 3
    assert d["config"]["parallelism"] == "dp2" and d["config"]["global_batch"] == 32
    assert d["metric"].startswith("waveform samples/sec") and d["value"] > 0
    assert np.isfinite(d["config"]["final_loss"])
