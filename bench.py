#!/usr/bin/env python3
"""Headline benchmark: waveform samples/sec, forward + backward (+ Adam, + gradient all-reduce
when N > 1) of the M1 12-level Wave-U-Net at BASELINE.json configs[1]: fp32, batch 16 per GPU,
~147k-sample context input (147443 -> 16389 samples), synthetic waveforms resident in HBM.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py ...`)

One "step" = one `sess.run([separator_solver, ...])` of /root/reference/Training.py:105:
get_output, MSE loss, full backward, TF-Adam update.  Rank 0 prints ONE JSON line.
`value` = output samples/s of the whole job (N * B * Tout * K / max-over-ranks time);
the input-sample rate (N * B * Tin) is reported in `config` for reference.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np   # noqa: E402
import torch         # noqa: E402
import torch.distributed as dist   # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3     # /opt/skills/guides/MI355X_MICROARCH.md: dense fp32 MFMA = fp32 vector peak
_T0 = time.time()


def log(msg):
    if os.environ.get("RANK", "0") == "0":
        print("[bench %7.1fs] %s" % (time.time() - _T0, msg), file=sys.stderr, flush=True)


def usable_cores():
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota
    (os.cpu_count() reports the host's CPUs even inside a limited container)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
        except Exception:
            pass
    return max(1, n)


def pmc_traffic(kernel_name):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/round1_pmc_traffic.json: FETCH_SIZE doubled per the gfx950 note + WRITE_SIZE,
    separate passes); PMC counters cannot be collected from inside this process."""
    path = os.path.join(ROOT, "profiles", "round1_pmc_traffic.json")
    try:
        table = json.load(open(path))["kernels"]
    except Exception:
        return None
    import re
    m = re.match(r"conv_mfma_kernel<(\d+, \d+, \d+, \d+, \d+), (?:true|false)(?:, (true|false|fold))?>", kernel_name)
    key = ("conv_mfma_kernel<%s%s>" % (m.group(1), ", fold" if m.group(2) in ("true", "fold") else "")) if m else kernel_name
    ent = table.get(key)
    return ent["hbm_bytes_per_launch"] if ent else None


def cpu_baseline(cfg_name, cfg_over, budget_s=20.0):
    """The oracle (torch-CPU fp32 restatement of the reference graph; TensorFlow 1.8 cannot be
    installed) timed on this box's host cores on a bounded sample of the same workload."""
    from oracle import shapes, waveunet_torch as wt       # checker / CPU baseline only
    ocfg = shapes.finalize_config(dict(shapes.BASE_MODEL_CONFIG, **cfg_over))
    cores = usable_cores()
    log("cpu baseline: os.cpu_count=%s usable=%d torch default threads=%d" % (os.cpu_count(), cores, torch.get_num_threads()))
    torch.set_num_threads(cores)
    B = 1
    i, o = shapes.get_padding(ocfg, [B, ocfg["num_frames"], 0])
    mix, targets = wt.synthetic_batch(ocfg, B, i[1], o[1], seed=1337)
    tp = wt.params_to_torch(wt.init_params(ocfg, 1337), torch.float32, requires_grad=True)
    m = [torch.zeros_like(p) for _, p in tp]
    v = [torch.zeros_like(p) for _, p in tp]
    tmix = torch.from_numpy(mix)
    ttg = {k: torch.from_numpy(x) for k, x in targets.items()}
    t0 = time.time()
    wt.train_step(ocfg, tp, tmix, ttg, m, v, 1, 1e-4)      # warm-up (also the fallback timing)
    warm = time.time() - t0
    log("cpu baseline: warm-up step %.2f s" % warm)
    times = []
    t_start = time.time()
    while len(times) < 20 and (time.time() - t_start) + (times[-1] if times else warm) < budget_s:
        t1 = time.time()
        wt.train_step(ocfg, tp, tmix, ttg, m, v, len(times) + 2, 1e-4)
        times.append(time.time() - t1)
    dt = float(np.median(times)) if times else warm
    log("cpu baseline: %d timed steps, median %.2f s" % (len(times), dt))
    return {"value": B * o[1] / dt, "unit": "output samples/s", "cores": cores, "kind": "port",
            "sample": "%d timed step(s) of batch %d (of 16) excerpt(s) %d->%d, torch-CPU fp32 oracle, "
                      "fwd+bwd+Adam, %.2f s/step%s" % (len(times), B, i[1], o[1], dt,
                                                       "" if times else " (warm-up step only)"),
            "threads": torch.get_num_threads()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--config", default="m1_context")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    args = ap.parse_args()

    import wave_u_net_amd as wun
    from wave_u_net_amd import _lib
    from wave_u_net_amd.training import Trainer, synthetic_source

    cfg = wun.get_config(args.config)
    log("building trainer")
    tr = Trainer(cfg, batch_size=args.batch)
    world = tr.world
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    source = synthetic_source(cfg, tr.batch, tr.t_in, tr.t_out, tr.device, seed=1337 + tr.rank)
    mix, targets = source()
    log("data ready: mix %s targets %s" % (tuple(mix.shape), tuple(targets.shape)))

    def barrier():
        if world > 1:
            dist.barrier()

    t_tune = time.time()
    tr.tune(mix, targets)                      # one-off autotuning of the per-launch tilings (untimed)
    log("autotune: %.2f s" % (time.time() - t_tune))
    for _ in range(args.warmup):
        loss = tr.step(mix, targets)
    torch.cuda.synchronize()
    log("warm-up done")
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = tr.step(mix, targets)
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=tr.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss_val = float(loss.item())
    log("timed region: %d steps in %.3f s (%.2f ms/step)" % (args.steps, elapsed, 1e3 * elapsed / args.steps))

    info = tr.sep.plan_info()
    ms_per_step = 1e3 * elapsed / args.steps
    out_samples = world * tr.batch * tr.t_out
    result = {
        "metric": "waveform samples/sec fwd+bwd, M1 12-level Wave-U-Net @1/2/4/8 GPU",
        "value": out_samples * args.steps / elapsed,
        "unit": "output samples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",
        "config": {"workload": ("BASELINE.json configs[1]: M1 (12 levels, 24 ch, 15/5 filters, mono) with context, "
                                "fwd+bwd+Adam, batch %d/GPU, %d -> %d samples" % (tr.batch, tr.t_in, tr.t_out))
                               if args.config == "m1_context" else
                               ("named config %s (%d levels, %d base channels, %s, %d sources, %s padding), fwd+bwd+Adam, "
                                "batch %d/GPU, %d -> %d samples" % (
                                    args.config, cfg["num_layers"], cfg["num_initial_filters"],
                                    "mono" if cfg["mono_downmix"] else "stereo", cfg["num_sources"],
                                    "valid (context)" if cfg["context"] else "same", tr.batch, tr.t_in, tr.t_out)),
                   "named_config": args.config, "global_batch": world * tr.batch,
                   "input_frames": tr.t_in, "output_frames": tr.t_out,
                   "input_samples_per_s": world * tr.batch * tr.t_in * args.steps / elapsed,
                   "parallelism": "dp%d" % world, "final_loss": loss_val,
                   "step_tflops_executed": (info.fwd_flops + info.bwd_flops) / 1e12,
                   "step_tflops_reference_graph": 3.0 * info.fwd_flops_dense / 1e12,
                   "achieved_tflops_executed": (info.fwd_flops + info.bwd_flops) / (ms_per_step * 1e9)},
    }

    if not args.no_roofline:
        # a few extra steps with HIP events around every heavy launch (all ranks step together,
        # rank 0 records): kept outside the timed region so the events do not perturb `value`
        lib = _lib.load()
        nprof = 2
        if tr.rank == 0:
            lib.wun_profile_begin()
        for _ in range(nprof):
            tr.step(mix, targets)
        torch.cuda.synchronize()
        if tr.rank == 0:
            buf = ctypes.create_string_buffer(1 << 20)
            _lib.check(lib.wun_profile_end(buf, len(buf)))
            prof = json.loads(buf.value.decode())
            if prof.get("launches") and os.environ.get("WUN_PROFILE_DETAIL"):
                with open(os.environ["WUN_PROFILE_DETAIL"], "w") as f:
                    json.dump(prof["launches"], f)
            kernels = prof["kernels"]
            kernels.sort(key=lambda k: -k["ms"])
            for k in kernels:
                log("  %-42s launches/step %5.1f  ms/step %8.3f  TFLOP/s %7.2f" % (
                    k["name"], k["launches"] / nprof, k["ms"] / nprof,
                    k["flops"] / max(k["ms"], 1e-9) / 1e9))
            top = kernels[0]
            avg_ms = top["ms"] / top["launches"]
            achieved = top["flops"] / top["launches"] / (avg_ms * 1e-3) / 1e12
            result["roofline"] = {
                "bound": "mfma", "kernel": top["name"], "achieved": achieved, "peak": PEAK_FP32_MFMA_TFLOPS,
                "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_MFMA_TFLOPS, "traffic": pmc_traffic(top["name"]),
                "avg_launch_ms": avg_ms, "launches_per_step": top["launches"] / nprof,
                "flops_per_launch": top["flops"] / top["launches"],
                "kernel_ms_per_step": {k["name"]: k["ms"] / nprof for k in kernels},
                "kernel_tflops": {k["name"]: k["flops"] / (k["ms"] * 1e-3) / 1e12 for k in kernels if k["ms"] > 0},
            }

    if tr.rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(args.config, wun.NAMED_CONFIGS[args.config])
        result["config"]["speedup_vs_cpu_baseline"] = result["value"] / result["cpu_baseline"]["value"]

    if tr.rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
