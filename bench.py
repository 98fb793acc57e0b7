#!/usr/bin/env python3
"""Headline benchmark: waveform samples/sec, forward + backward (+ Adam, + gradient all-reduce
when N > 1) of the M1 12-level Wave-U-Net at BASELINE.json configs[1]: fp32, batch 16 per GPU,
~147k-sample context input (147443 -> 16389 samples), synthetic waveforms resident in HBM.

  python bench.py --gpus N --steps K --warmup W
  N > 1: either launched by `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`
  (the driver's form: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from the environment) or started bare --
  `python bench.py --gpus N` re-executes itself under torch.distributed.run on 127.0.0.1 (one rank per GPU;
  on a box with fewer GPUs than ranks the ranks share devices and the exchange runs over gloo).
  --dry-run stops after process-group initialisation and the tuning-table broadcast (works without a GPU).

One "step" = one `sess.run([separator_solver, ...])` of /root/reference/Training.py:105:
get_output, MSE loss, full backward, TF-Adam update.  Rank 0 prints ONE JSON line.
`value` = output samples/s of the whole job (N * B * Tout * K / max-over-ranks time);
the input-sample rate (N * B * Tin) is reported in `config` for reference.

Tilings: the per-launch tile table is read (never written) from the committed profiles/*_tune_table.txt of
the named config when it matches this plan and library build, so the timed launches are the ones the
committed rocprofv3 / PMC profiles describe; otherwise rank 0 autotunes (untimed) and broadcasts.
WUN_TUNE_CACHE=<file> is a separate, writable cache.
Extra objects in the line: `roofline` (kernel FAMILY conv_mfma_kernel, HIP events on the launch
stream), `parity` (step 0 of the timed inputs vs the fp32 oracle), `cpu_baseline` (that oracle timed
on the host cores, full batch + a 1-thread figure).
"""
import argparse
import ctypes
import gc
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
PEAK_HBM_GBPS = 8000.0             # same guide: HBM3E 8 TB/s spec (a float4 copy reaches 6.3 TB/s)
PEAK_BF16_MFMA_TFLOPS = 2500.0    # same guide: dense bf16 MFMA (the 5 PF headline figure includes 2:1 sparsity)


def family_peak(family):
    return PEAK_BF16_MFMA_TFLOPS if "_bf16_" in family else PEAK_FP32_MFMA_TFLOPS
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


# committed, READ-ONLY tuning tables per (named config, batch, dtype); never used as a writable cache
PINNED_TUNE_TABLES = {
    ("m1_context", 16, "f32"): os.path.join(ROOT, "profiles", "round6_tune_table.txt"),
    ("baseline", 16, "f32"): os.path.join(ROOT, "profiles", "round6_tune_table_baseline.txt"),
}
PINNED_TUNE_TABLE = PINNED_TUNE_TABLES[("m1_context", 16, "f32")]


def relaunch_under_torchrun(n):
    """`python bench.py --gpus N` without a launcher: re-execute under torch.distributed.run, one rank per GPU of
    this node, rendezvous on 127.0.0.1 (the reference has no multi-device code: Training.py:103-109 is one process)."""
    import socket
    import subprocess
    sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if ndev < n and "WUN_DIST_BACKEND" not in env:
        # fewer devices than ranks (a 1-GPU test box, or no GPU at all for --dry-run): RCCL refuses duplicate
        # devices, so the exchange runs over gloo; everything else is the production path
        env["WUN_DIST_BACKEND"] = "gloo"
        log("only %d GPU(s) for %d ranks: ranks share devices, gradient exchange over gloo" % (ndev, n))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def dry_run(args):
    """Launcher / rendezvous check: process group up, rank 0's tuning table reaches every rank, one JSON line.
    Without a GPU only the host side runs (gloo); with one the Trainer is built and the table imported."""
    from wave_u_net_amd.parallel import init_distributed
    rank, local, world = init_distributed()
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    path = PINNED_TUNE_TABLES.get((args.config, args.batch, args.dtype))
    table = open(path).read() if (rank == 0 and path and os.path.exists(path)) else None
    imported = None
    if torch.cuda.is_available():
        import wave_u_net_amd as wun
        from wave_u_net_amd.training import Trainer, synthetic_source
        cfg = wun.get_config(args.config, **parse_overrides(args.set))
        tr = Trainer(cfg, batch_size=args.batch, bucket_mib=args.bucket_mib)
        mix, targets = synthetic_source(cfg, tr.batch, tr.t_in, tr.t_out, tr.device, seed=1337 + tr.rank)()
        tr.tune(mix, targets, pinned_table=table)
        imported = tr.tune_source
        table = tr.tune_table
    elif world > 1:
        box = [table]
        dist.broadcast_object_list(box, src=0)
        table = box[0]
    import hashlib
    sha = hashlib.sha256((table or "").encode()).hexdigest()[:16]
    shas = [None] * world
    if world > 1:
        dist.all_gather_object(shas, sha)
    else:
        shas = [sha]
    if len(set(shas)) != 1:
        raise SystemExit("ranks disagree on the tuning table: %s" % shas)
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "scaling": "weak", "backend": dist.get_backend() if world > 1 else None,
                          "gpu": bool(torch.cuda.is_available()), "tune_table_sha16": sha, "tilings": imported,
                          "config": {"named_config": args.config, "parallelism": "dp%d" % world}}), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def parse_overrides(items):
    out = {}
    for it in items or []:
        k, v = it.split("=", 1)
        try:
            out[k] = json.loads(v)
        except ValueError:
            out[k] = v
    return out


def family_of(kernel_name):
    return kernel_name.split("<")[0]


def pmc_traffic(family, table_text):
    """HBM bytes per launch of the kernel family (FETCH_SIZE doubled per the gfx950 note + WRITE_SIZE,
    separate rocprofv3 --pmc passes, tools/profile_round.sh) from the committed PMC summary -- but ONLY
    if that summary was collected with the very tuning table this run executes (its sha is stored
    beside the numbers); PMC counters cannot be collected from inside this process.  Else None."""
    import hashlib
    path = os.path.join(ROOT, "profiles", "round6_pmc_traffic.json")
    try:
        doc = json.load(open(path))
    except Exception:
        return None
    if not table_text or doc.get("tune_table_sha256") != hashlib.sha256(table_text.encode()).hexdigest():
        return None
    tot_b = tot_n = 0.0
    for name, ent in doc["kernels"].items():
        if family_of(name) == family:
            tot_b += ent["hbm_bytes_per_launch"] * ent["launches_profiled"]
            tot_n += ent["launches_profiled"]
    return tot_b / tot_n if tot_n else None


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline(cfg_over, params, mix, targets, names, gpu, budget_s=45.0):
    """The oracle (torch-CPU fp32 restatement of the reference graph; TensorFlow 1.8 cannot be
    installed) on THE batch the GPU timed: forward + backward of every excerpt (one excerpt at a time,
    all usable host cores) + one TF-Adam update = one step of /root/reference/Training.py:103-109 at
    batch_size 16; plus a 1-thread figure from one excerpt.  Its loss and gradients are also the
    checker of the `parity` object.  Bounded: stops adding excerpts after budget_s seconds and scales."""
    from oracle import shapes, waveunet_torch as wt       # checker / CPU baseline only
    ocfg = shapes.finalize_config(dict(shapes.BASE_MODEL_CONFIG, **cfg_over))
    cores = usable_cores()
    log("cpu baseline: os.cpu_count=%s usable=%d model=%s" % (os.cpu_count(), cores, cpu_model()))
    torch.set_num_threads(cores)
    B = mix.shape[0]
    named = [(n, p) for n, p in params]
    # warm-up (allocator, thread pool) on one excerpt, untimed
    wt.chunked_train_step(ocfg, named, mix[:1], {k: v[:1] for k, v in targets.items()}, chunk=1)
    times, loss_sum, acc, done = [], 0.0, None, 0
    t_start = time.time()
    for b in range(B):
        if b >= 4 and (time.time() - t_start) > budget_s:
            break
        tl = []
        l, g, _ = wt.chunked_train_step(ocfg, named, mix[b:b + 1], {k: v[b:b + 1] for k, v in targets.items()},
                                        chunk=1, timings=tl)
        times.append(tl[0]); loss_sum += l
        acc = g if acc is None else [a + x for a, x in zip(acc, g)]
        done += 1
    per_ex = float(np.median(times))
    # Adam on the arena (7 words / parameter)
    pp = [torch.tensor(p) for _, p in named]
    gg = [(a / done).float() for a in acc]
    mm = [torch.zeros_like(p) for p in pp]
    vv = [torch.zeros_like(p) for p in pp]
    t1 = time.time()
    wt.tf_adam_step(pp, gg, mm, vv, 1, 1e-4)
    t_adam = time.time() - t1
    step_s = per_ex * B + t_adam
    # 1-thread figure: one excerpt
    torch.set_num_threads(1)
    tl = []
    wt.chunked_train_step(ocfg, named, mix[:1], {k: v[:1] for k, v in targets.items()}, chunk=1, timings=tl)
    torch.set_num_threads(cores)
    step_1t = tl[0] * B + t_adam                        # (Adam as measured above: bandwidth-bound)
    t_out = targets[names[0]].shape[1]
    log("cpu baseline: %d/%d excerpts timed, %.3f s/excerpt (%d threads), %.2f s/excerpt (1 thread), Adam %.3f s" % (
        done, B, per_ex, cores, tl[0], t_adam))
    base = {"value": B * t_out / step_s, "unit": "output samples/s", "cores": cores, "kind": "port",
            "cpu_model": cpu_model(), "threads": cores, "s_per_step": step_s,
            "value_1_thread": B * t_out / step_1t, "s_per_step_1_thread": step_1t,
            "sample": "fwd+bwd of %d of the batch's %d excerpts (%d->%d samples each, one at a time, median %.3f s "
                      "x %d) + one TF-Adam update (%.3f s), torch-CPU fp32 oracle, %d threads; 1-thread figure from one "
                      "excerpt (%.2f s) x %d" % (done, B, mix.shape[1], t_out, per_ex, B, t_adam, cores, tl[0], B)}
    parity = None
    if gpu is not None and done == B:
        ograds = [a / B for a in acc]
        worst, worst_name = 0.0, ""
        for (n, _), og, (off, size) in zip(named, ograds, gpu["slices"]):
            gg_ = gpu["grads"][off:off + size].double().reshape(og.shape)
            rel = (gg_ - og).abs().max().item() / max(og.abs().max().item(), 1e-30)
            if rel > worst:
                worst, worst_name = rel, n
        parity = {"checker": "torch-CPU fp32 oracle (oracle/waveunet_torch.py), the step-0 batch of the timed run",
                  "loss_gpu": gpu["loss"], "loss_oracle": loss_sum / B,
                  "loss_rel_err": abs(gpu["loss"] - loss_sum / B) / max(abs(loss_sum / B), 1e-30),
                  "max_rel_grad_err": worst, "worst_tensor": worst_name, "tensors_checked": len(named),
                  "rel_err_definition": "max|g_gpu - g_oracle| / max|g_oracle| per gradient tensor, worst tensor"}
    return base, parity


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--settle-s", type=float, default=0.5,
                    help="keep taking untimed steps after the warm-up until this many seconds have passed since it began (0 = off)")
    ap.add_argument("--config", default="m1_context")
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--no-cpu-baseline", action="store_true", help="skip the CPU baseline (and the parity object)")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--force-allreduce", action="store_true",
                    help="world 1 only: run the bucketed, overlapped RCCL gradient all-reduce of the N>1 path anyway "
                         "(1-rank group: the collective kernels run and contend for CUs, the data is unchanged)")
    ap.add_argument("--bucket-mib", type=float, default=16.0)
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"],
                    help="f32 = the reference's arithmetic (headline); bf16 = the mode of BASELINE.json configs[2],[4]: "
                         "activations and their gradients live in HBM as bf16, bf16 MFMA, fp32 accumulate / weights / weight gradients / Adam")
    ap.add_argument("--set", action="append", default=[], metavar="KEY=VALUE",
                    help="model_config override (JSON value), e.g. --set num_layers=4 --set num_initial_filters=8")
    ap.add_argument("--dry-run", action="store_true",
                    help="stop after process-group initialisation + tuning-table broadcast (launcher check; runs without a GPU)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(relaunch_under_torchrun(args.gpus))
    if int(os.environ.get("WORLD_SIZE", "1")) != args.gpus:       # before any rendezvous: a mismatch must not hang
        raise SystemExit("--gpus %d but WORLD_SIZE=%s" % (args.gpus, os.environ.get("WORLD_SIZE")))
    if args.dry_run:
        return dry_run(args)

    import wave_u_net_amd as wun
    from wave_u_net_amd import _lib
    from wave_u_net_amd.training import Trainer, synthetic_source

    # the tilings the committed profiles describe: handed to the Trainer as TEXT (read-only; ADVICE round 2 -- the
    # tracked file must never double as the writable WUN_TUNE_CACHE)
    pinned_path = PINNED_TUNE_TABLES.get((args.config, args.batch, args.dtype)) if not args.set else None
    pinned_text = open(pinned_path).read() if (pinned_path and os.path.exists(pinned_path)
                                               and "WUN_TUNE_CACHE" not in os.environ) else None

    if args.force_allreduce and int(os.environ.get("WORLD_SIZE", "1")) == 1 and not dist.is_initialized():
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        import socket
        sk = socket.socket(); sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]; sk.close()
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)

    cfg = wun.get_config(args.config, **parse_overrides(args.set))
    if args.dtype == "bf16":
        cfg["compute_dtype"] = "bf16"
    log("building trainer")
    tr = Trainer(cfg, batch_size=args.batch, bucket_mib=args.bucket_mib)
    world = tr.world
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    forced = None
    if args.force_allreduce and world == 1:
        from wave_u_net_amd.parallel import OverlappedGradAllReducer
        plan = tr.sep._active
        forced = OverlappedGradAllReducer(plan.tensors, plan.info.arena_floats, args.bucket_mib, device=tr.device)
        log("forced all-reduce: %d buckets of >= %.0f MiB on a 1-rank RCCL group" % (len(forced.buckets), args.bucket_mib))
    source = synthetic_source(cfg, tr.batch, tr.t_in, tr.t_out, tr.device, seed=1337 + tr.rank)
    mix, targets = source()
    log("data ready: mix %s targets %s" % (tuple(mix.shape), tuple(targets.shape)))

    def barrier():
        if world > 1:
            dist.barrier()

    def step():
        if forced is None:
            return tr.step(mix, targets)
        tr.sep.get_output(mix, True)
        loss_ = tr.sep.loss_and_gradients(targets, *forced.begin())
        forced.launch(tr.sep.grads, force=True)
        forced.finish()
        tr.sep.adam_step(tr.lr, grad_scale=1.0)
        return loss_

    t_tune = time.time()
    tr.tune(mix, targets, pinned_table=pinned_text)   # pinned table, or one-off autotuning on rank 0 + broadcast (untimed)
    table_text = getattr(tr, "tune_table", None)
    pinned = tr.tune_source == "pinned"          # decided by whether the import of the committed text SUCCEEDED
    tilings = ("pinned:" + os.path.relpath(pinned_path, ROOT)) if pinned else tr.tune_source
    log("tilings: %s (%.2f s)" % (tilings, time.time() - t_tune))

    # ---- step 0 of the timed inputs, kept for the parity object (no optimizer step: weights untouched) ----
    gpu0 = None
    if tr.rank == 0 and world == 1 and not args.no_cpu_baseline and args.config in wun.NAMED_CONFIGS:
        tr.sep.get_output(mix, True)
        l0 = tr.sep.loss_and_gradients(targets)
        torch.cuda.synchronize()
        gpu0 = {"loss": float(l0.item()), "grads": tr.sep.grads.detach().cpu(),
                "slices": [(off, int(np.prod(shp))) for _, off, shp in tr.sep._active.tensors],
                "params": [(n, v.detach().cpu().numpy().copy()) for n, v in tr.sep.variables().items()]}

    t_w = time.perf_counter()
    for _ in range(args.warmup):
        loss = step()
    torch.cuda.synchronize()
    # settle (untimed, reported as config.settle_steps): a fresh box is still paging the library in and ramping its
    # clocks during the first few hundred ms; W steps of this model are < 0.1 s, so keep stepping until 0.5 s have passed
    # (the count is rank 0's decision: every rank must take the same number of steps, each has a collective)
    settle_steps = 0
    if args.warmup > 0 and args.settle_s > 0:
        t_warm = time.perf_counter() - t_w
        settle_steps = int(min(200, max(0, np.ceil((args.settle_s - t_warm) / (t_warm / args.warmup)))))
        if world > 1:
            n = torch.tensor([settle_steps], dtype=torch.int64, device=tr.device)
            dist.broadcast(n, src=0)
            settle_steps = int(n.item())
        for _ in range(settle_steps):
            loss = step()
        torch.cuda.synchronize()
    log("warm-up done (%d steps + %d settle steps)" % (args.warmup, settle_steps))
    barrier()
    torch.cuda.synchronize()
    # one HIP event per step on the stream the step is launched on (torch's current stream = the stream handed to
    # the C ABI): per-step durations for the median / p10 / p90 of SURVEY section 8(d); `value` stays the wall clock
    # of the whole region
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    gc_was = gc.isenabled()
    gc.disable()                                  # no collector pause between two launches of the timed region
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(args.steps):
        loss = step()
        evs[i + 1].record()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if gc_was:
        gc.enable()
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=tr.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    loss_val = float(loss.item())
    log("timed region: %d steps in %.3f s (%.2f ms/step)" % (args.steps, elapsed, 1e3 * elapsed / args.steps))

    step_ms = np.array([evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps)])
    info = tr.sep.plan_info()
    ms_per_step = 1e3 * elapsed / args.steps
    out_samples = world * tr.batch * tr.t_out
    step_flops = info.fwd_flops + info.bwd_flops
    result = {
        "metric": "waveform samples/sec fwd+bwd, M1 12-level Wave-U-Net @1/2/4/8 GPU",
        "value": out_samples * args.steps / elapsed,
        "unit": "output samples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "ms_median": float(np.median(step_ms)), "ms_p10": float(np.percentile(step_ms, 10)),
        "ms_p90": float(np.percentile(step_ms, 90)), "ms_max": float(step_ms.max()),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        # (the arithmetic that actually ran: a config that asks for bf16 without qualifying gets the exact-fp32 plan --
        #  wun_plan_info.compute_dtype_effective)
        "dtype": "f32" if int(info.compute_dtype_effective) == 0 else "bf16 (activations and their gradients in HBM + all MFMA operands; fp32 accumulate, weights, weight gradients, Adam)",
        "data": "synthetic",
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
                   "tilings": tilings, "settle_steps": settle_steps,
                   "forced_allreduce": bool(forced), "bucket_mib": args.bucket_mib,
                   "step_tflops_executed": step_flops / 1e12,
                   # every observed conv output computed once (== executed since round 6 for the exact-fp32 mode: the
                   # skip windows' even positions are no longer computed twice; the bf16 mode's context plans still do)
                   "step_tflops_nonredundant": (info.fwd_flops_unique + info.bwd_flops_unique) / 1e12,
                   "requested_dtype": args.dtype,
                   "step_tflops_reference_graph": 3.0 * info.fwd_flops_dense / 1e12,
                   "achieved_tflops_executed": step_flops / (ms_per_step * 1e9),
                   "step_frac_of_fp32_mfma_peak": step_flops / (ms_per_step * 1e9) / PEAK_FP32_MFMA_TFLOPS},
    }

    # ---- communication diagnostics (N > 1, or --force-allreduce): why the scaling efficiency is what it is ----
    # Untimed extra steps after the timed region, every rank in lock-step:
    #   exposed_ms          time the launch stream spends waiting for the gradient all-reduce AFTER its own last backward
    #                       kernel has finished (HIP events: end of backward -> all buckets reduced), i.e. the communication
    #                       the overlap did not hide; median over the diagnostic steps, max over ranks
    #   ms_per_step_no_comm the same step with the all-reduce skipped (same process, same streams and priorities)
    # ms_per_step - ms_per_step_no_comm = what communication costs in total (exposed wait + contention for CUs / queues).
    reducer = forced if forced is not None else (tr.reducer if world > 1 else None)
    if reducer is not None:
        ndiag = max(3, min(10, args.steps))
        overlapped = hasattr(reducer, "begin")
        exposed = []
        # The diagnostic steps are NOT training: the no-comm ones apply un-reduced (per-rank) gradients, which would leave the
        # replicas' parameters divergent for everything that follows.  Parameters, both Adam slots and the
        # step counter are snapshotted here and restored after the block.
        snap = (tr.sep.params.clone(), tr.sep.adam_m.clone(), tr.sep.adam_v.clone(), tr.sep.global_step)
        for _ in range(ndiag):
            tr.sep.get_output(mix, True)
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            if overlapped:
                tr.sep.loss_and_gradients(targets, *reducer.begin())
                e0.record()
                reducer.launch(tr.sep.grads, force=forced is not None)
                reducer.finish()
            else:
                tr.sep.loss_and_gradients(targets)
                e0.record()
                reducer.all_reduce(tr.sep.grads)
            e1.record()
            tr.sep.adam_step(tr.lr, grad_scale=1.0 if forced is not None else reducer.grad_scale)
            torch.cuda.synchronize()
            exposed.append(e0.elapsed_time(e1))
        barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(ndiag):
            tr.sep.get_output(mix, True)
            if overlapped:
                tr.sep.loss_and_gradients(targets, *reducer.begin())      # bucket events still recorded: same launch sequence
            else:
                tr.sep.loss_and_gradients(targets)
            tr.sep.adam_step(tr.lr, grad_scale=1.0 if forced is not None else reducer.grad_scale)
        torch.cuda.synchronize()
        no_comm_ms = 1e3 * (time.perf_counter() - t1) / ndiag
        stats = torch.tensor([float(np.median(exposed)), no_comm_ms], dtype=torch.float64, device=tr.device)
        if world > 1:
            dist.all_reduce(stats, op=dist.ReduceOp.MAX)
        result["comm"] = {"buckets": len(reducer.buckets), "bytes": 4 * int(sum(e - s for s, e in reducer.buckets)),
                          "bucket_bytes": [4 * int(e - s) for s, e in reducer.buckets],
                          "overlapped": bool(overlapped), "exposed_ms": float(stats[0].item()), "diagnostic_steps": ndiag,
                          "backend": dist.get_backend() if dist.is_initialized() else None}
        result["ms_per_step_no_comm"] = float(stats[1].item())
        result["comm"]["note"] = "diagnostic steps are not training: parameters / Adam state restored afterwards"
        tr.sep.params.copy_(snap[0]); tr.sep.adam_m.copy_(snap[1]); tr.sep.adam_v.copy_(snap[2]); tr.sep.global_step = snap[3]
        del snap
        torch.cuda.synchronize()
        log("comm: %d buckets, %.1f MB, exposed %.3f ms/step; step without the all-reduce %.3f ms" % (
            len(reducer.buckets), result["comm"]["bytes"] / 1e6, result["comm"]["exposed_ms"], no_comm_ms))

    if not args.no_roofline:
        # a few extra steps with HIP events around every heavy launch (all ranks step together,
        # rank 0 records): kept outside the timed region so the events do not perturb `value`
        lib = _lib.load()
        nprof = 2
        if tr.rank == 0:
            lib.wun_profile_begin()
        for _ in range(nprof):
            step()
        torch.cuda.synchronize()
        if tr.rank == 0:
            buf = ctypes.create_string_buffer(1 << 20)
            _lib.check(lib.wun_profile_end(buf, len(buf)))
            prof = json.loads(buf.value.decode())
            if prof.get("launches") and os.environ.get("WUN_PROFILE_DETAIL"):
                with open(os.environ["WUN_PROFILE_DETAIL"], "w") as f:
                    json.dump(prof["launches"], f)
            kernels = prof["kernels"]
            # an event pair costs ~5 us of packet processing that is not kernel time (the library brackets nothing after
            # every launch and reports the median): durations below are NET of it, so that they can agree with rocprofv3's
            # kernel durations; roofline.achieved_raw_events keeps the uncorrected figure
            ovh = float(prof.get("bracket_overhead_ms", 0.0))
            for k in kernels:
                k["ms_raw"] = k["ms"]
                k["ms"] = max(k["ms"] - ovh * k["launches"], 0.05 * k["ms"])
            log("event bracket overhead: %.2f us per launch (median empty bracket)" % (1e3 * ovh))
            kernels.sort(key=lambda k: -k["ms"])
            fam = {}
            for k in kernels:
                f = fam.setdefault(family_of(k["name"]), {"ms": 0.0, "ms_raw": 0.0, "flops": 0.0, "launches": 0})
                f["ms"] += k["ms"]; f["ms_raw"] += k["ms_raw"]; f["flops"] += k["flops"]; f["launches"] += k["launches"]
                log("  %-46s launches/step %5.1f  ms/step %8.3f  TFLOP/s %7.2f" % (
                    k["name"], k["launches"] / nprof, k["ms"] / nprof, k["flops"] / max(k["ms"], 1e-9) / 1e9))
            fams = sorted(fam.items(), key=lambda kv: -kv[1]["ms"])
            for name, f in fams:
                log("  family %-38s launches/step %5.1f  ms/step %8.3f  TFLOP/s %7.2f" % (
                    name, f["launches"] / nprof, f["ms"] / nprof, f["flops"] / max(f["ms"], 1e-9) / 1e9))
            top_name, top = fams[0]
            avg_ms = top["ms"] / top["launches"]
            achieved = top["flops"] / (top["ms"] * 1e-3) / 1e12
            result["roofline"] = {
                "bound": "mfma", "kernel": top_name + " (all instantiations)", "achieved": achieved,
                "peak": family_peak(top_name), "unit": "TFLOP/s", "frac": achieved / family_peak(top_name),
                "traffic": pmc_traffic(top_name, table_text),
                "event_bracket_overhead_us": 1e3 * ovh,
                "achieved_raw_events": top["flops"] / (top["ms_raw"] * 1e-3) / 1e12,
                "avg_launch_ms": avg_ms, "launches_per_step": top["launches"] / nprof,
                "flops_per_launch": top["flops"] / top["launches"],
                "family_ms_per_step": {n: f["ms"] / nprof for n, f in fams},
                "family_frac": {n: f["flops"] / (f["ms"] * 1e-3) / 1e12 / family_peak(n) for n, f in fams if f["flops"] > 0},
                "family_launches_per_step": {n: f["launches"] / nprof for n, f in fams},
                "kernel_ms_per_step": {k["name"]: k["ms"] / nprof for k in kernels},
                "kernel_tflops": {k["name"]: k["flops"] / (k["ms"] * 1e-3) / 1e12 for k in kernels if k["ms"] > 0},
                # the bandwidth-bound kernels against the HBM roof (north_star: "achieved fraction of HBM roofline"):
                # ALGORITHMIC bytes (what a perfect implementation must move: DESIGN.md section 5) / launch time by the same
                # HIP-event brackets; frac of the 8 TB/s the microarchitecture guide quotes (6.3 TB/s is what a copy reaches)
                "hbm": {k["name"]: {"GBps": k["bytes"] / (k["ms"] * 1e-3) / 1e9, "frac": k["bytes"] / (k["ms"] * 1e-3) / 1e9 / PEAK_HBM_GBPS,
                                    "bytes_per_launch": k["bytes"] / k["launches"], "us_per_launch": 1e3 * k["ms"] / k["launches"],
                                    "launches_per_step": k["launches"] / nprof}
                        for k in kernels if k.get("bytes", 0) > 0 and k["ms"] > 0},
                "hbm_peak_GBps": PEAK_HBM_GBPS,
            }

    if gpu0 is not None:
        names = list(cfg["source_names"])
        htg = {n: targets[i].cpu().numpy() for i, n in enumerate(names)}
        base, parity = cpu_baseline(wun.NAMED_CONFIGS[args.config], gpu0["params"], mix.cpu().numpy(), htg, names, gpu0)
        result["cpu_baseline"] = base
        if parity is not None:
            result["parity"] = parity
        result["config"]["speedup_vs_cpu_baseline"] = result["value"] / base["value"]

    if tr.rank == 0:
        print(json.dumps(result), flush=True)
    if dist.is_initialized():
        if world > 1:
            dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
