"""Training loop with the shape of the reference's Training.train
(/root/reference/Training.py:24-121): build the separator once, run `epoch_it` steps of
{forward, MSE loss, backward, Adam}, count global_step, return a checkpoint path.

The reference feeds the step from a tf.data pipeline over MUSDB (Datasets.py); here
`batch_source` is any callable returning (mix [B,Tin,C], targets [S,B,Tout,C]) GPU tensors
honouring that pipeline's output contract (float32, mix = sum of sources, targets
centre-cropped): datasets.DeviceSnippetSource for real tracks, `synthetic_source` for the
benchmark.
"""
import json
import os
import time

import numpy as np
import torch
import torch.distributed as dist

from .parallel import GradAllReducer, OverlappedGradAllReducer, broadcast_parameters, init_distributed
from .checkpoint import load_checkpoint, save_checkpoint
from .separator import UnetAudioSeparator


def synthetic_source(model_config, batch, t_in, t_out, device, seed=1337):
    """Band-limited noise sources (9-tap moving average of U(-1,1)), mix = sum of sources,
    targets = centre crop (Utils.crop_sample, Utils.py:38-42).  Generated once on the GPU and
    reused, so the timed loop has its inputs resident in HBM."""
    S, C = len(model_config["source_names"]), 1 if model_config["mono_downmix"] else 2
    gen = torch.Generator(device="cpu").manual_seed(seed)
    srcs = []
    for _ in range(S):
        w = torch.rand((batch, C, t_in + 8), generator=gen) * 2 - 1
        sm = torch.nn.functional.avg_pool1d(w, 9, stride=1)
        sm = sm * ((0.9 / S) / sm.abs().max().clamp_min(1e-9))
        srcs.append(sm.permute(0, 2, 1).contiguous())
    src = torch.stack(srcs)                           # [S, B, Tin, C]
    mix = src.sum(0).to(device)
    pad = (t_in - t_out) // 2
    targets = src[:, :, pad:t_in - pad, :].contiguous().to(device) if pad > 0 else src.to(device)

    def source():
        return mix, targets
    return source


class Trainer(object):
    """One process per GPU.  step() = sess.run([separator_solver, ...]) of Training.py:105."""

    def __init__(self, model_config, batch_size=None, device=None, seed=1337, bucket_mib=16.0):
        self.rank, self.local_rank, self.world = init_distributed()
        # Scheduling hint of the plan (include/wun.h): low-priority side streams only when no collective shares the
        # device -- with a process group initialised (multi-GPU, or bench.py --force-allreduce) they must stay normal.
        if "exclusive_streams" not in model_config:
            model_config = dict(model_config)
            model_config["exclusive_streams"] = not (dist.is_available() and dist.is_initialized())
        self.cfg = model_config
        if device is None:
            device = "cuda:%d" % (self.local_rank % max(1, torch.cuda.device_count()))
        self.device = torch.device(device)
        torch.cuda.set_device(self.device)
        self.sep = UnetAudioSeparator(model_config, device=self.device, seed=seed)
        self.batch = batch_size or model_config["batch_size"]
        in_shape, out_shape = self.sep.get_padding(np.array([self.batch, model_config["num_frames"], 0]))
        self.t_in, self.t_out = int(in_shape[1]), int(out_shape[1])
        plan = self.sep._plan(self.batch, self.t_in)
        self.sep._active = plan
        self.sep._ensure_variables(plan)
        broadcast_parameters(self.sep.params)
        self.overlap = self.world > 1 and os.environ.get("WUN_NO_OVERLAP") is None
        if self.overlap:
            self.reducer = OverlappedGradAllReducer(plan.tensors, plan.info.arena_floats, bucket_mib,
                                                    device=self.device)
        else:
            self.reducer = GradAllReducer(plan.tensors, plan.info.arena_floats, bucket_mib)
        self.lr = model_config["init_sup_sep_lr"]

    def tune(self, mix, targets, pinned_table=None):
        """One-off kernel autotuning on a real batch (skipped with WUN_NO_TUNE=1).  Rank 0 decides
        the tilings -- from `pinned_table` (the TEXT of a committed table: read-only, never written
        back), else from WUN_TUNE_CACHE=<file> (a writable cache) if either holds a table for this
        plan and library build, else by measuring (wun_plan_tune) -- and broadcasts the exported
        table; every rank imports that same table, so all replicas run bit-identical kernels
        (per-rank tuning would let replicas differ by fp32 rounding before the gradient all-reduce)
        and nobody reads a cache file that another rank is still writing.  The cache is written
        atomically.  `tune_source` records which of "pinned" / "cache" / "autotuned" / "heuristic"
        happened (decided by whether the import SUCCEEDED, not by comparing files afterwards)."""
        self.tune_source, self.tune_table = "heuristic", None
        if os.environ.get("WUN_NO_TUNE") is not None:
            return
        cache = os.environ.get("WUN_TUNE_CACHE")
        self.sep.get_output(mix, True)                    # makes this (batch, length) plan the active one
        table, source = None, None
        if self.rank == 0:
            if pinned_table:
                try:
                    self.sep.tune_import(pinned_table)
                    table, source = pinned_table, "pinned"
                except ValueError:
                    table = None                         # other plan / library build: fall through
            if table is None and cache and os.path.exists(cache):
                try:
                    text = open(cache).read()
                    self.sep.tune_import(text)
                    table, source = text, "cache"
                except ValueError:
                    table = None                         # other plan / library build / truncated: tune afresh
            if table is None:
                self.sep.tune(mix, targets)
                table, source = self.sep.tune_export(), "autotuned"
                if cache:
                    tmp = "%s.tmp.%d" % (cache, os.getpid())
                    with open(tmp, "w") as f:
                        f.write(table)
                    os.replace(tmp, cache)
        box = [table, source]
        if self.world > 1:
            dist.broadcast_object_list(box, src=0)
            if self.rank != 0:
                self.sep.tune_import(box[0])
        self.tune_table, self.tune_source = box[0], box[1]

    def step(self, mix, targets):
        self.sep.get_output(mix, True)
        if self.overlap:
            # bucket events are recorded by the backward pass; the all-reduces wait on them
            loss = self.sep.loss_and_gradients(targets, *self.reducer.begin())
            self.reducer.launch(self.sep.grads)
            self.reducer.finish()
        else:
            loss = self.sep.loss_and_gradients(targets)
            self.reducer.all_reduce(self.sep.grads)
        self.sep.adam_step(self.lr, grad_scale=self.reducer.grad_scale)
        return loss


def train(model_config, experiment_id, load_model=None, batch_source=None, log_every=None):
    """Training.train(model_config, experiment_id, load_model=None) -> save_path
    (Training.py:24-25,121)."""
    if model_config["network"] != "unet":
        raise NotImplementedError(model_config["network"])         # Training.py:28-33
    if log_every is None:
        log_every = 100 if model_config["epoch_it"] > 100 else 1
    tr = Trainer(model_config)
    if load_model is not None:
        # rank 0 reads the checkpoint (the only rank that is guaranteed to have the path: train()
        # hands save_path to every rank, but the file system need not be shared); parameters, both
        # Adam slots and global_step are then broadcast so every replica resumes the SAME state
        # (`.npz` of this package, or a TensorFlow V2 checkpoint prefix as the reference's Saver writes: checkpoint.py)
        if tr.rank == 0:
            load_checkpoint(tr.sep, load_model)
        if tr.world > 1:
            for t in (tr.sep.params, tr.sep.adam_m, tr.sep.adam_v):
                broadcast_parameters(t)
            box = [tr.sep.global_step]
            dist.broadcast_object_list(box, src=0)
            tr.sep.global_step = int(box[0])
    if batch_source is None:
        batch_source = synthetic_source(model_config, tr.batch, tr.t_in, tr.t_out, tr.device,
                                        seed=1337 + tr.rank)
    elif getattr(batch_source, "needs_trainer", False):
        batch_source = batch_source(tr)          # factory: needs the trainer's device / shapes / rank
    log_dir = os.path.join(model_config["log_dir"], str(experiment_id))
    if tr.rank == 0:
        os.makedirs(log_dir, exist_ok=True)
    log = open(os.path.join(log_dir, "train.jsonl"), "a") if tr.rank == 0 else None
    tr.tune(*batch_source())
    t0 = time.time()
    for it in range(model_config["epoch_it"]):                      # Training.py:103-109
        mix, targets = batch_source()
        loss = tr.step(mix, targets)
        if log is not None and (it % log_every == 0 or it == model_config["epoch_it"] - 1):
            log.write(json.dumps({"global_step": tr.sep.global_step, "sep_loss": float(loss.item()),
                                  "elapsed_s": time.time() - t0}) + "\n")
            log.flush()
    torch.cuda.synchronize()
    save_path = None
    if tr.rank == 0:                                                # Training.py:113
        ckpt_dir = os.path.join(model_config["model_base_dir"], str(experiment_id))
        os.makedirs(ckpt_dir, exist_ok=True)
        # Training.py:113 saves "<dir>/<id>-<step>" with the V2 Saver; model_config["checkpoint_format"] = "tf" writes
        # exactly that (restorable by the reference), the default is this package's .npz
        if model_config.get("checkpoint_format", "npz") == "tf":
            save_path = save_checkpoint(tr.sep, os.path.join(ckpt_dir, "%s-%d" % (experiment_id, tr.sep.global_step)), "tf")
        else:
            save_path = save_checkpoint(tr.sep, os.path.join(ckpt_dir, "%s-%d.npz" % (experiment_id, tr.sep.global_step)))
        log.close()
    if tr.world > 1:
        # every rank returns the checkpoint path (Training.py:121 has one process; here the callers --
        # optimise()'s next epoch, test() -- run on all ranks and must agree on it)
        box = [save_path]
        dist.broadcast_object_list(box, src=0)
        save_path = box[0]
    return save_path
