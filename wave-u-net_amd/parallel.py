"""Data-parallel gradient exchange: one process per GPU, flat fp32 gradient arena summed
with torch.distributed all-reduce (backend "nccl" = RCCL over xGMI on ROCm; "gloo" on CPU
for tests).  The reference has no multi-device code at all; the
semantics are: equal per-rank batches, loss is a mean, so the global gradient is the mean
of the per-rank gradients -- the 1/N is folded into the Adam kernel's grad_scale.

The arena is in TF creation order (down0..down11, bottleneck, up0..up11, head) while the
backward pass finishes tensors in the reverse order, so buckets are contiguous suffixes:
the big late-arena tensors (up path, bottleneck, deep down levels: ~80 % of the bytes) are
all-reduced first.  xGMI is point-to-point (7 links x ~153 GB/s per GPU): a few large
buckets (default 16 MiB) keep the ring per-link bound instead of latency bound.
"""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Initialise from torchrun's environment (RANK/LOCAL_RANK/WORLD_SIZE/MASTER_*).
    Returns (rank, local_rank, world_size)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        from . import _lib
        if _lib.LOW_PRIORITY_PLANS["created"] and os.environ.get("WUN_ALLOW_LOW_PRIO_WITH_COLLECTIVES") is None:
            # include/wun.h, wun_config.exclusive_streams: the lowest-priority hardware queues of an earlier
            # single-GPU plan stay with the process and cost the data-parallel step ~40 %
            raise RuntimeError(
                "a plan with exclusive_streams=1 (lowest-priority side streams) was created in this process before the "
                "process group: initialise torch.distributed first, or build that separator with "
                "model_config['exclusive_streams'] = False (override: WUN_ALLOW_LOW_PRIO_WITH_COLLECTIVES=1)")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend is None:
            # WUN_DIST_BACKEND=gloo runs the multi-rank flow without RCCL (e.g. two ranks sharing the
            # one GPU of a test box: RCCL refuses duplicate devices, gloo stages through the host)
            backend = os.environ.get("WUN_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local)
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, local, world


def bucket_bounds(tensor_table, arena_floats, bucket_floats):
    """Split [0, arena) at tensor boundaries into buckets of >= bucket_floats, walking from
    the END of the arena (= backward completion order).  Returns [(start, end)] in the order
    they should be reduced."""
    starts = sorted({int(off) for _, off, _ in tensor_table} | {0})
    out = []
    end = int(arena_floats)
    cur_end = end
    for st in reversed(starts):
        if cur_end - st >= bucket_floats or st == 0:
            if cur_end > st:
                out.append((st, cur_end))
            cur_end = st
    return out


class GradAllReducer(object):
    """Sum-all-reduce of a flat gradient tensor in buckets; works on CUDA (RCCL) and CPU (gloo)."""

    def __init__(self, tensor_table, arena_floats, bucket_mib=16.0, group=None):
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.group = group
        self.buckets = bucket_bounds(tensor_table, arena_floats, int(bucket_mib * (1 << 20) / 4))
        assert sum(e - s for s, e in self.buckets) == int(arena_floats)

    @property
    def grad_scale(self):
        return 1.0 / self.world

    def all_reduce(self, flat_grads):
        """In place; asynchronous launches, completed with respect to the current stream when
        this returns (NCCL work.wait() only inserts a stream dependency)."""
        if self.world == 1:
            return
        works = [dist.all_reduce(flat_grads[s:e], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                 for s, e in self.buckets]
        for w in works:
            w.wait()


class OverlappedGradAllReducer(GradAllReducer):
    """GPU variant: bucket k's all-reduce is issued on a communication stream that waits only for
    the event the backward pass records when that bucket's gradients are final, so the exchange
    over xGMI overlaps the remaining backward kernels.  Usage per step:
        loss = sep.loss_and_gradients(targets, *reducer.begin())
        reducer.launch(sep.grads); ...; reducer.finish()   # before the Adam kernel
    """

    def __init__(self, tensor_table, arena_floats, bucket_mib=16.0, group=None, device=None):
        super().__init__(tensor_table, arena_floats, bucket_mib, group)
        self.device = device
        # The communication stream runs on the highest-priority hardware queue (WUN_COMM_PRIO=normal: default priority):
        # a bucket's all-reduce then starts as soon as its event fires instead of queueing behind the backward kernels.
        # Measured with bench.py --force-allreduce (profiles/round5_force_allreduce.txt, same box, 3 buckets): step 8.60 ->
        # 8.58 ms, exposed wait 0.073 -> 0.052 ms.  The plan's side streams must stay at NORMAL priority beside it:
        # low-priority side streams cost 10.5 ms with a high-priority and 11.3 - 12.1 ms with a normal-priority collective.
        prio = 0 if os.environ.get("WUN_COMM_PRIO", "") == "normal" else -1
        self.comm_stream = torch.cuda.Stream(device=device, priority=prio)
        self.events = []
        for _ in self.buckets:
            ev = torch.cuda.Event(enable_timing=False)
            ev.record(torch.cuda.current_stream(device))      # forces creation of the HIP event handle
            self.events.append(ev)
        self._works = []

    def begin(self):
        """(bucket_starts, bucket_events) to pass to UnetAudioSeparator.loss_and_gradients."""
        return [s for s, _ in self.buckets], self.events

    def launch(self, flat_grads, force=False):
        """Queue one all-reduce per bucket; each waits (on the device) for its bucket event."""
        self._works = []
        if self.world == 1 and not force:
            return
        for (s, e), ev in zip(self.buckets, self.events):
            self.comm_stream.wait_event(ev)
            with torch.cuda.stream(self.comm_stream):
                self._works.append(dist.all_reduce(flat_grads[s:e], op=dist.ReduceOp.SUM,
                                                   group=self.group, async_op=True))

    def finish(self):
        """Make the current stream wait for all bucket all-reduces."""
        for w in self._works:
            w.wait()
        if self._works:
            torch.cuda.current_stream(self.device).wait_stream(self.comm_stream)
        self._works = []


def broadcast_parameters(flat_params, src=0, group=None):
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.broadcast(flat_params, src=src, group=group)
