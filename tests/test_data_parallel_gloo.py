"""Data-parallel layer on CPU: world_size-2 gloo processes.  The gradient all-reduce logic
(bucketing over the TF-order arena, sum + 1/N folded into Adam) is the same code that runs over
RCCL on the GPUs; here each rank's gradients come from the oracle on its half of the batch and
the result must equal the single-process gradient on the concatenated batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import shapes, waveunet_torch as wt
from oracle.golden_params import GOLDEN_CASES, golden_params

import wave_u_net_amd as wun
from wave_u_net_amd.parallel import GradAllReducer, bucket_bounds, init_distributed
from wave_u_net_amd.separator import UnetAudioSeparator

CASE = "full_multi_small"


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _flat_grads(table, named_grads, n):
    flat = torch.zeros(n, dtype=torch.float64)
    for (name, off, shp), (gname, g) in zip(table, named_grads):
        assert name == gname
        flat[off:off + g.numel()] = g.reshape(-1)
    return flat


def _setup():
    case = GOLDEN_CASES[CASE]
    ocfg = shapes.finalize_config(dict(shapes.BASE_MODEL_CONFIG, **case["cfg"]))
    params = golden_params(ocfg, case["seed"])
    sep = UnetAudioSeparator(wun.get_config("baseline", **case["cfg"]))
    i, o = shapes.get_padding(ocfg, [4, case["frames"], 0])
    plan = sep._plan(2, i[1])
    mix, targets = wt.synthetic_batch(ocfg, 4, i[1], o[1], seed=99)
    return ocfg, params, plan, mix, targets


def _grads(ocfg, params, mix, targets):
    tp = wt.params_to_torch(params, torch.float64, requires_grad=True)
    loss, grads = wt.train_step(ocfg, tp, torch.tensor(mix, dtype=torch.float64),
                                {k: torch.tensor(v, dtype=torch.float64) for k, v in targets.items()})
    return loss, [(n, g) for (n, _), g in zip(tp, grads)]


def _worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    r, lr, w = init_distributed(backend="gloo")
    assert (r, w) == (rank, world)
    ocfg, params, plan, mix, targets = _setup()
    half = slice(2 * rank, 2 * rank + 2)                       # each rank: its own 2 excerpts
    _, named = _grads(ocfg, params, mix[half], {k: v[half] for k, v in targets.items()})
    n = int(plan.info.arena_floats)
    flat = _flat_grads(plan.tensors, named, n)
    red = GradAllReducer(plan.tensors, n, bucket_mib=0.02)     # tiny buckets -> several all-reduces
    assert len(red.buckets) > 2
    red.all_reduce(flat)
    flat *= red.grad_scale                                     # what the Adam kernel's grad_scale does
    if rank == 0:
        np.save(os.path.join(out_dir, "dp_grads.npy"), flat.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradients_equal_full_batch(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    got = np.load(os.path.join(str(tmp_path), "dp_grads.npy"))
    ocfg, params, plan, mix, targets = _setup()
    _, named = _grads(ocfg, params, mix, targets)              # single process, all 4 excerpts
    want = _flat_grads(plan.tensors, named, int(plan.info.arena_floats)).numpy()
    assert np.abs(got - want).max() <= 1e-12 * max(1.0, np.abs(want).max())


def test_bucket_bounds_cover_arena_in_backward_order():
    sep = UnetAudioSeparator(wun.get_config("baseline"))
    plan = sep._plan(16, 16384)
    n = int(plan.info.arena_floats)
    b = bucket_bounds(plan.tensors, n, int(4 * (1 << 20) / 4))
    # contiguous cover of [0, n), walked from the end (head / up path first, down0 last)
    assert b[0][1] == n and b[-1][0] == 0
    for (s0, e0), (s1, e1) in zip(b[:-1], b[1:]):
        assert s0 == e1 and e0 > s0
    starts = {off for _, off, _ in plan.tensors}
    assert all(s in starts for s, _ in b)                      # cuts only at tensor boundaries
    assert all(e - s >= 4 * (1 << 20) / 4 for s, e in b[:-1])
    assert len(b) >= 4                                         # 41 MB arena -> several 4 MiB buckets


def test_single_process_is_a_noop():
    sep = UnetAudioSeparator(wun.get_config("baseline", num_layers=3, num_initial_filters=8))
    plan = sep._plan(1, 64)
    red = GradAllReducer(plan.tensors, plan.info.arena_floats)
    g = torch.arange(int(plan.info.arena_floats), dtype=torch.float32)
    ref = g.clone()
    red.all_reduce(g)
    assert torch.equal(g, ref) and red.grad_scale == 1.0


def _early_stop_worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    init_distributed(backend="gloo")
    from wave_u_net_amd import validation
    calls = []

    def fake_test(model_config, partition, model_folder, load_model, tracks=None, **kw):
        calls.append(partition)
        return 0.25 + rank                     # what a rank evaluating its own (different) model would see

    validation.test = fake_test
    loss = validation._rank0_test({}, "valid", "x", None, [])
    with open(os.path.join(out_dir, "r%d.txt" % rank), "w") as f:
        f.write("%r %d" % (loss, len(calls)))
    dist.barrier()
    dist.destroy_process_group()


def test_validation_loss_is_rank0s_on_every_rank(tmp_path):
    """optimise() must take identical early-stopping decisions on all ranks: the validation loss is
    computed on rank 0 only and broadcast (ranks diverged and the all-reduce hung)."""
    port = _free_port()
    mp.spawn(_early_stop_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0 = open(os.path.join(str(tmp_path), "r0.txt")).read().split()
    r1 = open(os.path.join(str(tmp_path), "r1.txt")).read().split()
    assert r0 == ["0.25", "1"] and r1 == ["0.25", "0"]


def _failing_validation_worker(rank, world, port, out_dir):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    init_distributed(backend="gloo")
    from wave_u_net_amd import validation

    def fake_test(model_config, partition, model_folder, load_model, tracks=None, **kw):
        raise ValueError("checkpoint is corrupt")          # only rank 0 ever calls it

    validation.test = fake_test
    try:
        validation._rank0_test({}, "valid", "x", None, [])
        got = "no exception"
    except RuntimeError as e:
        got = str(e)
    with open(os.path.join(out_dir, "r%d.txt" % rank), "w") as f:
        f.write(got)
    dist.barrier()
    dist.destroy_process_group()


def test_validation_failure_on_rank0_raises_on_every_rank(tmp_path):
    """A failure of rank 0's validation must not leave the other ranks parked in the broadcast:
    the error text is broadcast and every rank raises."""
    port = _free_port()
    mp.spawn(_failing_validation_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    for r in (0, 1):
        got = open(os.path.join(str(tmp_path), "r%d.txt" % r)).read()
        assert "validation on rank 0 failed" in got and "checkpoint is corrupt" in got, (r, got)


def _sharded_validation_worker(rank, world, port, out_dir, fail_rank):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    init_distributed(backend="gloo")
    from wave_u_net_amd import validation
    seen = []

    def fake_test(model_config, partition, model_folder, load_model, tracks=None, return_sums=False, **kw):
        assert return_sums
        seen.extend(tracks)
        if rank == fail_rank:
            raise ValueError("bad track")
        return float(sum(tracks)), len(tracks)          # one "batch" per track, its loss = the track's number

    validation.test = fake_test
    cfg = {"validation": "sharded", "log_dir": os.path.join(out_dir, "logs")}
    try:
        got = repr(validation._rank0_test(cfg, "valid", "x", None, [1, 2, 3, 4, 5]))
    except RuntimeError as e:
        got = "ERR " + str(e)
    with open(os.path.join(out_dir, "r%d.txt" % rank), "w") as f:
        f.write("%s|%s" % (got, seen))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_validation_is_the_mean_over_all_ranks_batches(tmp_path):
    """model_config["validation"] = "sharded": rank r takes tracks[r::world]; every rank returns the same mean over all
    batches; rank 0 writes the log line."""
    port = _free_port()
    mp.spawn(_sharded_validation_worker, args=(2, port, str(tmp_path), -1), nprocs=2, join=True)
    r0 = open(os.path.join(str(tmp_path), "r0.txt")).read().split("|")
    r1 = open(os.path.join(str(tmp_path), "r1.txt")).read().split("|")
    assert r0[0] == r1[0] == "3.0"                            # (1+2+3+4+5) / 5
    assert r0[1] == "[1, 3, 5]" and r1[1] == "[2, 4]"
    line = open(os.path.join(str(tmp_path), "logs", "x", "test.jsonl")).read()
    assert '"ranks": 2' in line and '"batches": 5' in line


def test_sharded_validation_failure_on_any_rank_raises_on_every_rank(tmp_path):
    port = _free_port()
    mp.spawn(_sharded_validation_worker, args=(2, port, str(tmp_path), 1), nprocs=2, join=True)
    for r in (0, 1):
        got = open(os.path.join(str(tmp_path), "r%d.txt" % r)).read()
        assert got.startswith("ERR sharded validation failed") and "rank 1: ValueError: bad track" in got, (r, got)


def _rank0_checkpoint_worker(rank, world, port, out_dir, corrupt):
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    init_distributed(backend="gloo")
    import torch
    from wave_u_net_amd import validation

    class FakeSeparator:                         # the arena of a separator, without the HIP library
        def __init__(self, cfg):
            self.params = None

        def variables(self):
            if self.params is None:
                self.params = torch.full((8,), float(-1 - rank))
            return {}

    loads = []

    def fake_load(sep, path):
        loads.append(path)
        if corrupt:
            raise OSError("truncated file")
        sep.params.copy_(torch.arange(8, dtype=torch.float32))
        return 7

    validation._load_checkpoint = fake_load
    sep, err = validation._load_on_rank0_and_broadcast({}, "/only/on/rank0.npz", make_separator=FakeSeparator)
    with open(os.path.join(out_dir, "r%d.txt" % rank), "w") as f:
        f.write("%s|%s|%s" % (sep.params.tolist(), err, loads))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_validation_reads_the_checkpoint_on_rank0_only(tmp_path):
    """Only rank 0 is guaranteed to have the checkpoint path (no shared file system): it loads, the parameters are
    broadcast; a load failure reaches every rank as text instead of leaving them in the broadcast."""
    port = _free_port()
    mp.spawn(_rank0_checkpoint_worker, args=(2, port, str(tmp_path), False), nprocs=2, join=True)
    want = str([float(i) for i in range(8)])
    r0 = open(os.path.join(str(tmp_path), "r0.txt")).read().split("|")
    r1 = open(os.path.join(str(tmp_path), "r1.txt")).read().split("|")
    assert r0[0] == r1[0] == want and r0[1] == r1[1] == "None"
    assert r0[2] == "['/only/on/rank0.npz']" and r1[2] == "[]"
    port = _free_port()
    mp.spawn(_rank0_checkpoint_worker, args=(2, port, str(tmp_path), True), nprocs=2, join=True)
    for r in (0, 1):
        got = open(os.path.join(str(tmp_path), "r%d.txt" % r)).read().split("|")
        assert "could not be loaded on rank 0: OSError: truncated file" in got[1], got
