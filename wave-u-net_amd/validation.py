"""Validation loss and the early-stopping driver around the training path: the reference's
Test.test (/root/reference/Test.py:11-92) and Training.optimise (/root/reference/Training.py:123-151).

test() = mean over the partition's batches of  (1/S) sum_src mean((real - est)^2)  with the
separator run in inference mode (training=False, so the "linear" head clips, Test.py:34), as a running
mean over batches (Test.py:77-84).  optimise() = epochs of train() until the validation loss has
not improved for `worse_epochs` epochs, then a fine-tuning round with batch_size doubled and
init_sup_sep_lr = 1e-5, then the test-partition loss of the best checkpoint.
"""
import json
import os

import numpy as np
import torch

from . import datasets
from .separator import UnetAudioSeparator
from .training import train
from .checkpoint import load_checkpoint


def _load_checkpoint(separator, load_model):
    return load_checkpoint(separator, load_model, with_optimizer=False)


def test(model_config, partition, model_folder, load_model, tracks=None, data_root=None, separator=None,
         return_sums=False):
    """Test.test(model_config, partition, model_folder, load_model) -> mean MSE.  The partition's
    tracks come from `tracks` (list of track dicts) or data_root/<partition>/<track>/.
    return_sums: (sum of the per-batch losses, number of batches) instead, nothing logged."""
    if model_config["network"] != "unet":
        raise NotImplementedError(model_config["network"])                        # Test.py:14-19
    if tracks is None:
        if data_root is None:
            raise ValueError("test() needs `tracks` or `data_root`")
        tracks = datasets.load_partition(data_root, partition, model_config)
    sep = separator if separator is not None else UnetAudioSeparator(model_config)
    disc_input_shape = [model_config["batch_size"], model_config["num_frames"], 0]
    in_shape, out_shape = sep.get_padding(np.array(disc_input_shape))            # Test.py:21
    assert (in_shape[1] - out_shape[1]) % 2 == 0                                  # Test.py:25
    global_step = _load_checkpoint(sep, load_model) if load_model is not None else 0

    total_loss, batch_num, loss_sum = 0.0, 1, 0.0
    names = list(model_config["source_names"])
    for batch in datasets.get_dataset(model_config, in_shape, out_shape, partition, tracks):
        outs = sep.get_output(batch["mix"], False)                                # Test.py:34
        loss = 0.0
        for key in names:                                                         # Test.py:61-74
            real = torch.as_tensor(batch[key], device=outs[key].device)
            loss = loss + torch.mean((real - outs[key]) ** 2)
        curr = float(loss.item()) / float(model_config["num_sources"])
        total_loss = total_loss + (1.0 / float(batch_num)) * (curr - total_loss)  # Test.py:80
        loss_sum += curr
        batch_num += 1
    if return_sums:                                                               # a shard of the partition (_sharded_test)
        return loss_sum, batch_num - 1

    log_dir = os.path.join(model_config["log_dir"], str(model_folder))            # Test.py:41,86-87
    os.makedirs(log_dir, exist_ok=True)
    with open(os.path.join(log_dir, "test.jsonl"), "a") as f:
        f.write(json.dumps({"global_step": global_step, "partition": partition, "test_loss": total_loss}) + "\n")
    return total_loss


def optimise(model_config, experiment_id, data=None, data_root=None, max_epochs=None):
    """Training.optimise -> (best_model_path, test_loss).  `data` = {"train": [...], "valid": [...],
    "test": [...]} track lists (or data_root with those sub-directories).  max_epochs bounds the
    total number of epochs (the reference has no bound; tests need one)."""
    if data is None:
        if data_root is None:
            raise ValueError("optimise() needs `data` or `data_root`")
        data = {part: datasets.load_partition(data_root, part, model_config) for part in ("train", "valid", "test")}
    model_config = dict(model_config)
    epoch = 0
    best_loss = 10000
    model_path = None
    best_model_path = None
    for i in range(2):
        worse_epochs = 0
        if i == 1:                                                                # Training.py:131-134
            model_config["batch_size"] *= 2
            model_config["init_sup_sep_lr"] = 1e-5
        while worse_epochs < model_config["worse_epochs"]:                        # Training.py:135
            if max_epochs is not None and epoch >= max_epochs:
                break
            model_path = train(model_config, experiment_id, load_model=model_path,
                               batch_source=_train_source(model_config, data["train"], seed=1337 + epoch))
            curr_loss = _rank0_test(model_config, "valid", str(experiment_id), model_path, data["valid"])
            epoch += 1
            if curr_loss < best_loss:
                worse_epochs = 0
                best_model_path = model_path
                best_loss = curr_loss
            else:
                worse_epochs += 1
    test_loss = _rank0_test(model_config, "test", str(experiment_id), best_model_path, data["test"])
    return best_model_path, test_loss


def _load_on_rank0_and_broadcast(model_config, load_model, make_separator=None):
    """-> (separator holding the checkpoint's parameters on every rank, error text or None).  Rank 0 reads the file;
    a failure there is broadcast as text so that no rank is left waiting in the parameter broadcast."""
    import torch.distributed as dist
    from .parallel import broadcast_parameters
    sep = (make_separator or UnetAudioSeparator)(model_config)
    sep.variables()                                  # (creates the parameter arena)
    box = [None]
    if dist.get_rank() == 0:
        try:
            _load_checkpoint(sep, load_model)
        except Exception as e:                       # noqa: BLE001 -- reported on every rank by the caller
            box[0] = "checkpoint %s could not be loaded on rank 0: %s: %s" % (load_model, type(e).__name__, e)
    dist.broadcast_object_list(box, src=0)
    if box[0] is None:
        broadcast_parameters(sep.params)
    return sep, box[0]


def _sharded_test(model_config, partition, model_folder, load_model, tracks):
    """model_config["validation"] = "sharded": rank r evaluates tracks[r::world] and the ranks all-reduce (sum of batch
    losses, batch count) -- no GPU idles through a validation pass and nothing waits in a broadcast for rank 0's whole
    partition.  The loss is the mean over all ranks' batches: identical on every rank, equal to
    test()'s for one rank, and differing from the single-process value only through where the snippet stream of a
    shard is cut into batches (the reference weighs a short last batch like a full one too, Test.py:80).  A failure on
    any rank is raised on every rank."""
    import torch.distributed as dist
    rank, world = dist.get_rank(), dist.get_world_size()
    sums, err = (0.0, 0), None
    # the checkpoint is read on rank 0 only -- the one rank train() guarantees to have the path; the file system need
    # not be shared -- and its parameters are broadcast into every rank's separator
    sep, load_err = None, None
    if load_model is not None:
        sep, load_err = _load_on_rank0_and_broadcast(model_config, load_model)
    try:
        if load_err:
            raise RuntimeError(load_err)
        mine = list(tracks)[rank::world]
        if mine:
            sums = test(model_config, partition, model_folder, load_model if sep is None else None, tracks=mine,
                        return_sums=True, separator=sep)
    except Exception as e:                           # noqa: BLE001 -- re-raised below, on EVERY rank
        err = "rank %d: %s: %s" % (rank, type(e).__name__, e)
    errs = [None] * world
    dist.all_gather_object(errs, err)
    bad = [e for e in errs if e]
    if bad:
        raise RuntimeError("sharded validation failed: " + "; ".join(bad))
    t = torch.tensor([float(sums[0]), float(sums[1])], dtype=torch.float64)
    if dist.get_backend() == "nccl":
        t = t.cuda()
    dist.all_reduce(t)
    if float(t[1]) == 0:
        raise RuntimeError("sharded validation: the partition produced no batch on any rank")
    loss = float(t[0] / t[1])
    if rank == 0:
        log_dir = os.path.join(model_config["log_dir"], str(model_folder))
        os.makedirs(log_dir, exist_ok=True)
        with open(os.path.join(log_dir, "test.jsonl"), "a") as f:
            f.write(json.dumps({"partition": partition, "test_loss": loss, "ranks": world, "batches": int(t[1])}) + "\n")
    return loss


def _rank0_test(model_config, partition, model_folder, load_model, tracks):
    """test() on rank 0 only, its loss broadcast to every rank: the early-stopping decisions of
    optimise() (worse_epochs, best checkpoint, loop exit) are then identical on all ranks -- ranks
    that disagreed would leave the next epoch's gradient all-reduce waiting forever.
    (model_config["validation"] = "sharded" evaluates on all ranks instead: _sharded_test.)"""
    import torch.distributed as dist
    multi = dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1
    if multi and model_config.get("validation") == "sharded":
        return _sharded_test(model_config, partition, model_folder, load_model, tracks)
    loss, err = None, None
    if not multi or dist.get_rank() == 0:
        try:
            loss = test(model_config, partition, model_folder, load_model, tracks=tracks)
        except Exception as e:                       # noqa: BLE001 -- re-raised below, on EVERY rank
            if not multi:
                raise
            err = "%s: %s" % (type(e).__name__, e)
    if multi:
        # rank 0's failure (bad checkpoint, ValueError from test()) reaches the waiting ranks instead of leaving them
        # parked in the collective until the watchdog fires
        box = [loss, err]
        dist.broadcast_object_list(box, src=0)
        loss, err = box
        if err is not None:
            raise RuntimeError("validation on rank 0 failed: " + err)
    return loss


def _train_source(model_config, tracks, seed):
    """Deferred constructor: train() builds the Trainer first (it fixes device and shapes), then
    asks for the batch source."""
    def make(trainer):
        return datasets.DeviceSnippetSource(model_config, tracks, trainer.t_in, trainer.t_out, trainer.batch,
                                            trainer.device, seed=seed + trainer.rank)
    make.needs_trainer = True
    return make
