"""Worker of tests/test_data_parallel_gpu.py: one rank of a data-parallel training run.
Launched by torch.distributed.run; writes rank 0's parameters after a few steps."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import wave_u_net_amd as wun                         # noqa: E402
from wave_u_net_amd import training                  # noqa: E402


def make_cfg():
    return wun.get_config("full", num_layers=4, num_initial_filters=8, num_frames=72, batch_size=3,
                          init_sup_sep_lr=1e-3)


def global_batch(cfg, t_in, t_out, n):
    src = training.synthetic_source(cfg, n, t_in, t_out, "cpu", seed=99)
    return src()


def resume_main(out):
    """Two epochs of training.train with a checkpoint hand-over in between (what validation.optimise
    does): every rank must come out with the same parameters, Adam slots and global_step."""
    cfg = dict(make_cfg(), epoch_it=2, model_base_dir=os.path.join(os.path.dirname(out), "ckpt"),
               log_dir=os.path.join(os.path.dirname(out), "logs"))
    path = training.train(cfg, "dp", None)
    assert path is not None and path.endswith("dp-2.npz"), path           # broadcast to every rank
    tr_holder = {}
    orig = training.Trainer

    class Spy(orig):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            tr_holder["tr"] = self
    training.Trainer = Spy
    path2 = training.train(cfg, "dp", load_model=path)
    tr = tr_holder["tr"]
    assert path2.endswith("dp-4.npz"), path2
    np.savez("%s.rank%d.npz" % (out, tr.rank), params=tr.sep.params.cpu().numpy(), m=tr.sep.adam_m.cpu().numpy(),
             v=tr.sep.adam_v.cpu().numpy(), step=tr.sep.global_step, table=np.array(getattr(tr, "tune_table", "") or ""))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def main():
    out = sys.argv[1]
    if sys.argv[2] == "resume":
        return resume_main(out)
    steps = int(sys.argv[2])
    cfg = make_cfg()
    tr = training.Trainer(cfg)
    mix, targets = global_batch(cfg, tr.t_in, tr.t_out, tr.batch * tr.world)
    lo = tr.rank * tr.batch
    mix = mix[lo:lo + tr.batch].to(tr.device).contiguous()
    targets = targets[:, lo:lo + tr.batch].to(tr.device).contiguous()
    losses = [float(tr.step(mix, targets).item()) for _ in range(steps)]
    torch.cuda.synchronize()
    if tr.rank == 0:
        np.savez(out, params=tr.sep.params.cpu().numpy(), losses=np.array(losses), world=tr.world,
                 overlap=int(tr.overlap))
    if tr.world > 1:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
