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


def main():
    out = sys.argv[1]
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
