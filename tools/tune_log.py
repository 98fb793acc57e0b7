import os, sys
sys.path.insert(0, '/root/repo')
import torch
import wave_u_net_amd as wun
from wave_u_net_amd.training import Trainer, synthetic_source
cfg = wun.get_config("m1_context")
tr = Trainer(cfg, batch_size=16)
mix, targets = synthetic_source(cfg, tr.batch, tr.t_in, tr.t_out, tr.device, seed=1337)()
tr.tune(mix, targets)
