#!/usr/bin/env python3
"""Experiment: replay one training step as a captured HIP graph vs eager launches."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import wave_u_net_amd as wun
from wave_u_net_amd.training import Trainer, synthetic_source

cfg = wun.get_config("m1_context")
tr = Trainer(cfg, batch_size=16)
mix, targets = synthetic_source(cfg, tr.batch, tr.t_in, tr.t_out, tr.device)()
tr.tune(mix, targets)
for _ in range(3): tr.step(mix, targets)
torch.cuda.synchronize()

def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
print("eager: %.2f ms/step" % timeit(lambda: tr.step(mix, targets)))
g = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(g):
        tr.sep.get_output(mix, True)
        loss = tr.sep.loss_and_gradients(targets)
    print("graph (fwd+bwd) replay: %.2f ms" % timeit(g.replay))
    def eager_fb():
        tr.sep.get_output(mix, True); tr.sep.loss_and_gradients(targets)
    print("eager (fwd+bwd): %.2f ms" % timeit(eager_fb))
except Exception as e:
    print("capture failed:", repr(e)[:500])
