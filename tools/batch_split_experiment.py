#!/usr/bin/env python3
"""Experiment: one B=16 step vs two concurrent B=8 half-batch chains on separate streams
(forward + backward only; gradients of the halves would be averaged before Adam)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import wave_u_net_amd as wun
from wave_u_net_amd.training import synthetic_source

cfg = wun.get_config(sys.argv[1] if len(sys.argv) > 1 else "m1_context")
nsplit = int(sys.argv[2]) if len(sys.argv) > 2 else 2
B = 16
dev = torch.device("cuda:0")
full = wun.UnetAudioSeparator(cfg, device=dev)
i, o = full.get_padding(np.array([B, cfg["num_frames"], 0]))
t_in, t_out = int(i[1]), int(o[1])
mix, targets = synthetic_source(cfg, B, t_in, t_out, dev)()

def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3

full.tune(mix, targets)
def step_full():
    full.get_output(mix, True)
    full.loss_and_gradients(targets)
print("B=16 single chain: %.2f ms (fwd+bwd)" % timeit(step_full))

hb = B // nsplit
seps = [wun.UnetAudioSeparator(cfg, device=dev) for _ in range(nsplit)]
streams = [torch.cuda.Stream(device=dev) for _ in range(nsplit)]
mixes = [mix[k * hb:(k + 1) * hb].contiguous() for k in range(nsplit)]
tgs = [targets[:, k * hb:(k + 1) * hb].contiguous() for k in range(nsplit)]
for k, s in enumerate(seps):
    s.tune(mixes[k], tgs[k])
    if k > 0:
        s.params = seps[0].params
def step_one(k):
    seps[k].get_output(mixes[k], True); seps[k].loss_and_gradients(tgs[k])
print("B=%d alone: %.2f ms" % (hb, timeit(lambda: step_one(0))))
def step_split():
    cur = torch.cuda.current_stream(dev)
    for s in streams: s.wait_stream(cur)
    for k in range(nsplit):
        with torch.cuda.stream(streams[k]): seps[k].get_output(mixes[k], True)
    for k in range(nsplit):
        with torch.cuda.stream(streams[k]): seps[k].loss_and_gradients(tgs[k])
    for s in streams: cur.wait_stream(s)
print("%d x B=%d concurrent chains: %.2f ms (fwd+bwd)" % (nsplit, hb, timeit(step_split)))
