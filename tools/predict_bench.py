#!/usr/bin/env python3
"""Full-track inference throughput (evaluate.predict_track, Evaluate.py:82-145): a synthetic
3-minute 22.05 kHz track through the M1+context separator, hops batched 16 at a time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import wave_u_net_amd as wun
from wave_u_net_amd.evaluate import predict_track

name = sys.argv[1] if len(sys.argv) > 1 else "m1_context"
cfg = wun.get_config(name)
sep = wun.UnetAudioSeparator(cfg, device="cuda:0")
n = 180 * cfg["expected_sr"]
C = 1 if cfg["mono_downmix"] else 2
audio = np.random.default_rng(0).uniform(-0.5, 0.5, (n, C)).astype(np.float32)
predict_track(cfg, sep, audio[: 20 * cfg["expected_sr"]])           # warm-up (plan creation)
torch.cuda.synchronize()
t0 = time.perf_counter()
preds = predict_track(cfg, sep, audio, batch_hops=16)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print("%s: %d samples (%.0f s of audio) separated in %.3f s = %.1f M samples/s (%.0fx real time), host tiling included"
      % (name, n, n / cfg["expected_sr"], dt, n / dt / 1e6, n / cfg["expected_sr"] / dt))
