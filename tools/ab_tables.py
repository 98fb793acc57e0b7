#!/usr/bin/env python3
"""A/B of tuning tables on one box: median step time of each table, interleaved over several rounds.
usage: python tools/ab_tables.py [--config m1_context] tableA tableB [...]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--config", default="m1_context")
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("tables", nargs="+")
args = ap.parse_args()

import wave_u_net_amd as wun
from wave_u_net_amd.training import Trainer, synthetic_source

cfg = wun.get_config(args.config)
tr = Trainer(cfg, batch_size=args.batch)
mix, targets = synthetic_source(cfg, tr.batch, tr.t_in, tr.t_out, tr.device, seed=1337)()
res = {t: [] for t in args.tables}
for _ in range(args.rounds):
    for t in args.tables:
        tr.sep.tune_import(open(t).read())
        for _ in range(5):
            tr.step(mix, targets)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
        ev[0].record()
        for i in range(args.steps):
            tr.step(mix, targets)
            ev[i + 1].record()
        torch.cuda.synchronize()
        res[t].append(float(np.median([ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps)])))
for t in args.tables:
    print("%-40s %s  min %.4f ms" % (t, " ".join("%.4f" % x for x in res[t]), min(res[t])))
