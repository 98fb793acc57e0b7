#!/usr/bin/env python3
"""Whole-step tuner: refines a tuning table by measuring the TRAINING STEP, not isolated launches.

wun_plan_tune times every candidate tile of a launch position on its own (one stream).  In the real step three
streams run concurrently, so the best tile in isolation is not always the best tile beside two other kernels
(register / LDS co-residency, tails that another stream can or cannot fill).  This tool starts from the isolated
table, takes for every launch position the near-best candidates the library logged (WUN_TUNE_ALTS), and does
coordinate descent on the median step time: a change is kept if it improves the median of `--steps` steps by more
than `--gain` and the improvement is confirmed by a second measurement against the incumbent.

usage: python tools/step_tune.py --out profiles/round4_tune_table.txt [--config m1_context] [--passes 2]
(runs on the GPU box; ~1 minute per pass)"""
import argparse
import os
import sys
import tempfile
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

ap = argparse.ArgumentParser()
ap.add_argument("--out", required=True)
ap.add_argument("--config", default="m1_context")
ap.add_argument("--batch", type=int, default=16)
ap.add_argument("--steps", type=int, default=12)
ap.add_argument("--passes", type=int, default=2)
ap.add_argument("--gain", type=float, default=0.0012)
ap.add_argument("--start", default=None, help="start from this table instead of a fresh isolated tuning pass (alternatives are still collected)")
args = ap.parse_args()

alts_file = tempfile.mktemp(prefix="wun_alts_")
os.environ["WUN_TUNE_ALTS"] = alts_file
os.environ.pop("WUN_TUNE_CACHE", None)

import wave_u_net_amd as wun
from wave_u_net_amd.training import Trainer, synthetic_source

cfg = wun.get_config(args.config)
tr = Trainer(cfg, batch_size=args.batch)
mix, targets = synthetic_source(cfg, tr.batch, tr.t_in, tr.t_out, tr.device, seed=1337)()
t0 = time.time()
tr.tune(mix, targets)                                   # isolated tuning pass; logs the alternatives
table = tr.tune_table
print("isolated tuning pass: %.1f s, %d alternative lines" % (time.time() - t0, sum(1 for _ in open(alts_file))), flush=True)
if args.start:
    table = open(args.start).read()
    tr.sep.tune_import(table)

lines = table.strip().split("\n")
header, body = lines[0], lines[1:-1]
assert lines[-1] == "end"
kinds = [ln.split()[0] for ln in body]
first = {k: kinds.index(k) for k in ("cf", "cb", "wg") if k in kinds}

alts = {}                                               # (kind, idx) -> [line text]
for ln in open(alts_file):
    f = ln.split()
    kind, idx = f[0], int(f[1])
    text = "%s %s" % (kind, " ".join(f[2:-1]))
    alts.setdefault((kind, idx), [])
    if text not in alts[(kind, idx)]:
        alts[(kind, idx)].append(text)


def make_table(b):
    return "\n".join([header] + b + ["end"]) + "\n"


def measure(b, steps=args.steps):
    tr.sep.tune_import(make_table(b))
    for _ in range(2):
        tr.step(mix, targets)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    ev[0].record()
    for i in range(steps):
        tr.step(mix, targets)
        ev[i + 1].record()
    torch.cuda.synchronize()
    return float(np.median([ev[i].elapsed_time(ev[i + 1]) for i in range(steps)]))


cur = list(body)
best = measure(cur, 30)
print("start: %.4f ms/step (median)" % best, flush=True)
for ps in range(args.passes):
    changed = 0
    # heaviest positions first: order by the isolated time of the incumbent where known (just use table order reversed
    # for the backward pass, whose big launches come last)
    for (kind, idx), cands in sorted(alts.items(), key=lambda kv: (kv[0][0], -kv[0][1])):
        pos = first[kind] + idx
        if pos >= len(cur) or kinds[pos] != kind:
            continue
        for text in cands:
            if text == cur[pos]:
                continue
            trial = list(cur)
            trial[pos] = text
            t = measure(trial)
            if t < best * (1 - args.gain):
                # confirm against the incumbent, back to back
                t_inc = measure(cur)
                t2 = measure(trial)
                if t2 < t_inc * (1 - args.gain / 2):
                    print("  pass %d %s #%d: '%s' -> '%s'  %.4f -> %.4f ms" % (ps, kind, idx, cur[pos], text, t_inc, t2), flush=True)
                    cur, best, changed = trial, min(t2, t), changed + 1
    print("pass %d: %d changes, %.4f ms/step" % (ps, changed, measure(cur, 30)), flush=True)
    if not changed:
        break
final = measure(cur, 30)
base = measure(list(body), 30)
print("isolated table %.4f ms/step, whole-step table %.4f ms/step" % (base, final), flush=True)
with open(args.out, "w") as f:
    f.write(make_table(cur if final <= base else list(body)))
print("wrote", args.out)
