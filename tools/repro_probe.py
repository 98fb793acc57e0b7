#!/usr/bin/env python3
"""Is a training step bit-reproducible across fresh separators of one process?  (GPU box.)

Runs a handful of small configurations first (process history: earlier plans, their streams, recycled allocations), then
the M4 configuration (BASELINE.json configs[2]: context + stereo + difference output, 147443 -> 16389 samples, B = 2) nine
times on fresh separators and compares loss, outputs, every gradient tensor and every tensor the step leaves in the
workspace (wun_plan_activation kinds 0 - 9) BITWISE with the first run.  This is the probe that found the bf16 mode's
head weight gradient differing from run to run when wgrad_bf16_kernel ran beside narrow_wgrad_kernel;
(the library built WITH packed fp32 ops: make -C wave-u-net_amd/csrc pkdiag; WUN_LIB=libwun_pk.so shows the defect again; WUN_BF16_HEAD_SERIAL=1 is round 5's serialised placement).
usage: python tools/repro_probe.py [bf16|f32] [m4|m4_same|m1_context|m5|multi|multi_direct]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import wave_u_net_amd as wun
from wave_u_net_amd.separator import UnetAudioSeparator
from oracle import shapes, waveunet_torch as wt
from oracle.golden_params import GOLDEN_CASES, golden_params

KINDS = ["dec", "skip", "dz_skip", "dz_dec", "ups", "up", "dz_up", "d_ups"]

def step(over, seed_p, B, frames, seed_d, dtype="bf16", collect=True):
    ocfg = shapes.finalize_config(dict(shapes.BASE_MODEL_CONFIG, **over))
    params = golden_params(ocfg, seed_p)
    sep = UnetAudioSeparator(wun.get_config("baseline", compute_dtype=dtype, **over), device="cuda:0")
    i, o = shapes.get_padding(ocfg, [B, frames, 0])
    mix, targets = wt.synthetic_batch(ocfg, B, i[1], o[1], seed=seed_d)
    sep._plan(B, i[1]); sep._active = sep._plans[(B, i[1])]
    sep.load_variables(params)
    outs = sep.get_output(torch.from_numpy(mix).cuda(), True)
    loss = sep.loss_and_gradients({k: torch.from_numpy(v) for k, v in targets.items()})
    torch.cuda.synchronize()
    res = {"loss": loss.detach().cpu().clone()}
    for n, t in sep.gradients().items(): res["g:" + n] = t.detach().cpu().clone()
    for n, t in outs.items(): res["o:" + n] = t.detach().cpu().clone()
    if collect:
        L = ocfg["num_layers"]
        for k in range(L):
            for kind in KINDS:
                if kind == "dz_dec" and not ocfg["context"]: continue
                res["%s%d" % (kind, k)] = sep.activation(kind, k)[0].cpu().clone()
        res["bottleneck"] = sep.activation("bottleneck")[0].cpu().clone()
        res["dz_bottleneck"] = sep.activation("dz_bottleneck")[0].cpu().clone()
    return res

CONFIGS = {
    "m4": dict(output_type="difference", context=True, mono_downmix=False),                       # BASELINE.json configs[2]
    "m4_same": dict(output_type="difference", context=False, mono_downmix=False),
    "m1_context": dict(context=True),                                                             # the headline shape (mono)
    "m5": dict(output_type="difference", context=True, mono_downmix=False, upsampling="learned"),
    "multi": dict(output_type="difference", context=True, mono_downmix=False, task="multi_instrument"),
    "multi_direct": dict(context=True, mono_downmix=False, task="multi_instrument"),
}
dt = sys.argv[1] if len(sys.argv) > 1 else "bf16"
which = sys.argv[2] if len(sys.argv) > 2 else "m4"
over = CONFIGS[which]
for name in ["baseline_small", "baseline_stereo_small", "full_small", "full_multi_small", "learned_same_small"]:
    case = GOLDEN_CASES[name]
    step(case["cfg"], case["seed"], 3, case["frames"], case["seed"] + 100, collect=False)
ref = step(over, 91, 2, 16384, 92, dt)
nbad = 0
NREP = int(os.environ.get('REPRO_REPEATS', '8'))
for r in range(NREP):
    got = step(over, 91, 2, 16384, 92, dt)
    diffs = []
    for k in ref:
        if not torch.equal(got[k], ref[k]):
            a, b = got[k].double().flatten(), ref[k].double().flatten()
            nz = (a != b).nonzero().flatten()
            diffs.append("%s: %d of %d differ (first idx %d..%d), max rel %.2e" % (k, nz.numel(), a.numel(), int(nz[0]), int(nz[-1]),
                         float((a - b).abs().max() / max(b.abs().max().item(), 1e-30))))
    nbad += bool(diffs)
    if len(diffs) > 6:
        # a whole-step event: name the first tensors that differ in EXECUTION order (forward: dec_i / skip_i per level,
        # bottleneck, ups_j / up_j per up level; then the backward pass in reverse)
        L = 12
        order = []
        for i in range(L): order += ["dec%d" % i, "skip%d" % i]
        order += ["bottleneck"]
        for j in range(L): order += ["ups%d" % j, "up%d" % j]
        order += [k for k in ref if k.startswith("o:")] + ["loss"]
        for j in range(L - 1, -1, -1): order += ["dz_up%d" % j, "dz_skip%d" % (L - 1 - j), "d_ups%d" % j]
        order += ["dz_bottleneck"] + ["dz_dec%d" % i for i in range(L - 1, -1, -1)]
        first = [k for k in order if k in ref and not torch.equal(got[k], ref[k])][:4]
        firstd = []
        for k in first:
            a, b = got[k].double(), ref[k].double()
            nz = (a != b).nonzero()
            firstd.append("%s: %d of %d differ, first at %s, last at %s" % (k, nz.shape[0], a.numel(), nz[0].tolist(), nz[-1].tolist()))
        diffs = ["FIRST IN EXECUTION ORDER: " + " ; ".join(firstd)] + diffs[:4] + ["... %d tensors in all" % len(diffs)]
    print("[%s %s] repeat %d: %s" % (which, dt, r, "bitwise identical" if not diffs else " | ".join(diffs)), flush=True)
print("[%s %s] %d of %d repeats differ from the first run" % (which, dt, nbad, NREP))
