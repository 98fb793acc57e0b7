#!/usr/bin/env python3
"""Why a float32 evaluation of the reference graph sits ~1e-4 .. 5e-4 (of max|g|) away from the float64 oracle on a
few weight-gradient tensors, whatever the kernels: LeakyReLU'(x) jumps from 0.2 to 1 at x = 0, and among the ~1e8
pre-activations of one M1 + context batch a handful lie within float32 rounding of zero.  Their sign -- hence a factor
5 on that element's gradient -- depends on the summation order of the conv that produced them.

CPU only (the torch oracle, nothing of the product): runs the oracle in float64 and in float32 on the inputs of
tests/test_gpu_parity.py::test_autotuned_full_size_m1_context, counts the pre-activations whose sign differs, reports
the float32 oracle's own per-tensor gradient deviation from float64, and then re-runs the FLOAT64 backward with the
float32 run's LeakyReLU masks: if the deviation is reproduced, the masks explain it.
usage: python tools/lrelu_flip_study.py [batch=2] [out.json]"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import shapes, waveunet_torch as wt
from oracle.golden_params import golden_params

def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    out = sys.argv[2] if len(sys.argv) > 2 else None
    ocfg = shapes.finalize_config(dict(shapes.BASE_MODEL_CONFIG, context=True))
    params = golden_params(ocfg, 77)
    i, o = shapes.get_padding(ocfg, [B, 16384, 0])
    mix, targets = wt.synthetic_batch(ocfg, B, i[1], o[1], seed=78)
    rec = {"pre": None, "force": None, "k": 0}
    orig_apply = wt._TfLeakyReLU.apply

    class Forced(torch.autograd.Function):
        """LeakyReLU whose derivative mask is given (the forward value is the true one)."""
        @staticmethod
        def forward(ctx, x, mask):
            ctx.save_for_backward(mask)
            return torch.maximum(0.2 * x, x)
        @staticmethod
        def backward(ctx, g):
            (mask,) = ctx.saved_tensors
            return g * torch.where(mask, torch.ones_like(g), torch.full_like(g, 0.2)), None

    def lrelu(x):
        k = rec["k"]; rec["k"] += 1
        if rec["pre"] is not None:
            rec["pre"].append(x.detach().clone())
        if rec["force"] is not None:
            return Forced.apply(x, rec["force"][k])
        return orig_apply(x)
    wt.leaky_relu = lrelu

    def run(dtype, force=None, keep=False):
        rec["k"] = 0; rec["pre"] = [] if keep else None; rec["force"] = force
        tp = wt.params_to_torch(params, dtype, requires_grad=True)
        loss, grads = wt.train_step(ocfg, tp, torch.tensor(mix, dtype=dtype), {k: torch.tensor(v, dtype=dtype) for k, v in targets.items()})
        pre = rec["pre"]; rec["pre"] = None; rec["force"] = None
        return loss.item(), [g.double() for g in grads], pre

    torch.set_num_threads(max(1, (os.cpu_count() or 2) - 1))
    l64, g64, pre64 = run(torch.float64, keep=True)
    l32, g32, pre32 = run(torch.float32, keep=True)
    flips, total, tiny = 0, 0, []
    per_layer = []
    for a, b in zip(pre64, pre32):
        d = (a > 0) != (b.double() > 0)
        n = int(d.sum())
        flips += n; total += a.numel()
        per_layer.append(n)
        if n:
            tiny.append(float((a[d].abs() / a.abs().mean()).max()))
    names = [n for n, _ in params]
    def dev(gs):
        rows = []
        for n, g, r in zip(names, gs, g64):
            rows.append(((g - r).abs().max().item() / max(r.abs().max().item(), 1e-30), ((g - r).norm() / max(r.norm().item(), 1e-30)).item(), n))
        return sorted(rows, reverse=True)
    d32 = dev(g32)
    # float64 arithmetic, float32 masks
    masks = [(p.double() > 0) for p in pre32]
    _, g64m, _ = run(torch.float64, force=masks)
    dm = dev(g64m)
    # the pre-activations nearest to zero (relative to their layer's mean magnitude): flip ONE mask bit each, float64
    # arithmetic otherwise -- the size of the gradient change a single sign decision carries
    cands = []
    for li_, a in enumerate(pre64):
        flat = a.contiguous().abs().flatten()
        v, idx = torch.topk(flat, 1, largest=False)
        cands.append((float(v[0] / a.abs().mean()), li_, int(idx[0])))
    cands.sort()
    base_masks = [(p > 0) for p in pre64]
    single = []
    for relmag, li_, idx in cands[:4]:
        masks1 = [m.contiguous().clone() for m in base_masks]
        flat = masks1[li_].view(-1)
        flat[idx] = ~flat[idx]
        _, g1, _ = run(torch.float64, force=masks1)
        d1 = dev(g1)
        single.append({"leaky_relu_index": li_, "shape": list(pre64[li_].shape), "|x|_over_layer_mean": relmag,
                       "worst": [{"tensor": n, "max_err_over_max_ref": a, "rel_l2": b} for a, b, n in d1[:4]]})
    # float32 oracle under last-bit input perturbations (the float64 reference re-computed on the SAME perturbed input):
    # how far does a float32 evaluation of the reference arithmetic itself jump?
    rng = np.random.default_rng(5)
    perturbed = []
    for t in range(7):
        m = (mix * (1.0 + 3e-7 * rng.standard_normal(mix.shape))).astype(np.float32)
        def run_on(dtype):
            tp = wt.params_to_torch(params, dtype, requires_grad=True)
            _, grads = wt.train_step(ocfg, tp, torch.tensor(m, dtype=dtype), {k: torch.tensor(v, dtype=dtype) for k, v in targets.items()})
            return [g.double() for g in grads]
        r64, r32 = run_on(torch.float64), run_on(torch.float32)
        rows = sorted([((a - b).abs().max().item() / max(b.abs().max().item(), 1e-30), n) for n, a, b in zip(names, r32, r64)], reverse=True)
        perturbed.append({"perturbation": t + 1, "worst": [{"tensor": n, "max_err_over_max_ref": e} for e, n in rows[:3]],
                          "tensors_above_1e-5": sum(1 for e, _ in rows if e > 1e-5)})
    res = {
        "float32_oracle_vs_float64_under_3e-7_input_perturbations": perturbed,
        "single_mask_flips_of_the_inputs_nearest_zero": single,
        "config": "M1 + context, B=%d, %d -> %d samples, golden_params seed 77, synthetic_batch seed 78" % (B, i[1], o[1]),
        "leaky_relu_inputs": total, "sign_differs_float32_vs_float64": flips, "per_layer": per_layer,
        "largest_|x|_of_a_flipped_input_over_mean|x|": max(tiny) if tiny else 0.0,
        "loss_rel_diff_float32": abs(l32 - l64) / abs(l64),
        "float32_oracle_vs_float64_worst": [{"tensor": n, "max_err_over_max_ref": a, "rel_l2": b} for a, b, n in d32[:6]],
        "float64_with_float32_masks_vs_float64_worst": [{"tensor": n, "max_err_over_max_ref": a, "rel_l2": b} for a, b, n in dm[:6]],
    }
    print(json.dumps(res, indent=1))
    if out:
        json.dump(res, open(out, "w"), indent=1)

if __name__ == "__main__":
    main()
