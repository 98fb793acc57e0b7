#!/usr/bin/env python3
"""Which launch choice of a tuned plan moves a gradient tensor away from the float64 oracle?

M1 + context, B = 2, full length (the case tests/test_gpu_parity.py::test_autotuned_full_size_m1_context checks):
autotune once, then re-run the SAME inputs with (a) the heuristic plan, (b) the tuned table, (c) the tuned table with
one category (cf / cb / wg) reset to the heuristic, (d) each entry of the guilty category alone -- and print per
gradient tensor max-error / max|ref| and relative L2 against the float64 oracle.
usage: python tools/parity_bisect.py [entry-scan category: cf|cb|wg]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import wave_u_net_amd as wun
from wave_u_net_amd import UnetAudioSeparator
from oracle import shapes, waveunet_torch as wt          # checker only
from oracle.golden_params import golden_params

def main():
    scan = sys.argv[1] if len(sys.argv) > 1 else None
    ocfg = shapes.finalize_config(dict(shapes.BASE_MODEL_CONFIG, context=True))
    params = golden_params(ocfg, 77)
    B = 2
    i, o = shapes.get_padding(ocfg, [B, 16384, 0])
    mix, targets = wt.synthetic_batch(ocfg, B, i[1], o[1], seed=78)
    tg = {k: torch.from_numpy(v) for k, v in targets.items()}
    oloss, ograds, oouts = wt.chunked_train_step(ocfg, params, mix, targets, dtype=torch.float64, chunk=1, want_outputs=True)
    names = [n for n, _ in params]

    def run(table, label):
        sep = UnetAudioSeparator(wun.get_config("m1_context"), device="cuda:0")
        sep._plan(B, i[1]); sep._active = sep._plans[(B, i[1])]
        sep.load_variables(params)
        dmix = torch.from_numpy(mix).cuda()
        if table == "tune":
            sep.tune(dmix, tg)
            text = sep.tune_export()
        elif table is not None:
            sep.get_output(dmix, True)
            sep.tune_import(table)
            text = table
        else:
            text = None
        sep.get_output(dmix, True)
        sep.loss_and_gradients(tg)
        torch.cuda.synchronize()
        g = sep.gradients()
        rows = []
        for n, og in zip(names, ograds):
            got = g[n].cpu().double(); og = og.double()
            rows.append(((got - og).abs().max().item() / max(og.abs().max().item(), 1e-30),
                         ((got - og).norm() / max(og.norm().item(), 1e-30)).item(), n))
        rows.sort(reverse=True)
        print("%-34s worst %.3e (%s)  relL2 worst %.3e | top: %s" % (
            label, rows[0][0], rows[0][2].replace("separator/", ""), max(r[1] for r in rows),
            " ".join("%s=%.1e" % (r[2].replace("separator/", ""), r[0]) for r in rows[:4])))
        sys.stdout.flush()
        return text, rows

    run(None, "heuristic plan")
    text, _ = run("tune", "autotuned")
    lines = text.strip().split("\n")
    head, body = lines[0], lines[1:]
    def reset(cat, keep=None):
        out = []
        k = 0
        for ln in body:
            if ln.startswith(cat + " "):
                if keep is not None and k == keep:
                    out.append(ln)
                else:
                    out.append("wg 0 0 0 0" if cat == "wg" else "%s -1 0" % cat)
                k += 1
            else:
                out.append(ln)
        return head + "\n" + "\n".join(out) + "\n"
    for cat in ("cf", "cb", "wg"):
        run(reset(cat), "tuned, %s reset to heuristic" % cat)
    if scan:
        n = sum(1 for ln in body if ln.startswith(scan + " "))
        base = text
        for cat in ("cf", "cb", "wg"):
            if cat != scan:
                lines2 = base.strip().split("\n")
                base = lines2[0] + "\n" + "\n".join(("wg 0 0 0 0" if cat == "wg" else "%s -1 0" % cat) if ln.startswith(cat + " ") else ln for ln in lines2[1:]) + "\n"
        for k in range(n):
            lines2 = base.strip().split("\n")
            out, j = [], 0
            for ln in lines2[1:]:
                if ln.startswith(scan + " "):
                    out.append(ln if j == k else ("wg 0 0 0 0" if scan == "wg" else "%s -1 0" % scan))
                    j += 1
                else:
                    out.append(ln)
            ent = [ln for ln in lines2[1:] if ln.startswith(scan + " ")][k]
            run(lines2[0] + "\n" + "\n".join(out) + "\n", "only %s[%d] = %s" % (scan, k, ent))

if __name__ == "__main__":
    main()
