#!/usr/bin/env python3
"""Which workspace tensor does a tuned launch choice change beyond rounding?  M1 + context, B = 2, full length: one
forward + backward with the heuristic plan, one with a tuning table in which only ONE entry is active (category,
index, variant, ksplit -- e.g. `cf 3 9 1`), then every activation / gradient buffer of the workspace is compared on
its valid positions (the map comes from the library's WUN_DUMP_LAYOUT=1 lines).
usage: python tools/ws_diff.py cf 3 9 1"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["WUN_DUMP_LAYOUT"] = "1"
LAY = "/tmp/ws_layout.txt"
_saved = os.dup(2)
_f = open(LAY, "w")
os.dup2(_f.fileno(), 2)                       # the library prints its layout lines to stderr
import numpy as np, torch
import wave_u_net_amd as wun
from wave_u_net_amd import UnetAudioSeparator
from oracle import shapes, waveunet_torch as wt
from oracle.golden_params import golden_params


def main():
    cat, idx, var, ks = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    ocfg = shapes.finalize_config(dict(shapes.BASE_MODEL_CONFIG, context=True))
    params = golden_params(ocfg, 77)
    B = 2
    i, o = shapes.get_padding(ocfg, [B, 16384, 0])
    mix, targets = wt.synthetic_batch(ocfg, B, i[1], o[1], seed=78)
    tg = {k: torch.from_numpy(v) for k, v in targets.items()}
    dmix = torch.from_numpy(mix).cuda()

    def run(table):
        sep = UnetAudioSeparator(wun.get_config("m1_context"), device="cuda:0")
        sep._plan(B, i[1]); sep._active = sep._plans[(B, i[1])]
        sep.load_variables(params)
        sep.get_output(dmix, True)
        if table == "export":
            sep.tune(dmix, tg)
            return sep.tune_export(), None, None
        if table is not None:
            sep.tune_import(table)
        sep.get_output(dmix, True)
        sep.loss_and_gradients(tg)
        torch.cuda.synchronize()
        return None, sep._ws[sep._last_key].clone().cpu().numpy(), sep.grads.clone().cpu().numpy()

    text, _, _ = run("export")
    lines = text.strip().split("\n")
    out, k = [], {"cf": 0, "cb": 0, "wg": 0}
    for ln in lines[1:]:
        c = ln.split()[0]
        if c in k:
            if c == cat and k[c] == idx:
                out.append("%s %d %d" % (c, var, ks))
            else:
                out.append("wg 0 0 0 0" if c == "wg" else "%s -1 0" % c)
            k[c] += 1
        else:
            out.append(ln)
    table = lines[0] + "\n" + "\n".join(out) + "\n"
    _, ws0, g0 = run(None)
    _, ws1, g1 = run(table)
    os.dup2(_saved, 2)
    _f.close()
    print("gradient arena: max |diff| / max|g| = %.3e" % (np.abs(g1 - g0).max() / np.abs(g0).max()))
    seen = set()
    rows = []
    for ln in open(LAY):
        if not ln.startswith("[wun-layout]"):
            continue
        t = ln.split()
        name, ix = t[1], int(t[2])
        kv = dict(x.split("=") for x in t[3:])
        if (name, ix) in seen:
            continue
        seen.add((name, ix))
        off, Bn, C, T, pitch = int(kv["off"]), int(kv["B"]), int(kv["C"]), int(kv["T"]), int(kv["pitch"])
        if int(kv.get("eb", 4)) == 2:                  # bf16 mode: the tensor holds bfloat16 elements (off is still in floats)
            def as_f32(ws):
                u16 = ws[off:off + (Bn * C * pitch + 1) // 2].view(np.uint16)[:Bn * C * pitch]
                return (u16.astype(np.uint32) << 16).view(np.float32).reshape(Bn, C, pitch)[:, :, :T]
            a0, a1 = as_f32(ws0), as_f32(ws1)
        else:
            a0 = ws0[off:off + Bn * C * pitch].reshape(Bn, C, pitch)[:, :, :T]
            a1 = ws1[off:off + Bn * C * pitch].reshape(Bn, C, pitch)[:, :, :T]
        d = np.abs(a1 - a0)
        sc = max(np.abs(a0).max(), 1e-30)
        rel = d.max() / sc
        nbad = int((d > 1e-4 * sc).sum())
        w = np.unravel_index(np.argmax(d), d.shape)
        rows.append((off, name, ix, rel, nbad, w, a0[w], a1[w], (C, T)))
    rows.sort()
    for off, name, ix, rel, nbad, w, v0, v1, ct in rows:
        flag = "  <<<" if rel > 1e-5 else ""
        print("%-8s %2d C,T=%-12s max|diff|/max %.2e  elements > 1e-4: %-6d worst at (b,c,t)=%s  %.6g -> %.6g%s" % (name, ix, ct, rel, nbad, w, v0, v1, flag))


if __name__ == "__main__":
    main()
