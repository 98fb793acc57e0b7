#!/usr/bin/env python3
"""Op-level A/B of the weight-gradient kernels on the layer shapes of the headline configuration (B = 16):
the LDS-tiled wgrad_mfma_kernel (every geometry x split targets) against the register-window wgrad_win_kernel (target
grids 256 .. 2048), interleaved round-robin, kernel times from the library's HIP-event brackets (wgrad + its split
reduction).
usage: python tools/wgrad_win_bench.py [shape-filter] [rounds]"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wave_u_net_amd import _lib

SHAPES = [  # (Cin, Cout, K, stride, Tq)
    (24, 48, 15, 2, 36851), (48, 72, 15, 2, 18419), (72, 96, 15, 2, 9203), (96, 120, 15, 2, 4595),
    (120, 144, 15, 2, 2291), (144, 168, 15, 2, 1139), (168, 192, 15, 2, 563),
    (24, 48, 15, 1, 8201), (48, 72, 15, 1, 4105), (72, 96, 15, 1, 2057), (96, 120, 15, 1, 1033), (120, 144, 15, 1, 521),
    (72, 24, 5, 1, 16389), (120, 48, 5, 1, 8197), (168, 72, 5, 1, 4101), (216, 96, 5, 1, 2053), (264, 120, 5, 1, 1029),
    (312, 144, 5, 1, 517),
]
B = 16


def main():
    filt = sys.argv[1] if len(sys.argv) > 1 else ""
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    lib = _lib.load()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    out = []
    for (Cin, Cout, K, stride, Tq) in SHAPES:
        name = "C%d_N%d_K%d_s%d_T%d" % (Cin, Cout, K, stride, Tq)
        if filt and filt not in name:
            continue
        T = (Tq - 1) * stride + K
        x = torch.rand(B, Cin, T, device="cuda") * 2 - 1
        dz = torch.rand(B, Cout, Tq, device="cuda") * 2 - 1
        dw = torch.empty(K, Cin, Cout, device="cuda"); db = torch.empty(Cout, device="cuda")
        flops = 2.0 * K * Cin * Cout * Tq * B
        M = Cin * K + 1
        cands = [("old", 0, 0, 0, 0)]
        nws = sorted(set(nw for nw in range(1, 6) if ((Cout + 16 * nw - 1) // (16 * nw)) * 16 * nw <= 1.2 * Cout + 8))
        for win in (0,):
            for mtw in (6, 4, 2):
                if M <= 64 * (mtw // 2) and mtw > 2:
                    continue
                for nw in nws:
                    per = ((M + 64 * mtw - 1) // (64 * mtw)) * ((Cout + 16 * nw - 1) // (16 * nw))
                    for tgt in (512, 1024):
                        ns = max(1, tgt // per)
                        cands.append(("mf", mtw, nw, ns, win))
        if os.environ.get("ONLY_WIN"):
            cands = [c for c in cands if c[0] == "old" or (c[0] == "mf" and c[3] in (0,))][:1]
        tiles_guess = 1
        for tgt in (256, 512, 768, 1024, 1536, 2048):
            cands.append(("win", 0, 0, -tgt, 1))
        cands = list(dict.fromkeys(cands))
        res = {}
        ref = None
        live = []
        for c in cands:
            kind, mtw, nw, ns, win = c
            lib.wun_op_set_wgrad_win(win)
            lib.wun_op_force_wgrad_variant(mtw, nw, ns)
            if kind == "win":
                lib.wun_op_force_wgrad_variant(0, 0, -(1 << 14))
            n = lib.wun_op_conv1d_wgrad_scratch(B, Cin, Cout, K, Tq)
            lib.wun_op_force_wgrad_variant(mtw, nw, ns)
            scr = torch.empty(int(n), device="cuda")
            rc = lib.wun_op_conv1d_wgrad(x.data_ptr(), dz.data_ptr(), dw.data_ptr(), db.data_ptr(), scr.data_ptr(),
                                         B, Cin, Cout, K, T, Tq, stride, 0, st)
            torch.cuda.synchronize()
            if rc != 0:
                continue
            if ref is None:
                ref = (dw.clone(), db.clone())
                err = 0.0
            else:
                err = max(((dw - ref[0]).abs().max() / ref[0].abs().max()).item(),
                          ((db - ref[1]).abs().max() / ref[1].abs().max()).item())
            live.append((c, scr, err))
            res[c] = []
        for r in range(rounds):
            for (c, scr, err) in live:
                kind, mtw, nw, ns, win = c
                lib.wun_op_set_wgrad_win(win)
                lib.wun_op_force_wgrad_variant(mtw, nw, ns)
                lib.wun_profile_begin()
                for _ in range(3):
                    lib.wun_op_conv1d_wgrad(x.data_ptr(), dz.data_ptr(), dw.data_ptr(), db.data_ptr(), scr.data_ptr(),
                                            B, Cin, Cout, K, T, Tq, stride, 0, st)
                torch.cuda.synchronize()
                buf = C.create_string_buffer(1 << 20)
                _lib.check(lib.wun_profile_end(buf, len(buf)))
                pj = json.loads(buf.value.decode())
                ovh = pj.get("bracket_overhead_ms", 0.0)
                wg = sum(k["ms"] / k["launches"] - ovh for k in pj["kernels"] if k["name"].startswith("wgrad_mfma") or k["name"].startswith("wgrad_win_kernel"))
                rd = sum(k["ms"] / k["launches"] - ovh for k in pj["kernels"] if k["name"].startswith("wgrad_reduce") or k["name"].startswith("wgrad_win_reduce"))
                nm = [k["name"] for k in pj["kernels"] if k["name"].startswith("wgrad_mfma") or k["name"].startswith("wgrad_win_kernel")][0]
                res[c].append((wg, rd, nm))
        lib.wun_op_set_wgrad_win(0)
        lib.wun_op_force_wgrad_variant(0, 0, 0)
        rows = []
        for (c, scr, err) in live:
            wg = min(t[0] for t in res[c]); rd = min(t[1] for t in res[c])
            rows.append((wg + rd, wg, rd, c, res[c][0][2], err))
        rows.sort()
        base = [r for r in rows if r[3][0] == "old"][0]
        print("== %s  %.2f GFLOP  old: %s %.1f us (+%.1f reduce) = %.1f TF" % (name, flops / 1e9, base[4], base[1] * 1e3, base[2] * 1e3, flops / base[1] / 1e9))
        for tot, wg, rd, c, nm, err in rows[:8]:
            print("   %-28s ns=%-4d wg %7.1f us  red %5.1f us  tot %7.1f  %6.1f TF  (x%.2f vs old) err %.1e" % (
                nm, c[3], wg * 1e3, rd * 1e3, tot * 1e3, flops / wg / 1e9, base[0] / tot, err))
        bestwin = [r for r in rows if r[3][0] == "win"]
        bestmf = [r for r in rows if r[3][0] in ("mf", "old")]
        out.append({"shape": name, "old_us": base[0] * 1e3, "best_mf_us": bestmf[0][0] * 1e3 if bestmf else None,
                    "best_win_us": bestwin[0][0] * 1e3 if bestwin else None,
                    "best_win": bestwin[0][3] if bestwin else None, "max_err_win": max([r[5] for r in bestwin] or [0])})
        sys.stdout.flush()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
