#!/usr/bin/env python3
"""Op-level A/B of the conv kernels on the layer shapes of the headline configuration (B = 16): conv_mfma_kernel
(heuristic tile and its near alternatives) against the register-window tiles of conv_win_kernel (variants >= 42),
interleaved round-robin, kernel times from the library's HIP-event brackets; outputs compared with the heuristic launch.
usage: python tools/conv_win_bench.py [shape-filter] [rounds]"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wave_u_net_amd import _lib

SHAPES = [  # (Cin, Cout, K, stride, Tout, pad_left)
    (24, 48, 15, 2, 36851, 0), (48, 72, 15, 2, 18419, 0), (72, 96, 15, 2, 9203, 0), (96, 120, 15, 2, 4595, 0),
    (120, 144, 15, 2, 2291, 0), (144, 168, 15, 2, 1139, 0),
    (24, 48, 15, 1, 8201, 0), (48, 72, 15, 1, 4105, 0), (72, 96, 15, 1, 2057, 0), (96, 120, 15, 1, 1033, 0),
    (72, 24, 5, 1, 16389, 0), (120, 48, 5, 1, 8197, 0), (168, 72, 5, 1, 4101, 0), (216, 96, 5, 1, 2053, 0),
    (264, 120, 5, 1, 1029, 0), (312, 144, 5, 1, 517, 0),
    # input-gradient shapes (full correlation: pad K - 1 on both sides)
    (24, 72, 5, 1, 16393, 4), (48, 120, 5, 1, 8201, 4), (72, 168, 5, 1, 4105, 4), (96, 216, 5, 1, 2057, 4),
    (72, 48, 15, 1, 4119, 14), (96, 72, 15, 1, 2071, 14), (120, 96, 15, 1, 1047, 14),
]
B = 16
FIRST_WIN = 42


def main():
    filt = sys.argv[1] if len(sys.argv) > 1 else ""
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    lib = _lib.load()
    nvar = lib.wun_op_num_conv_variants()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    out = []
    for (Cin, Cout, K, stride, Tout, pad) in SHAPES:
        Tout = (Tout + 3) // 4 * 4          # (the op entry takes dense rows: 16-byte row pitch for the vector epilogues)
        name = "C%d_N%d_K%d_s%d_T%d_p%d" % (Cin, Cout, K, stride, Tout, pad)
        if filt and filt not in name:
            continue
        T = (Tout - 1) * stride + K - 2 * pad
        x = torch.rand(B, Cin, T, device="cuda") * 2 - 1
        w = (torch.rand(K, Cin, Cout, device="cuda") * 2 - 1) / (K * Cin) ** 0.5
        bias = torch.rand(Cout, device="cuda") - 0.5
        y = torch.empty(B, Cout, Tout, device="cuda")
        flops = 2.0 * K * Cin * Cout * Tout * B
        fn = lambda: lib.wun_op_conv1d(x.data_ptr(), w.data_ptr(), bias.data_ptr(), y.data_ptr(), B, Cin, Cout, K, T, Tout, stride, pad, 1, st)
        cands = [(-1, 0)] + [(v, 1) for v in range(nvar)]
        live, ref = [], None
        for c in cands:
            lib.wun_op_force_conv_variant(c[0], c[1])
            y.zero_()
            rc = fn()
            torch.cuda.synchronize()
            if rc != 0:
                continue
            if ref is None:
                ref = y.clone(); err = 0.0
            else:
                err = ((y - ref).abs().max() / ref.abs().max()).item()
            live.append((c, err))
        res = {c: [] for c, _ in live}
        for r in range(rounds):
            for c, _ in live:
                lib.wun_op_force_conv_variant(c[0], c[1])
                lib.wun_profile_begin()
                for _ in range(3):
                    fn()
                torch.cuda.synchronize()
                buf = C.create_string_buffer(1 << 20)
                _lib.check(lib.wun_profile_end(buf, len(buf)))
                pj = json.loads(buf.value.decode())
                ovh = pj.get("bracket_overhead_ms", 0.0)
                ks = [k for k in pj["kernels"] if k["name"].startswith("conv_mfma") or k["name"].startswith("conv_win_kernel")]
                res[c].append((sum(k["ms"] / k["launches"] - ovh for k in ks), ks[0]["name"]))
        lib.wun_op_force_conv_variant(-1, 0)
        rows = sorted((min(t[0] for t in res[c]), c, res[c][0][1], err) for c, err in live)
        base = [r for r in rows if r[1][0] == -1][0]
        print("== %s  %.2f GFLOP  heuristic: %s %.1f us = %.1f TF" % (name, flops / 1e9, base[2], base[0] * 1e3, flops / base[0] / 1e9))
        for ms, c, nm, err in rows[:6]:
            print("   v=%-3d %-44s %7.1f us  %6.1f TF  (x%.2f) err %.1e" % (c[0], nm, ms * 1e3, flops / ms / 1e9, base[0] / ms, err))
        win = [r for r in rows if r[1][0] >= FIRST_WIN]
        old = [r for r in rows if r[1][0] < FIRST_WIN]
        out.append({"shape": name, "heuristic_us": base[0] * 1e3, "best_old_us": old[0][0] * 1e3, "best_old": old[0][1][0],
                    "best_win_us": win[0][0] * 1e3 if win else None, "best_win": win[0][1][0] if win else None,
                    "max_err_win": max([r[3] for r in win] or [0.0])})
        sys.stdout.flush()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
