#!/usr/bin/env python3
"""Per-chunk / per-workgroup cost decomposition of the conv kernel (needs the -DWUN_ABLATION build,
WUN_LIB=...): time vs number of input-channel chunks at a fixed grid; slope = cost per chunk,
intercept = fixed cost per workgroup round.
usage: ablate_sweep.py variant Cout K T_out stride"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wave_u_net_amd import _lib

variant, Cout, K, t_out, stride = [int(v) for v in sys.argv[1:6]]
lib = _lib.load()
B = 16
T = (t_out - 1) * stride + K
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
os.environ["WUN_VARIANT"] = str(variant)
for abl in (0, 8, 16, 27, 1 + 2 + 16, 4):
    os.environ["WUN_ABLATE"] = str(abl)
    res = []
    for Cin in (24, 48, 96, 192, 384):
        x = torch.rand(B, Cin, T, device="cuda")
        w = torch.rand(K, Cin, Cout, device="cuda") / (K * Cin)
        b = torch.zeros(Cout, device="cuda")
        y = torch.empty(B, Cout, t_out, device="cuda")
        fn = lambda: lib.wun_op_conv1d(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), B, Cin, Cout, K, T, t_out, stride, 0, 1, st)
        for _ in range(3): _lib.check(fn())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        res.append((Cin, e0.elapsed_time(e1) / 10))
    (c0, t0), (c1, t1) = res[1], res[-1]
    slope = (t1 - t0) / (c1 - c0)           # ms per input channel
    icpt = t0 - slope * c0
    flops_per_ch = 2.0 * K * Cout * t_out * B
    print("ablate %2d | " % abl + " ".join("C%d %.3f" % r for r in res) + " | slope %.1f TF  intercept %.3f ms" % (flops_per_ch / slope / 1e9, icpt))
