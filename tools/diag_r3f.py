#!/usr/bin/env python3
"""Round-3: phase ablations of the exact-fp32 weight-gradient kernel on representative layers (ablation build):
bits 1 no global loads, 2 no LDS stores, 4 no MFMA, 8 no partial store.  Back-to-back wall time of kernel + reduction."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wave_u_net_amd import _lib
lib = _lib.load()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
B = 16
for name, cin, cout, k, t_out, stride in (("down3 dec 72->96 s2", 72, 96, 15, 9203, 2), ("down4 dec 96->120 s2", 96, 120, 15, 4595, 2),
                                         ("down2 win 48->72 s1", 48, 72, 15, 4105, 1), ("up9 168->72 K5", 168, 72, 5, 4101, 1)):
    T = (t_out - 1) * stride + k
    Tp = (T + 3) // 4 * 4
    x = torch.rand(B, cin, T, device="cuda") * 2 - 1
    dz = torch.rand(B, cout, t_out, device="cuda") * 2 - 1
    n = lib.wun_op_conv1d_wgrad_scratch(B, cin, cout, k, t_out)
    scr = torch.empty(int(n), device="cuda")
    dw = torch.empty(k, cin, cout, device="cuda"); db = torch.empty(cout, device="cuda")
    fn = lambda: lib.wun_op_conv1d_wgrad(x.data_ptr(), dz.data_ptr(), dw.data_ptr(), db.data_ptr(), scr.data_ptr(), B, cin, cout, k, T, t_out, stride, 0, st)
    flops = 2.0 * k * cin * cout * t_out * B
    for abl, what in ((0, "full"), (8, "no partial store"), (3, "no loads / LDS stores"), (11, "MFMA + barriers"), (4, "no MFMA"), (2, "loads, no LDS stores"), (1, "LDS stores, no loads")):
        os.environ["WUN_ABLATE"] = str(abl)
        for _ in range(3): _lib.check(fn())
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        print("%-22s ablate %2d %-24s %.1f us (%.1f TFLOP/s equiv)" % (name, abl, what, us, flops / us / 1e6), flush=True)
    os.environ.pop("WUN_ABLATE")
