#!/usr/bin/env python3
"""Kernel-only wgrad time vs excerpt length at a fixed geometry (needs the -DWUN_ABLATION build for
WUN_ABLATE != 0): slope = steady-state rate, intercept = fixed cost per launch.
usage: wgrad_sweep.py Cin Cout K stride"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wave_u_net_amd import _lib

Cin, Cout, K, stride = [int(v) for v in sys.argv[1:5]]
lib = _lib.load()
B = 16
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for abl in (0, 8, 3, 11):
    os.environ["WUN_ABLATE"] = str(abl)
    res = []
    for t_out in (2048, 4096, 8192, 16384):
        T = (t_out - 1) * stride + K
        x = torch.rand(B, Cin, T, device="cuda")
        dz = torch.rand(B, Cout, t_out, device="cuda")
        n = lib.wun_op_conv1d_wgrad_scratch(B, Cin, Cout, K, t_out)
        scr = torch.empty(int(n), device="cuda")
        dw = torch.empty(K, Cin, Cout, device="cuda"); db = torch.empty(Cout, device="cuda")
        fn = lambda: lib.wun_op_conv1d_wgrad(x.data_ptr(), dz.data_ptr(), dw.data_ptr(), db.data_ptr(), scr.data_ptr(),
                                             B, Cin, Cout, K, T, t_out, stride, 0, st)
        for _ in range(2): _lib.check(fn())
        torch.cuda.synchronize()
        lib.wun_profile_begin()
        for _ in range(5): fn()
        torch.cuda.synchronize()
        buf = C.create_string_buffer(1 << 20)
        _lib.check(lib.wun_profile_end(buf, len(buf)))
        kk = [k for k in json.loads(buf.value.decode())["kernels"] if k["name"].startswith("wgrad")][0]
        res.append((t_out, kk["ms"] / kk["launches"], kk["name"]))
    (t0, m0, _), (t1, m1, nm) = res[1], res[-1]
    slope = (m1 - m0) / (t1 - t0)
    fl = 2.0 * K * Cin * Cout * B
    print("%s ablate %2d | " % (nm, abl) + " ".join("T%d %.3f" % r[:2] for r in res) + " | slope %.1f TF intercept %.3f ms" % (fl / slope / 1e9, m0 - slope * t0))
