#!/usr/bin/env python3
"""Round-3 experiment: workgroup phase durations (trace) under ablations.  usage: WUN_LIB=libwun_abl.so python tools/diag_r3d.py"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from wave_u_net_amd import _lib
lib = _lib.load()
lib.wun_dbg_trace_read.restype = C.c_int; lib.wun_dbg_trace_read.argtypes = [C.c_void_p, C.c_int, C.c_int]
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
B = 16
LAYERS = [("down3_s2_72_96", "fwd", 72, 96, 15, 18420, 2, 18), ("down4_s2_96_120", "fwd", 96, 120, 15, 9204, 2, 24)]

def make(kind, cin, cout, k, t, stride):
    t_out = (t - k) // stride + 1
    x = torch.rand(B, cin, t, device="cuda") * 2 - 1
    w = (torch.rand(k, cin, cout, device="cuda") * 2 - 1) / (k * cin) ** 0.5
    t_y = (t_out + 3) // 4 * 4
    b = torch.zeros(cout, device="cuda"); y = torch.empty(B, cout, t_y, device="cuda")
    dz = torch.rand(B, cout, t_out, device="cuda") * 2 - 1
    if kind == "fwd":
        fn = lambda: lib.wun_op_conv1d_ex(x.data_ptr(), cin, None, 0, w.data_ptr(), b.data_ptr(), y.data_ptr(), None, B, cout, k, t, t_out, t_y, stride, 0, 1, 0, 1, 0, st)
    else:
        wts = torch.empty(2 * (k + 1) * cin * cout + 64, device="cuda"); dx = torch.empty(B, cin, t, device="cuda")
        fn = lambda: lib.wun_op_conv1d_dgrad(dz.data_ptr(), w.data_ptr(), dx.data_ptr(), wts.data_ptr(), B, cin, cout, k, t, t_out, stride, 0, st)
    return fn, 2.0 * k * cin * cout * t_out * B, (x, w, b, y, dz)

def trace(fn, abl):
    os.environ["WUN_ABLATE"] = str(64 + abl)
    fn(); torch.cuda.synchronize(); lib.wun_dbg_trace_read(None, 0, 1)
    fn(); torch.cuda.synchronize()
    host = np.zeros((16384, 16), dtype=np.uint64)
    lib.wun_dbg_trace_read(host.ctypes.data, 16384, 1)
    os.environ.pop("WUN_ABLATE")
    return host[host[:, 0] != 0].astype(np.int64)

for name, kind, cin, cout, k, t, stride, variant in LAYERS:
    fn, flops, keep = make(kind, cin, cout, k, t, stride)
    lib.wun_op_force_conv_variant(variant, 1 if variant >= 0 else 0)
    for abl, what in ((0, "baseline"), (8, "no epilogue"), (1, "no DMA / loads"), (4, "no MFMA")):
        u = trace(fn, abl)
        mhz = np.median((u[:, 3] - u[:, 0]) / np.maximum(1, (u[:, 6] - u[:, 5]))) * 100
        f = lambda a, b_: np.median(u[:, a] - u[:, b_]) / mhz
        span = (u[:, 6].max() - u[:, 5].min()) / 100.0
        # second-round workgroups only (first round = blockIdx < 768 is in lockstep)
        print("%-16s %-12s span %6.1f us | WG median: prologue %5.2f loop %6.2f epilogue %5.2f total %6.2f | clock %4.0f MHz" % (
            name, what, span, f(1, 0), f(2, 1), f(3, 2), f(3, 0), mhz), flush=True)
