#!/usr/bin/env python3
"""Round-3 experiment: first-round stagger of co-resident workgroups and raised wave priority around the staging /
prologue / epilogue code of conv_mfma_kernel (ablation build knobs).  usage: WUN_LIB=libwun_abl.so python tools/diag_r3b.py"""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from wave_u_net_amd import _lib
import importlib.util
lib = _lib.load()
lib.wun_dbg_trace_read.restype = C.c_int; lib.wun_dbg_trace_read.argtypes = [C.c_void_p, C.c_int, C.c_int]
lib.wun_dbg_set_knob.restype = C.c_int; lib.wun_dbg_set_knob.argtypes = [C.c_int, C.c_int]
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
B = 16
LAYERS = [("fwd_s2_72_96", "fwd", 72, 96, 15, 18421, 2, 18, 82944), ("fwd_s2_96_120", "fwd", 96, 120, 15, 9205, 2, 24, 147456),
          ("fwd_s1_168_72_k5", "fwd", 168, 72, 5, 4108, 1, 18, 60480), ("dgrad_s2_96_120", "dgrad", 96, 120, 15, 9204, 2, -1, 92160)]

def make(kind, cin, cout, k, t, stride):
    t_out = (t - k) // stride + 1
    x = torch.rand(B, cin, t, device="cuda") * 2 - 1
    w = (torch.rand(k, cin, cout, device="cuda") * 2 - 1) / (k * cin) ** 0.5
    b = torch.zeros(cout, device="cuda"); y = torch.empty(B, cout, t_out, device="cuda")
    dz = torch.rand(B, cout, t_out, device="cuda") * 2 - 1
    if kind == "fwd":
        fn = lambda: lib.wun_op_conv1d(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), B, cin, cout, k, t, t_out, stride, 0, 1, st)
    else:
        wts = torch.empty(2 * (k + 1) * cin * cout + 64, device="cuda"); dx = torch.empty(B, cin, t, device="cuda")
        fn = lambda: lib.wun_op_conv1d_dgrad(dz.data_ptr(), w.data_ptr(), dx.data_ptr(), wts.data_ptr(), B, cin, cout, k, t, t_out, stride, 0, st)
    return fn, 2.0 * k * cin * cout * t_out * B, (x, w, b, y, dz)

def span(fn):
    os.environ["WUN_ABLATE"] = "64"
    fn(); torch.cuda.synchronize(); lib.wun_dbg_trace_read(None, 0, 1)
    sp = []
    for _ in range(3):
        fn(); torch.cuda.synchronize()
        host = np.zeros((16384, 16), dtype=np.uint64)
        lib.wun_dbg_trace_read(host.ctypes.data, 16384, 1)
        u = host[host[:, 0] != 0].astype(np.int64)
        sp.append((u[:, 6].max() - u[:, 5].min()) / 100.0)
    os.environ.pop("WUN_ABLATE")
    return min(sp)

def wall(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

for name, kind, cin, cout, k, t, stride, variant, mcyc in LAYERS:
    fn, flops, keep = make(kind, cin, cout, k, t, stride)
    lib.wun_op_force_conv_variant(variant, 1 if variant >= 0 else 0)
    for lag, prio in ((0, 0), (mcyc, 0), (mcyc // 2, 0), (0, 1), (mcyc, 1)):
        lib.wun_dbg_set_knob(0, lag); lib.wun_dbg_set_knob(1, prio)
        w = wall(fn); s = span(fn)
        print("%-18s lag %6d prio %d : back-to-back %.1f us (%.1f TFLOP/s)   trace span %.1f us" % (name, lag, prio, w, flops / w / 1e6, s), flush=True)
    lib.wun_dbg_set_knob(0, 0); lib.wun_dbg_set_knob(1, 0)
