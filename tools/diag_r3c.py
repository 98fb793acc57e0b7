#!/usr/bin/env python3
"""Round-3 experiment: is the conv kernel's non-MFMA time memory latency or instruction issue?  Ablation bits
128 (every workgroup loads excerpt 0 / tile 0: L2-resident input) and 256 (every workgroup stores to tile 0)."""
import ctypes as C, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from wave_u_net_amd import _lib
lib = _lib.load()
lib.wun_dbg_set_knob.restype = C.c_int; lib.wun_dbg_set_knob.argtypes = [C.c_int, C.c_int]
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
B = 16
LAYERS = [("fwd_s2_72_96", "fwd", 72, 96, 15, 18421, 2, 18), ("fwd_s2_96_120", "fwd", 96, 120, 15, 9205, 2, 24),
          ("fwd_s1_168_72_k5", "fwd", 168, 72, 5, 4108, 1, 18), ("dgrad_s2_96_120", "dgrad", 96, 120, 15, 9204, 2, -1)]

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

def wall(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

for name, kind, cin, cout, k, t, stride, variant in LAYERS:
    fn, flops, keep = make(kind, cin, cout, k, t, stride)
    lib.wun_op_force_conv_variant(variant, 1 if variant >= 0 else 0)
    for abl, what in ((0, "baseline"), (128, "hot loads"), (256, "hot stores"), (384, "hot loads+stores"), (8, "no epilogue"),
                      (8 + 128, "no epilogue, hot loads"), (27, "MFMA loop only"), (3, "no loads / LDS stores"), (2, "loads but no LDS stores"),
                      (1, "LDS stores but no loads")):
        os.environ["WUN_ABLATE"] = str(abl)
        w = wall(fn)
        print("%-18s ablate %3d %-26s: %.1f us (%.1f TFLOP/s equiv)" % (name, abl, what, w, flops / w / 1e6), flush=True)
    os.environ.pop("WUN_ABLATE")
