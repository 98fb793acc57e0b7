#!/usr/bin/env python3
"""Round-3 experiment: DMA staging vs register staging per representative stride-1 launch (single-operator entry
points, aligned rows), back-to-back wall time.  usage: python tools/diag_r3e.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wave_u_net_amd import _lib
lib = _lib.load()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
B = 16
ENVNAME = sys.argv[1] if len(sys.argv) > 1 else "WUN_NO_DMA"
LAYERS = [("down3 s2 72->96 K15", "fwd", 72, 96, 15, 18421, 2, 18), ("down4 s2 96->120 K15", "fwd", 96, 120, 15, 9205, 2, 24), ("down1 s2 24->48 K15", "fwd", 24, 48, 15, 73717, 2, 18), ("down6 s2 144->168", "fwd", 144, 168, 15, 2293, 2, 18),
          ("up9   168->72  K5  T4104", "fwd", 168, 72, 5, 4108, 1, 18), ("up7   264->120 K5  T1028", "fwd", 264, 120, 5, 1032, 1, 17),
          ("win3   72->96  K15 T2062", "fwd", 72, 96, 15, 2076, 1, 18), ("win2   48->72  K15 T4106", "fwd", 48, 72, 15, 4120, 1, 18),
          ("dgrad2 s2 96<-120 K15", "dgrad", 96, 120, 15, 9205, 2, -1), ("dgrad2 s2 72<-96 K15", "dgrad", 72, 96, 15, 18421, 2, -1),
          ("dgrad1 up 48+72<-48... K5", "dgrad", 120, 48, 5, 8200, 1, -1)]

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
    for _ in range(3): _lib.check(fn())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3

for name, kind, cin, cout, k, t, stride, variant in LAYERS:
    fn, flops, keep = make(kind, cin, cout, k, t, stride)
    lib.wun_op_force_conv_variant(variant, 1 if variant >= 0 else 0)
    res = []
    for rep in range(2):
        for dma in (1, 0):
            if dma: os.environ.pop(ENVNAME, None)
            else: os.environ[ENVNAME] = "1"
            res.append((dma, wall(fn)))
    d = min(w for m, w in res if m == 1); r = min(w for m, w in res if m == 0)
    print("%-28s DMA %.1f us (%.1f TFLOP/s)   registers %.1f us (%.1f TFLOP/s)   %+.1f %%" % (name, d, flops / d / 1e6, r, flops / r / 1e6, 100 * (d / r - 1)), flush=True)
os.environ.pop("WUN_NO_DMA", None)
