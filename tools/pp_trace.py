#!/usr/bin/env python3
"""Segment timeline of wgrad_pp_kernel (diagnostic build: make -C wave-u-net_amd/csrc pptrace; WUN_LIB=libwun_pptrace.so).
usage: WUN_LIB=... python tools/pp_trace.py Cin Cout K stride Tq mtw nw nsplit"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from wave_u_net_amd import _lib

Cin, Cout, K, stride, Tq, mtw, nw, ns = [int(v) for v in sys.argv[1:9]]
B = 16
lib = _lib.load()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
T = (Tq - 1) * stride + K
x = torch.rand(B, Cin, T, device="cuda") * 2 - 1
dz = torch.rand(B, Cout, Tq, device="cuda") * 2 - 1
dw = torch.empty(K, Cin, Cout, device="cuda"); db = torch.empty(Cout, device="cuda")
lib.wun_op_set_wgrad_pp(1)
lib.wun_op_force_wgrad_variant(mtw, nw, ns)
n = lib.wun_op_conv1d_wgrad_scratch(B, Cin, Cout, K, Tq)
scr = torch.empty(int(n), device="cuda")
for _ in range(4):
    _lib.check(lib.wun_op_conv1d_wgrad(x.data_ptr(), dz.data_ptr(), dw.data_ptr(), db.data_ptr(), scr.data_ptr(), B, Cin, Cout, K, T, Tq, stride, 0, st))
torch.cuda.synchronize()
SEGS, WGS = 20, 1024
W = 4 + 3 * SEGS
buf = (C.c_ulonglong * (WGS * 2 * W))()
dll = C.CDLL(_lib.LIB_PATH)
dll.wun_dbg_pp_trace_read.argtypes = [C.c_void_p, C.c_int]
nr = dll.wun_dbg_pp_trace_read(buf, WGS * 2 * W)
a = np.frombuffer(buf, dtype=np.uint64).reshape(WGS, 2, W).astype(np.int64)
live = a[:, 0, 3] > 0
a = a[live]
print("workgroups traced", a.shape[0], "units/WG median", np.median(a[:, 0, 3]))
wall = (a[:, 0, 2] - a[:, 0, 0]) / 100.0      # us (100 MHz)
t0 = a[:, 0, 0].min()
print("kernel span (first entry -> last exit): %.1f us; WG lifetime median %.1f us" % ((a[:, :, 2].max() - t0) / 100.0, np.median(wall)))
st0 = a[:, :, 4:].reshape(a.shape[0], 2, SEGS, 3)
for setn in (0, 1):
    print("set", setn)
    for k in range(min(SEGS, int(np.median(a[:, 0, 3])) + 1)):
        s = st0[:, setn, k]
        ok = s[:, 2] > 0
        if not ok.any(): continue
        work = np.median(s[ok, 1] - s[ok, 0]); wait = np.median(s[ok, 2] - s[ok, 1])
        role = "stage" if (k & 1) == setn else "mfma"
        print("  seg %2d %-5s work %7.0f cyc  barrier-wait %7.0f cyc" % (k, role, work, wait))
# effective clock: cycles over the lifetime vs 100 MHz ticks
cyc = st0[:, 0, :, 2].max(axis=1) - a[:, 0, 1]
print("effective shader clock: %.2f GHz" % np.median(cyc / np.maximum(wall, 1e-9) / 1e3))
