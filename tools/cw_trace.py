#!/usr/bin/env python3
"""Chunk timeline of conv_win_kernel (diagnostic build: make -C wave-u-net_amd/csrc cwtrace; WUN_LIB=libwun_cwtrace.so).
usage: WUN_LIB=libwun_cwtrace.so python tools/cw_trace.py Cin Cout K stride Tout pad variant"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from wave_u_net_amd import _lib

Cin, Cout, K, stride, Tout, pad, var = [int(v) for v in sys.argv[1:8]]
B = 16
lib = _lib.load()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
T = (Tout - 1) * stride + K - 2 * pad
x = torch.rand(B, Cin, T, device="cuda") * 2 - 1
w = (torch.rand(K, Cin, Cout, device="cuda") * 2 - 1) / (K * Cin) ** 0.5
bias = torch.zeros(Cout, device="cuda")
y = torch.empty(B, Cout, Tout, device="cuda")
lib.wun_op_force_conv_variant(var, 1)
for _ in range(4):
    _lib.check(lib.wun_op_conv1d(x.data_ptr(), w.data_ptr(), bias.data_ptr(), y.data_ptr(), B, Cin, Cout, K, T, Tout, stride, pad, 1, st))
torch.cuda.synchronize()
CH, WGS = 12, 2048
W = 6 + 3 * CH
buf = (C.c_ulonglong * (WGS * 4 * W))()
dll = C.CDLL(_lib.LIB_PATH)
dll.wun_dbg_cw_trace_read.argtypes = [C.c_void_p, C.c_int]
dll.wun_dbg_cw_trace_read(buf, WGS * 4 * W)
a = np.frombuffer(buf, dtype=np.uint64).reshape(WGS, 4, W).astype(np.int64)
a = a[a[:, 0, 0] > 0]
t0 = a[:, :, 0].min()
life = (a[:, 0, 2] - a[:, 0, 0]) / 100.0
st_us = (a[:, 0, 0] - t0) / 100.0
print("workgroups traced", a.shape[0], "chunks", int(np.median(a[:, 0, 3])))
print("kernel span %.1f us; WG start p50/p90/max %.1f %.1f %.1f; lifetime p10/p50/p90 %.1f %.1f %.1f us" % (
    (a[:, :, 2].max() - t0) / 100.0, np.median(st_us), np.percentile(st_us, 90), st_us.max(),
    np.percentile(life, 10), np.median(life), np.percentile(life, 90)))
c = a[:, :, 6:].reshape(a.shape[0], 4, CH, 3)
pro = np.median(c[:, 0, 0, 0] - a[:, 0, 1]); epi = np.median(a[:, 0, 5] - a[:, 0, 4])
print("wave 0: prologue %d cyc, epilogue %d cyc" % (pro, epi))
for wv in range(4):
    rows = []
    for k in range(min(CH, int(np.median(a[:, 0, 3])))):
        s = c[:, wv, k]
        ok = s[:, 2] > 0
        rows.append("c%d wait %5.0f mfma %6.0f" % (k, np.median(s[ok, 1] - s[ok, 0]), np.median(s[ok, 2] - s[ok, 1])))
    print("wave %d: " % wv + " | ".join(rows[:7]))
cyc = a[:, 0, 5] - a[:, 0, 1]
print("effective shader clock: %.2f GHz" % np.median(cyc / np.maximum(life, 1e-9) / 1e3))
