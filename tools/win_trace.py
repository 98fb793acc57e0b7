#!/usr/bin/env python3
"""Unit timeline of wgrad_win_kernel (diagnostic build: make -C wave-u-net_amd/csrc wintrace; WUN_LIB=libwun_wintrace.so).
usage: WUN_LIB=libwun_wintrace.so python tools/win_trace.py Cin Cout K stride Tq grid_target"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from wave_u_net_amd import _lib

Cin, Cout, K, stride, Tq, tgt = [int(v) for v in sys.argv[1:7]]
B = 16
lib = _lib.load()
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
T = (Tq - 1) * stride + K
x = torch.rand(B, Cin, T, device="cuda") * 2 - 1
dz = torch.rand(B, Cout, Tq, device="cuda") * 2 - 1
dw = torch.empty(K, Cin, Cout, device="cuda"); db = torch.empty(Cout, device="cuda")
lib.wun_op_set_wgrad_win(1)
lib.wun_op_force_wgrad_variant(0, 0, -tgt)
n = lib.wun_op_conv1d_wgrad_scratch(B, Cin, Cout, K, Tq)
scr = torch.empty(int(n), device="cuda")
for _ in range(4):
    _lib.check(lib.wun_op_conv1d_wgrad(x.data_ptr(), dz.data_ptr(), dw.data_ptr(), db.data_ptr(), scr.data_ptr(), B, Cin, Cout, K, T, Tq, stride, 0, st))
torch.cuda.synchronize()
UN, WGS = 12, 1024
W = 4 + 4 * UN
buf = (C.c_ulonglong * (WGS * 8 * W))()
dll = C.CDLL(_lib.LIB_PATH)
dll.wun_dbg_win_trace_read.argtypes = [C.c_void_p, C.c_int]
dll.wun_dbg_win_trace_read(buf, WGS * 8 * W)
a = np.frombuffer(buf, dtype=np.uint64).reshape(WGS, 8, W).astype(np.int64)
live = a[:, 0, 0] > 0
a = a[live]
nu = (a[:, 0, 3] >> 32)
hw = a[:, :, 3] & 0xFFFFFFFF
print("workgroups traced", a.shape[0], "units/WG median", np.median(nu), "waves/WG", int((a[0, :, 0] > 0).sum()))
t0 = a[:, :, 0][a[:, :, 0] > 0].min()
print("kernel span %.1f us; WG lifetime median %.1f us" % ((a[:, :, 2].max() - t0) / 100.0, np.median((a[:, 0, 2] - a[:, 0, 0]) / 100.0)))
st_us = (a[:, 0, 0] - t0) / 100.0; life = (a[:, 0, 2] - a[:, 0, 0]) / 100.0
print("WG start (us) p0/p50/p90/p100: %.1f %.1f %.1f %.1f | lifetime p0/p50/p90/p100: %.1f %.1f %.1f %.1f" % (
    st_us.min(), np.median(st_us), np.percentile(st_us, 90), st_us.max(), life.min(), np.median(life), np.percentile(life, 90), life.max()))
# SIMD placement: HW_ID bits: wave_id[3:0], simd_id[5:4], pipe[7:6], cu_id[11:8], sh[12], se[15:13]...
simd = (hw >> 4) & 3
cu = ((hw >> 8) & 0xF) | (((hw >> 13) & 0x7) << 4) | (((hw >> 12) & 1) << 7)
wl = a[:, :, 0] > 0
print("SIMD of waves 0..7 of the first 6 WGs:", [list(simd[i][wl[i]]) for i in range(min(6, a.shape[0]))])
st4 = a[:, :, 4:].reshape(a.shape[0], 8, UN, 4)
for w in range(8):
    if not wl[:, w].any(): continue
    rows = []
    for k in range(min(UN, int(np.median(nu)))):
        s = st4[wl[:, w], w, k]
        ok = s[:, 3] > 0
        if not ok.any(): continue
        rows.append((k, np.median(s[ok, 1] - s[ok, 0]), np.median(s[ok, 2] - s[ok, 1]), np.median(s[ok, 3] - s[ok, 2])))
    print("wave %d: " % w + " | ".join("u%d wait %5.0f dma %4.0f mfma %6.0f" % r for r in rows[:6]))
cyc = st4[:, 0, :, 3].max(axis=1) - a[:, 0, 1]
wall = (a[:, 0, 2] - a[:, 0, 0]) / 100.0
print("effective shader clock (approx): %.2f GHz" % np.median(cyc / np.maximum(wall, 1e-9) / 1e3))
