#!/usr/bin/env python3
"""Round-3 diagnostics of the exact-fp32 conv kernel (needs the -DWUN_ABLATION build: WUN_LIB=libwun_abl.so).

Per representative layer: (a) time vs resident workgroups per CU (WUN_LDS_PAD), (b) the phase ablations,
(c) the workgroup life-cycle trace (entry / first chunk staged / chunk loop done / end stamps, HW_ID, XCC_ID,
100 MHz reference clock) written to gpurun_out/diag/trace_<name>.npy for tools/diag_r3_report.py.
usage: WUN_LIB=libwun_abl.so python tools/diag_r3.py [outdir]"""
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from wave_u_net_amd import _lib

OUT = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/diag"
os.makedirs(OUT, exist_ok=True)
lib = _lib.load()
lib.wun_dbg_trace_read.restype = C.c_int
lib.wun_dbg_trace_read.argtypes = [C.c_void_p, C.c_int, C.c_int]
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
B = 16

# name, kind, Cin, Cout, K, T_in, stride, variant (kConvVariants index, -1 = heuristic)
LAYERS = [
    ("fwd_s2_72_96", "fwd", 72, 96, 15, 18421, 2, 18),
    ("fwd_s2_96_120", "fwd", 96, 120, 15, 9205, 2, 24),
    ("fwd_s2_144_168", "fwd", 144, 168, 15, 2293, 2, 18),
    ("fwd_s1_168_72_k5", "fwd", 168, 72, 5, 4108, 1, 18),
    ("dgrad_s2_96_120", "dgrad", 96, 120, 15, 9204, 2, -1),
]


def make(kind, cin, cout, k, t, stride):
    t_out = (t - k) // stride + 1
    x = torch.rand(B, cin, t, device="cuda") * 2 - 1
    w = (torch.rand(k, cin, cout, device="cuda") * 2 - 1) / (k * cin) ** 0.5
    b = torch.zeros(cout, device="cuda")
    y = torch.empty(B, cout, t_out, device="cuda")
    dz = torch.rand(B, cout, t_out, device="cuda") * 2 - 1
    if kind == "fwd":
        fn = lambda: lib.wun_op_conv1d(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), B, cin, cout, k, t,
                                       t_out, stride, 0, 1, st)
    else:
        wts = torch.empty(2 * (k + 1) * cin * cout + 64, device="cuda")
        dx = torch.empty(B, cin, t, device="cuda")
        fn = lambda: lib.wun_op_conv1d_dgrad(dz.data_ptr(), w.data_ptr(), dx.data_ptr(), wts.data_ptr(), B, cin,
                                             cout, k, t, t_out, stride, 0, st)
    keep = (x, w, b, y, dz)
    return fn, 2.0 * k * cin * cout * t_out * B, keep


def timed(fn, iters=10):
    for _ in range(2):
        _lib.check(fn())
    torch.cuda.synchronize()
    lib.wun_profile_begin()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    buf = C.create_string_buffer(1 << 20)
    _lib.check(lib.wun_profile_end(buf, len(buf)))
    ks = json.loads(buf.value.decode())["kernels"]
    ks = [k for k in ks if k["name"].startswith("conv_mfma")]
    ms = sum(k["ms"] for k in ks) / iters
    return ms, ",".join(k["name"][16:] for k in ks)


def setenv(**kw):
    for k, v in kw.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = str(v)


results = []
for name, kind, cin, cout, k, t, stride, variant in LAYERS:
    fn, flops, keep = make(kind, cin, cout, k, t, stride)
    lib.wun_op_force_conv_variant(variant, 1 if variant >= 0 else 0)
    setenv(WUN_ABLATE=None, WUN_LDS_PAD=None)
    base, kname = timed(fn)
    print("%-22s %-28s base %.1f us  %.1f TFLOP/s" % (name, kname, base * 1e3, flops / base / 1e9), flush=True)
    row = {"layer": name, "kernel": kname, "flops": flops, "base_ms": base}
    # (a) resident workgroups: pad the dynamic LDS request
    for pad in (16, 40, 90):
        setenv(WUN_LDS_PAD=pad)
        ms, _ = timed(fn)
        row["ldspad_%d" % pad] = ms
        print("    LDS +%2d KiB          : %.1f us  %.1f TFLOP/s" % (pad, ms * 1e3, flops / ms / 1e9), flush=True)
    setenv(WUN_LDS_PAD=None)
    # (b) phase ablations (bits: 1 noload, 2 nostore, 4 nomfma, 8 noepi, 16 nobar, 32 nolds)
    for abl, what in ((8, "no epilogue"), (3, "no global loads / LDS stores"), (11, "MFMA loop + barriers only"),
                      (27, "MFMA loop only"), (4, "no MFMA"), (12, "staging only (no MFMA, no epilogue)")):
        setenv(WUN_ABLATE=abl)
        ms, _ = timed(fn)
        row["abl_%d" % abl] = ms
        print("    ablate %2d %-34s: %.1f us  (%.1f TFLOP/s equiv)" % (abl, what, ms * 1e3, flops / ms / 1e9), flush=True)
    # (c) trace
    setenv(WUN_ABLATE=64)
    _lib.check(fn())
    torch.cuda.synchronize()
    lib.wun_dbg_trace_read(None, 0, 1)
    _lib.check(fn())
    torch.cuda.synchronize()
    host = np.zeros((16384, 16), dtype=np.uint64)
    n = lib.wun_dbg_trace_read(host.ctypes.data, 16384, 1)
    used = host[host[:, 0] != 0]
    np.save(os.path.join(OUT, "trace_%s.npy" % name), used)
    setenv(WUN_ABLATE=None)
    results.append(row)
    del keep
lib.wun_op_force_conv_variant(-1, 0)
json.dump(results, open(os.path.join(OUT, "diag.json"), "w"), indent=1)
print("done")
