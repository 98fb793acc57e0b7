#!/usr/bin/env python3
"""Micro-benchmark of single operators through the C ABI (wun_op_conv1d / _wgrad / _dgrad).
usage: python tools/op_bench.py fwd|wgrad|dgrad B Cin Cout K T_in stride [iters]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from wave_u_net_amd import _lib


def main():
    kind = sys.argv[1]
    B, Cin, Cout, K, T, stride = [int(x) for x in sys.argv[2:8]]
    iters = int(sys.argv[8]) if len(sys.argv) > 8 else 20
    lib = _lib.load()
    t_out = (T - K) // stride + 1
    x = torch.rand(B, Cin, T, device="cuda") * 2 - 1
    w = (torch.rand(K, Cin, Cout, device="cuda") * 2 - 1) / (K * Cin) ** 0.5
    b = torch.zeros(Cout, device="cuda")
    y = torch.empty(B, Cout, t_out, device="cuda")
    dz = torch.rand(B, Cout, t_out, device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    if kind == "fwd":
        fn = lambda: lib.wun_op_conv1d(x.data_ptr(), w.data_ptr(), b.data_ptr(), y.data_ptr(), B, Cin, Cout, K, T,
                                       t_out, stride, 0, 1, st)
    elif kind == "wgrad":
        n = lib.wun_op_conv1d_wgrad_scratch(B, Cin, Cout, K, t_out)
        scr = torch.empty(int(n), device="cuda")
        dw = torch.empty(K, Cin, Cout, device="cuda"); db = torch.empty(Cout, device="cuda")
        fn = lambda: lib.wun_op_conv1d_wgrad(x.data_ptr(), dz.data_ptr(), dw.data_ptr(), db.data_ptr(),
                                             scr.data_ptr(), B, Cin, Cout, K, T, t_out, stride, 0, st)
    else:
        wts = torch.empty(2 * K * Cin * Cout, device="cuda")
        dx = torch.empty(B, Cin, T, device="cuda")
        fn = lambda: lib.wun_op_conv1d_dgrad(dz.data_ptr(), w.data_ptr(), dx.data_ptr(), wts.data_ptr(), B, Cin,
                                             Cout, K, T, t_out, stride, 0, st)
    for _ in range(3):
        _lib.check(fn())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    flops = 2.0 * K * Cin * Cout * t_out * B
    # kernel-only times (HIP events around each heavy launch inside the library)
    import json
    lib.wun_profile_begin()
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    buf = C.create_string_buffer(1 << 20)
    _lib.check(lib.wun_profile_end(buf, len(buf)))
    for kk in json.loads(buf.value.decode())["kernels"]:
        if kk["ms"] > 0:
            print("    %-44s %.3f ms/launch  %.1f TFLOP/s" % (kk["name"], kk["ms"] / kk["launches"],
                                                              kk["flops"] / kk["ms"] / 1e9))
    print("%s B%d %d->%d K%d T%d s%d | env ABLATE=%s VARIANT=%s NOVEC=%s : %.3f ms  %.1f TFLOP/s" % (
        kind, B, Cin, Cout, K, T, stride, "-", "-",
        "-", ms, flops / ms / 1e9))


if __name__ == "__main__":
    main()
