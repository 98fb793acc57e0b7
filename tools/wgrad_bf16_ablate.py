#!/usr/bin/env python3
"""Diagnostic (GPU box): time the bf16 weight-gradient operator on one layer shape with phases of the kernel switched
off (WUN_WGB_ABL bits: 1 no global loads, 2 no LDS stores, 4 no MFMA loop, 8 no barriers, 16 no write rotation;
needs a library built with make -C wave-u-net_amd/csrc EXTRA=-DWUN_BF_ABLATION).
usage: python tools/wgrad_bf16_ablate.py B Cin Cout K T stride [mtw nw nsplit]"""
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(shape, geom):
    import torch
    from wave_u_net_amd import _lib
    lib = _lib.load()
    B, Cin, Cout, K, T, stride = shape
    t_out = (T - K) // stride + 1
    x = torch.randn(B, Cin, T, device="cuda")
    dz = torch.randn(B, Cout, t_out, device="cuda")
    dw = torch.empty(K, Cin, Cout, device="cuda")
    db = torch.empty(Cout, device="cuda")
    lib.wun_op_set_wgrad_bf16(1)
    lib.wun_op_force_wgrad_variant(*geom)
    scr = torch.empty(int(lib.wun_op_conv1d_wgrad_scratch(B, Cin, Cout, K, t_out)), device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def call():
        _lib.check(lib.wun_op_conv1d_wgrad(x.data_ptr(), dz.data_ptr(), dw.data_ptr(), db.data_ptr(), scr.data_ptr(), B, Cin, Cout, K, T,
                                           t_out, stride, 0, st))
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    lib.wun_profile_begin()
    for _ in range(10):
        call()
    buf = C.create_string_buffer(1 << 16)
    lib.wun_profile_end(buf, len(buf))
    k = [k for k in json.loads(buf.value.decode())["kernels"] if "wgrad_bf16" in k["name"]][0]
    us = 1e3 * k["ms"] / k["launches"]
    print("ABL=%-2s %s: %8.1f us/launch  %7.1f TFLOP/s" % (os.environ.get("WUN_WGB_ABL", "0"), k["name"], us,
                                                          k["flops"] / k["launches"] / us / 1e6), flush=True)


if __name__ == "__main__":
    shape = [int(v) for v in sys.argv[1:7]]
    geom = [int(v) for v in sys.argv[7:10]] if len(sys.argv) >= 10 else [0, 0, 0]
    if os.environ.get("WUN_WGB_CHILD"):
        run(shape, geom)
    else:
        for abl in [int(v) for v in os.environ.get("WUN_WGB_LIST", "0,1,2,4,8,3,5,6,7,15").split(",")]:
            env = dict(os.environ, WUN_WGB_ABL=str(abl), WUN_WGB_CHILD="1")
            r = subprocess.run([sys.executable, __file__] + sys.argv[1:], env=env, capture_output=True, text=True)
            print((r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1], flush=True)
