#!/usr/bin/env python3
"""Diagnostic (GPU box): time the bf16 conv operator on one layer shape with phases of the kernel switched off
(WUN_BF_ABL bits: 1 no MFMA, 2 no epilogue, 4 no input loads, 8 no weight DMA; needs a library built with
make -C wave-u-net_amd/csrc EXTRA=-DWUN_BF_ABLATION).
usage: python tools/bf16_ablate.py B Cin Cout K T stride"""
import ctypes as C
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(shape):
    import torch
    from wave_u_net_amd import _lib
    lib = _lib.load()
    B, Cin, Cout, K, T, stride = shape
    t_out = (T - K) // stride + 1
    x = torch.randn(B, Cin, T, device="cuda")
    w = torch.randn(K, Cin, Cout, device="cuda") * 0.05
    bias = torch.zeros(Cout, device="cuda")
    y = torch.empty(B, Cout, t_out, device="cuda")
    scr = torch.empty(int(lib.wun_op_conv1d_bf16_scratch(Cin, Cout, K)), device="cuda")
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    def call():
        lib.wun_op_conv1d_bf16(x.data_ptr(), w.data_ptr(), bias.data_ptr(), y.data_ptr(), scr.data_ptr(), B, Cin, Cout, K, T, t_out, stride, 0, 1, st)
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    lib.wun_profile_begin()
    for _ in range(10):
        call()
    buf = C.create_string_buffer(1 << 16)
    lib.wun_profile_end(buf, len(buf))
    import json
    k = [k for k in json.loads(buf.value.decode())["kernels"] if "conv_bf16" in k["name"]][0]
    print("ABL=%s %s: %.1f us/launch" % (os.environ.get("WUN_BF_ABL", "0"), k["name"], 1e3 * k["ms"] / k["launches"]), flush=True)


if __name__ == "__main__":
    shape = [int(v) for v in sys.argv[1:7]]
    if os.environ.get("WUN_BF_CHILD"):
        run(shape)
    else:
        for abl in (0, 1, 2, 4, 8, 3, 5, 6, 7, 15):
            env = dict(os.environ, WUN_BF_ABL=str(abl), WUN_BF_CHILD="1")
            r = subprocess.run([sys.executable, __file__] + sys.argv[1:7], env=env, capture_output=True, text=True)
            print((r.stdout.strip().splitlines() or [r.stderr[-300:]])[-1], flush=True)
