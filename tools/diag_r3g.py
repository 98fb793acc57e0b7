#!/usr/bin/env python3
"""Round-3: workgroup life-cycle trace of the weight-gradient kernel (ablation build, WUN_ABLATE bit 64)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from wave_u_net_amd import _lib
lib = _lib.load()
lib.wun_dbg_trace_read.restype = C.c_int; lib.wun_dbg_trace_read.argtypes = [C.c_void_p, C.c_int, C.c_int]
st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
B = 16
for name, cin, cout, k, t_out, stride in (("down3 dec 72->96 s2", 72, 96, 15, 9203, 2), ("down2 win 48->72 s1", 48, 72, 15, 4105, 1), ("up9 168->72 K5", 168, 72, 5, 4101, 1)):
    T = (t_out - 1) * stride + k
    x = torch.rand(B, cin, T, device="cuda") * 2 - 1
    dz = torch.rand(B, cout, t_out, device="cuda") * 2 - 1
    n = lib.wun_op_conv1d_wgrad_scratch(B, cin, cout, k, t_out)
    scr = torch.empty(int(n), device="cuda")
    dw = torch.empty(k, cin, cout, device="cuda"); db = torch.empty(cout, device="cuda")
    fn = lambda: lib.wun_op_conv1d_wgrad(x.data_ptr(), dz.data_ptr(), dw.data_ptr(), db.data_ptr(), scr.data_ptr(), B, cin, cout, k, T, t_out, stride, 0, st)
    for abl, what in ((0, "full"), (3, "no loads/stores"), (4, "no MFMA")):
        os.environ["WUN_ABLATE"] = str(64 + abl)
        fn(); torch.cuda.synchronize(); lib.wun_dbg_trace_read(None, 0, 1)
        fn(); torch.cuda.synchronize()
        host = np.zeros((16384, 16), dtype=np.uint64)
        lib.wun_dbg_trace_read(host.ctypes.data, 16384, 1)
        u = host[host[:, 0] != 0].astype(np.int64)
        mhz = np.median((u[:, 3] - u[:, 0]) / np.maximum(1, (u[:, 6] - u[:, 5]))) * 100
        f = lambda a, b_: np.median(u[:, a] - u[:, b_]) / mhz
        span = (u[:, 6].max() - u[:, 5].min()) / 100.0
        hw = u[:, 4] & 0xFFFFFFFF; xcc = (u[:, 4] >> 32) & 15
        cuid = ((xcc * 8 + ((hw >> 13) & 7)) * 2 + ((hw >> 12) & 1)) * 16 + ((hw >> 8) & 15)
        percu = np.bincount(np.unique(cuid, return_inverse=True)[1])
        print("%-22s %-16s %d WGs (per CU %d..%d) span %6.1f us | WG median: setup %5.2f first-load-issue %5.2f loop %6.2f store %5.2f total %6.2f | %4.0f MHz" % (
            name, what, len(u), percu.min(), percu.max(), span, f(7, 0), f(1, 7), f(2, 1), f(3, 2), f(3, 0), mhz), flush=True)
    os.environ.pop("WUN_ABLATE")
