#!/usr/bin/env python3
"""Effective shader clock inside the real training step: runs the headline step with the diagnostic build
(WUN_LIB=libwun_abl.so, WUN_ABLATE=64: every conv launch stamps s_memtime / s_memrealtime per workgroup) and reports
the clock seen by the workgroups of the last conv launch of a step after N warm steps."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import wave_u_net_amd as wun
from wave_u_net_amd import _lib
from wave_u_net_amd.training import Trainer, synthetic_source
lib = _lib.load()
lib.wun_dbg_trace_read.restype = C.c_int; lib.wun_dbg_trace_read.argtypes = [C.c_void_p, C.c_int, C.c_int]
lib.wun_dbg_set_knob.restype = C.c_int; lib.wun_dbg_set_knob.argtypes = [C.c_int, C.c_int]
cfg = wun.get_config("m1_context")
os.environ["WUN_NO_TUNE"] = "1"
tr = Trainer(cfg, batch_size=16)
mix, targets = synthetic_source(cfg, 16, tr.t_in, tr.t_out, tr.device)()
for _ in range(20):
    tr.step(mix, targets)
torch.cuda.synchronize()
os.environ["WUN_ABLATE"] = "64"
for cin, nout in ((72, 96), (96, 72), (96, 120), (168, 72), (264, 288)):
  lib.wun_dbg_set_knob(2, cin); lib.wun_dbg_set_knob(3, nout)
  lib.wun_dbg_trace_read(None, 0, 1)
  print("launches with Cin=%d N=%d:" % (cin, nout))
  for n in (1, 20):
    for _ in range(n):
        tr.step(mix, targets)
    torch.cuda.synchronize()
    host = np.zeros((16384, 16), dtype=np.uint64)
    lib.wun_dbg_trace_read(host.ctypes.data, 16384, 1)
    u = host[host[:, 0] != 0].astype(np.int64)
    d = (u[:, 6] - u[:, 5]); ok = d > 20
    if ok.sum() == 0:
        print("   no workgroups stamped"); continue
    r = (u[ok, 3] - u[ok, 0]) / d[ok]
    print("   after %2d traced steps: %d stamped workgroups, shader clock %.0f MHz (p10 %.0f p90 %.0f), launch span %.0f us" % (
        n, ok.sum(), np.median(r) * 100, np.percentile(r, 10) * 100, np.percentile(r, 90) * 100, (u[:, 6].max() - u[:, 5].min()) / 100.0))
