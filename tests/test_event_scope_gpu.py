"""The plan's cross-stream events carry no system-scope fence by default (wun_plan.hip event_flags()).
Consumers that are NOT kernels follow on the caller's stream: a device-to-host copy of the gradient arena right after
`wun_loss_backward`, and the host-staged all-reduce of the non-overlapped reducer.  This file runs both in every
fence mode (default = none, WUN_EVENT_SCOPE=system, =device) and requires the same bytes from all three."""
import hashlib
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

D2H = r'''
import hashlib, sys
sys.path.insert(0, %r)
import numpy as np, torch
import wave_u_net_amd as wun
from wave_u_net_amd import training
cfg = wun.get_config("full", num_layers=6, num_initial_filters=8, num_frames=300, batch_size=4)
tr = training.Trainer(cfg)
mix, targets = training.synthetic_source(cfg, tr.batch, tr.t_in, tr.t_out, tr.device, seed=5)()
first = None
for it in range(25):
    tr.sep.grads.fill_(float("nan"))
    tr.sep.get_output(mix, True)
    tr.sep.loss_and_gradients(targets)
    # no torch.cuda.synchronize(): the copy below is ordered only by the caller's stream, like any D2H consumer
    g = tr.sep.grads.to("cpu", non_blocking=False).numpy()
    assert np.isfinite(g).all(), it
    h = hashlib.sha256(g.tobytes()).hexdigest()
    first = first or h
    assert h == first, "iteration %%d differs" %% it
print("GRADS", first)
''' % ROOT


def _run(code_or_cmd, mode, extra=None, as_cmd=False):
    env = dict(os.environ, WUN_NO_TUNE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("WUN_EVENT_SCOPE", None)
    if mode:
        env["WUN_EVENT_SCOPE"] = mode
    env.update(extra or {})
    cmd = code_or_cmd if as_cmd else [sys.executable, "-c", code_or_cmd]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, (mode, r.stderr[-3000:])
    return r.stdout


def test_device_to_host_copy_after_backward_sees_every_side_stream_write():
    hashes = {}
    for mode in (None, "system", "device"):
        out = _run(D2H, mode)
        hashes[mode] = [l.split()[1] for l in out.splitlines() if l.startswith("GRADS")][0]
    assert len(set(hashes.values())) == 1, hashes


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_non_overlapped_two_rank_all_reduce_in_every_fence_mode(tmp_path):
    """WUN_NO_OVERLAP=1: the gradient arena is reduced after the backward pass returned -- over gloo here, i.e. through
    a host copy that nothing but the caller's stream orders against the side streams' gradient writes."""
    params = {}
    for mode in (None, "system", "device"):
        out = os.path.join(str(tmp_path), "dp_%s.npz" % (mode or "none"))
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
               "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
               os.path.join(ROOT, "tests", "dp_worker.py"), out, "3"]
        _run(cmd, mode, extra={"WUN_DIST_BACKEND": "gloo", "WUN_NO_OVERLAP": "1"}, as_cmd=True)
        params[mode] = hashlib.sha256(np.load(out)["params"].tobytes()).hexdigest()
    assert len(set(params.values())) == 1, params
