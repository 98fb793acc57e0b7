"""Fused 2x upsampling (UnetAudioSeparator.py:109-118, InterpolationLayer.py:19-39): a producer launch that ends in the
split-K epilogue kernel writes the upsampled copy of its output from there, and (linear interpolation) the split-K
epilogue of the up conv's input gradient applies the adjoint instead of storing d_up; `WUN_NO_FUSE_UPS=1` launches
`upsample_vec_kernel` / `upsample_bwd_vec_kernel` for every level instead.  Both must give the same network: outputs, loss and every gradient
(the backward pass reads the upsampled tensors), bit for bit with linear interpolation and to one rounding of the
interpolation with learned weights (the two kernels may contract `s*a + (1-s)*b` differently)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import wave_u_net_amd as wun                        # noqa: E402
from wave_u_net_amd import training                 # noqa: E402

CASES = [
    # (named config, overrides, batch): shallow nets at short excerpts -> every conv launch splits K, so every level
    # takes the fused path; context (2n - 1 outputs) and same padding (2n, clamped / zero-padded last sample); linear
    # and learned weights; odd and even lengths of the low-rate tensor
    ("m1_context", dict(num_layers=5, num_initial_filters=8, num_frames=300), 3, True),
    ("baseline", dict(num_layers=4, num_initial_filters=8, num_frames=256), 2, True),
    ("full", dict(num_layers=4, num_initial_filters=8, num_frames=210), 2, False),
    ("full", dict(num_layers=3, num_initial_filters=12, num_frames=96, context=False), 4, False),
    ("m1_context", dict(), 2, True),                    # the headline architecture at B = 2: mixed fused / separate levels
]


def _run(cfg, batch, fused, monkeypatch):
    if fused:
        monkeypatch.delenv("WUN_NO_FUSE_UPS", raising=False)
    else:
        monkeypatch.setenv("WUN_NO_FUSE_UPS", "1")
    tr = training.Trainer(dict(cfg, batch_size=batch))
    mix, targets = training.synthetic_source(cfg, tr.batch, tr.t_in, tr.t_out, tr.device, seed=11)()
    outs = tr.sep.get_output(mix, True)
    outs = {k: v.detach().cpu().numpy().copy() for k, v in outs.items()}
    loss = float(tr.sep.loss_and_gradients(targets).item())
    grads = {k: v.detach().cpu().numpy().copy() for k, v in tr.sep.gradients().items()}
    return outs, loss, grads


@pytest.mark.parametrize("case", CASES, ids=[c[0] + str(sorted(c[1].items())) for c in CASES])
def test_fused_upsample_equals_separate_kernel(case, monkeypatch):
    name, over, batch, exact = case
    monkeypatch.setenv("WUN_NO_TUNE", "1")              # same tilings in both runs
    cfg = wun.get_config(name, **over)
    a = _run(cfg, batch, True, monkeypatch)
    b = _run(cfg, batch, False, monkeypatch)
    for k in a[0]:
        if exact:
            assert np.array_equal(a[0][k], b[0][k]), k
        else:
            assert np.abs(a[0][k] - b[0][k]).max() <= 2e-6, k
    assert (a[1] == b[1]) if exact else abs(a[1] - b[1]) <= 1e-6 * max(1.0, abs(b[1]))
    for k in a[2]:
        if exact:
            assert np.array_equal(a[2][k], b[2][k]), k
        else:
            scale = max(1e-12, np.abs(b[2][k]).max())
            assert np.abs(a[2][k] - b[2][k]).max() <= 1e-5 * scale, (k, np.abs(a[2][k] - b[2][k]).max() / scale)
