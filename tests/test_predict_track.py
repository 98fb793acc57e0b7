"""predict_track tiling (Evaluate.py:82-145): host logic on CPU with a stand-in separator, and on
the GPU against the oracle evaluated hop by hop."""
import numpy as np
import pytest
import torch

from oracle import shapes, waveunet_torch as wt
from oracle.golden_params import GOLDEN_CASES, golden_params
from oracle.predict_np import predict_track_ref

import wave_u_net_amd as wun
from wave_u_net_amd.evaluate import predict_track


class FakeSeparator(object):
    """Deterministic stand-in with the separator surface: output = centre crop * per-source gain."""

    def __init__(self, cfg, t_in, t_out):
        self.cfg, self.t_in, self.t_out = cfg, t_in, t_out
        self.calls = 0

    def get_padding(self, shape):
        c = 1 if self.cfg["mono_downmix"] else 2
        return np.array([shape[0], self.t_in, c]), np.array([shape[0], self.t_out, c])

    def get_output(self, batch, training):
        assert training is False
        self.calls += 1
        pad = (self.t_in - self.t_out) // 2
        core = np.asarray(batch)[:, pad:pad + self.t_out, :]
        return {n: core * (i + 1) + 0.01 * i for i, n in enumerate(self.cfg["source_names"])}


@pytest.mark.parametrize("n_frames", [50, 1000, 1024, 1033, 4099])
@pytest.mark.parametrize("mono,chan", [(True, 2), (False, 1), (False, 2)])
def test_tiling_matches_reference_restatement(n_frames, mono, chan):
    cfg = wun.get_config("baseline", mono_downmix=mono, task="multi_instrument")
    t_in, t_out = 1324, 300
    audio = np.random.default_rng(n_frames).uniform(-1, 1, (n_frames, chan)).astype(np.float32)
    fake = FakeSeparator(cfg, t_in, t_out)
    got = predict_track(cfg, fake, audio, batch_hops=4)
    ref_sep = FakeSeparator(cfg, t_in, t_out)
    c = 1 if mono else 2
    want = predict_track_ref(cfg, lambda part: ref_sep.get_output(part, False), audio, [1, t_in, c], [1, t_out, c])
    assert list(got.keys()) == cfg["source_names"]
    for n in cfg["source_names"]:
        assert got[n].shape == want[n].shape == (n_frames, c)
        assert np.array_equal(got[n], want[n])
    assert fake.calls <= (ref_sep.calls + 3) // 4 + 1          # hops are batched


def test_resampling_is_rejected():
    cfg = wun.get_config("baseline")
    with pytest.raises(NotImplementedError):
        predict_track(cfg, FakeSeparator(cfg, 100, 100), np.zeros((10, 1), np.float32), mix_sr=44100)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["baseline_context_small", "linear_act_eval_small", "baseline_small"])
def test_predict_track_on_gpu_vs_oracle(name):
    from wave_u_net_amd.separator import UnetAudioSeparator
    case = GOLDEN_CASES[name]
    ocfg = shapes.finalize_config(dict(shapes.BASE_MODEL_CONFIG, **case["cfg"]))
    frames = case["frames"]
    cfg = wun.get_config("baseline", num_frames=frames, **case["cfg"])
    params = golden_params(ocfg, case["seed"])
    sep = UnetAudioSeparator(cfg, device="cuda:0")
    i, o = shapes.get_padding(ocfg, [1, frames, 0])
    sep._plan(1, i[1]); sep._active = sep._plans[(1, i[1])]
    sep.load_variables(params)
    n_frames = 5 * o[1] + 17
    audio = np.random.default_rng(3).uniform(-1.5, 1.5, (n_frames, 2)).astype(np.float32)
    got = predict_track(cfg, sep, audio, batch_hops=3)
    tp = wt.params_to_torch(params, torch.float32)

    def run(part):
        outs = wt.get_output(ocfg, tp, torch.from_numpy(np.ascontiguousarray(part)), False)
        return {k: v.numpy() for k, v in outs.items()}
    want = predict_track_ref(dict(ocfg, num_frames=frames), run, audio, i, o)
    for n in ocfg["source_names"]:
        assert got[n].shape == want[n].shape
        assert np.abs(got[n] - want[n]).max() <= 2e-4
