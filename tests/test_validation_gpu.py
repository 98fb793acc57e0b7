"""Validation loss (Test.py:11-92), the early-stopping driver (Training.py:123-151), the on-GPU batch
producer and the file-level predict entry (Evaluate.py:160-194) on the MI355X path."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import shapes, waveunet_torch as wt            # noqa: E402   (checker only)

import wave_u_net_amd as wun                               # noqa: E402
from wave_u_net_amd import datasets, validation, evaluate  # noqa: E402


def _cfg(tmp, **kw):
    base = dict(num_layers=3, num_initial_filters=8, num_frames=40, batch_size=4, epoch_it=3, worse_epochs=1,
                num_snippets_per_track=6, cache_size=8, model_base_dir=os.path.join(tmp, "ckpt"),
                log_dir=os.path.join(tmp, "logs"), init_sup_sep_lr=1e-3)
    base.update(kw)
    return wun.get_config("baseline_stereo", **base)


def _tracks(cfg, lengths, seed):
    rng = np.random.default_rng(seed)
    return [datasets.make_track({k: (rng.uniform(-0.4, 0.4, (n, 2))).astype(np.float32) for k in cfg["source_names"]}, cfg)
            for n in lengths]


def test_validation_loss_matches_oracle(tmp_path):
    cfg = _cfg(str(tmp_path))
    sep = wun.UnetAudioSeparator(cfg, seed=11)
    tracks = _tracks(cfg, [700, 900], 3)
    got = validation.test(cfg, "valid", "exp", None, tracks=tracks, separator=sep)
    # oracle: same weights, inference-mode forward, running mean over the same batches
    ocfg = shapes.finalize_config(dict(shapes.BASE_MODEL_CONFIG, **{k: cfg[k] for k in shapes.BASE_MODEL_CONFIG if k in cfg}))
    in_shape, out_shape = sep.get_padding(np.array([cfg["batch_size"], cfg["num_frames"], 0]))
    params = {n: v.detach().cpu().numpy() for n, v in sep.variables().items()}
    tp = wt.params_to_torch([(n, params[n]) for n, _ in wt.init_params(ocfg, 0)], torch.float64)
    total, k = 0.0, 1
    for b in datasets.get_dataset(cfg, in_shape, out_shape, "valid", tracks):
        outs = wt.get_output(ocfg, tp, torch.from_numpy(b["mix"]).double(), training=False)
        cur = sum(float(torch.mean((torch.from_numpy(b[n]).double() - outs[n]) ** 2)) for n in cfg["source_names"]) / cfg["num_sources"]
        total += (cur - total) / k
        k += 1
    assert k > 2
    assert abs(got - total) <= 1e-5 * max(1.0, abs(total))                                      # fp32 tolerance
    assert os.path.exists(os.path.join(cfg["log_dir"], "exp", "test.jsonl"))


def test_device_snippet_source_contract():
    cfg = _cfg("/tmp")
    sep = wun.UnetAudioSeparator(cfg)
    in_shape, out_shape = sep.get_padding(np.array([4, cfg["num_frames"], 0]))
    t_in, t_out = int(in_shape[1]), int(out_shape[1])
    tracks = _tracks(cfg, [600, 450, 800], 5)
    src = datasets.DeviceSnippetSource(cfg, tracks, t_in, t_out, 4, "cuda:0", seed=1)
    pad = (t_in - t_out) // 2
    mixes = []
    for _ in range(5):
        mix, targets = src()
        assert mix.shape == (4, t_in, 2) and targets.shape == (2, 4, t_out, 2) and mix.dtype == torch.float32
        assert torch.allclose(mix[:, pad:t_in - pad], targets.sum(0), atol=1e-6)                # mix = sum of amplified sources
        mixes.append(mix.cpu().numpy())
    assert not np.array_equal(mixes[0], mixes[1])
    # same seed => the host pipeline's batches, bit for bit (both consume datasets.snippet_descriptors)
    host = datasets.get_dataset(dict(cfg, batch_size=4), [4, t_in, 2], [4, t_out, 2], "train", tracks, seed=1)
    src_b = datasets.DeviceSnippetSource(cfg, tracks, t_in, t_out, 4, "cuda:0", seed=1)
    for _ in range(3):
        hb = next(host)
        mix, targets = src_b()
        assert np.array_equal(mix.cpu().numpy(), hb["mix"])
        for si, name in enumerate(cfg["source_names"]):
            assert np.array_equal(targets[si].cpu().numpy(), hb[name]), name
    # without augmentation a snippet is a window of the stored (padded) mix
    cfg2 = dict(cfg, augmentation=False)
    src2 = datasets.DeviceSnippetSource(cfg2, tracks[:1], t_in, t_out, 2, "cuda:0", seed=2)
    mix, _ = src2()
    padded = datasets.pad_track(tracks[0], pad)["mix"]
    win = np.lib.stride_tricks.sliding_window_view(padded, (t_in, 2))[:, 0]
    for i in range(2):
        assert (np.abs(win - mix[i].cpu().numpy()[None]).reshape(win.shape[0], -1).max(axis=1) == 0).any()


def test_optimise_early_stopping_driver(tmp_path, monkeypatch):
    monkeypatch.setenv("WUN_NO_TUNE", "1")
    cfg = _cfg(str(tmp_path))
    data = {"train": _tracks(cfg, [700, 800], 1), "valid": _tracks(cfg, [600], 2), "test": _tracks(cfg, [650], 3)}
    best, test_loss = validation.optimise(cfg, "exp7", data=data, max_epochs=3)
    assert best is not None and os.path.exists(best) and os.path.basename(best).startswith("exp7-")
    assert np.isfinite(test_loss) and test_loss > 0
    lines = [eval(l) for l in open(os.path.join(cfg["log_dir"], "exp7", "test.jsonl"))]
    assert [l["partition"] for l in lines].count("valid") == 3 and lines[-1]["partition"] == "test"
    steps = [l["global_step"] for l in lines if l["partition"] == "valid"]
    assert steps == [3, 6, 9]                                                                   # epochs resume from the previous checkpoint


def test_produce_source_estimates_writes_one_file_per_source(tmp_path):
    from scipy.io import wavfile
    cfg = _cfg(str(tmp_path), expected_sr=22050)
    rng = np.random.default_rng(4)
    audio = (rng.uniform(-0.5, 0.5, (3000, 2)) * 32767).astype(np.int16)
    inp = os.path.join(str(tmp_path), "song.wav")
    wavfile.write(inp, 22050, audio)
    sep = wun.UnetAudioSeparator(cfg, seed=3)
    preds = evaluate.produce_source_estimates(cfg, None, inp, os.path.join(str(tmp_path), "out"), separator=sep)
    for name in cfg["source_names"]:
        f = os.path.join(str(tmp_path), "out", "song.wav_" + name + ".wav")                     # Evaluate.py:193
        sr, data = wavfile.read(f)
        assert sr == 22050 and data.shape == (3000, 2)
        assert np.allclose(data, preds[name], atol=1e-6)
    # difference output: the estimates sum to the (clipped) mixture
    mixf = audio.astype(np.float32) / 32768.0
    assert np.abs(sum(preds.values()) - mixf).max() < 1e-3
