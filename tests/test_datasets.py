"""The data contract in front of the hot path (reference Datasets.py:16-34,76,188-216, Utils.py:26-42):
host-side snippet pipeline, checked on CPU against direct restatements of the reference rules."""
import os

import numpy as np
import pytest

import wave_u_net_amd as wun
from wave_u_net_amd import datasets


def _tracks(cfg, lengths, seed=0):
    rng = np.random.default_rng(seed)
    C = 1 if cfg["mono_downmix"] else 2
    out = []
    for n in lengths:
        out.append(datasets.make_track({k: rng.uniform(-0.4, 0.4, (n, C)).astype(np.float32)
                                        for k in cfg["source_names"]}, cfg))
    return out


def test_make_track_mix_is_sum_and_mono_is_tiled():
    cfg = wun.get_config("baseline_stereo")
    rng = np.random.default_rng(1)
    t = datasets.make_track({"accompaniment": rng.uniform(-1, 1, 100).astype(np.float32),      # mono -> duplicated
                             "vocals": rng.uniform(-1, 1, (100, 2)).astype(np.float32)}, cfg)
    assert t["accompaniment"].shape == (100, 2)
    assert np.array_equal(t["accompaniment"][:, 0], t["accompaniment"][:, 1])                  # Datasets.py:64-66
    assert np.allclose(t["mix"], t["accompaniment"] + t["vocals"])
    with pytest.raises(AssertionError):                                                         # Datasets.py:79-84
        datasets.make_track({"accompaniment": np.zeros((10, 2), np.float32), "vocals": np.zeros((11, 2), np.float32)}, cfg)


def test_positions_follow_the_reference_rules():
    assert list(datasets.all_positions(1000, 300, 100)) == list(range(0, 700, 100))            # tf.range(0, len-in, out)
    assert len(datasets.all_positions(300, 300, 100)) == 0
    rng = np.random.default_rng(0)
    pos = datasets.random_positions(1000, 300, 5000, rng)
    assert pos.min() >= 0 and pos.max() < 700 and pos.max() > 650                               # U[0, len-in)
    with pytest.raises(ValueError):
        datasets.random_positions(300, 300, 1, rng)


def test_random_amplify_and_crop():
    rng = np.random.default_rng(3)
    s = {"a": np.ones((50, 1), np.float32), "b": 2 * np.ones((50, 1), np.float32), "mix": np.zeros((50, 1), np.float32)}
    out = datasets.random_amplify(s, rng)
    ga, gb = out["a"][0, 0], out["b"][0, 0] / 2
    assert 0.7 <= ga <= 1.0 and 0.7 <= gb <= 1.0 and ga != gb                                   # independent scalar gains
    assert np.allclose(out["a"], ga) and np.allclose(out["mix"], out["a"] + out["b"])           # mix re-summed
    c = datasets.crop_sample(out, 7)
    assert c["a"].shape == (36, 1) and c["mix"].shape == (50, 1)                                # targets only
    assert np.array_equal(c["a"], out["a"][7:-7])
    assert datasets.crop_sample(out, 0)["a"].shape == (50, 1)


def test_eval_partition_is_every_hop_in_order_and_drops_the_remainder():
    cfg = wun.get_config("baseline", batch_size=4)
    t_in, t_out = 120, 40
    tracks = _tracks(cfg, [500, 333])
    batches = list(datasets.get_dataset(cfg, [4, t_in, 1], [4, t_out, 1], "valid", tracks))
    pad = (t_in - t_out) // 2
    expect = []
    for t in tracks:
        p = datasets.pad_track(t, pad)
        for pos in range(0, p["mix"].shape[0] - t_in, t_out):
            expect.append((p["mix"][pos:pos + t_in], p["vocals"][pos + pad:pos + t_in - pad]))
    assert len(batches) == len(expect) // 4                                                     # remainder dropped
    k = 0
    for b in batches:
        assert b["mix"].shape == (4, t_in, 1) and b["vocals"].shape == (4, t_out, 1)
        for i in range(4):
            assert np.array_equal(b["mix"][i], expect[k][0]) and np.array_equal(b["vocals"][i], expect[k][1])
            k += 1
    # zero padding at the very start of the first track (Datasets.py:76)
    assert np.all(batches[0]["mix"][0, :pad] == 0)


def test_train_partition_is_endless_shuffled_and_augmented():
    cfg = wun.get_config("baseline", batch_size=8, num_snippets_per_track=5, cache_size=16)
    t_in, t_out = 64, 32
    tracks = _tracks(cfg, [400, 300, 350])
    gen = datasets.get_dataset(cfg, [8, t_in, 1], [8, t_out, 1], "train", tracks, seed=5)
    seen = [next(gen) for _ in range(12)]                                                       # > one pass over 15 snippets
    pad = (t_in - t_out) // 2
    for b in seen:
        assert b["mix"].shape == (8, t_in, 1) and b["accompaniment"].shape == (8, t_out, 1)
        # augmentation keeps mix == sum of (uncropped) amplified sources: check on the cropped core
        core = b["mix"][:, pad:t_in - pad]
        assert np.allclose(core, b["accompaniment"] + b["vocals"], atol=1e-6)
    a = np.concatenate([b["mix"] for b in seen]).reshape(96, -1)
    assert len({row.tobytes() for row in a}) > 80                                               # random positions / gains
    cfg2 = dict(cfg, augmentation=False)
    b = next(datasets.get_dataset(cfg2, [8, t_in, 1], [8, t_out, 1], "train", tracks, seed=5))
    assert np.allclose(b["mix"][:, pad:t_in - pad], b["accompaniment"] + b["vocals"], atol=1e-6)


def test_load_audio_wav_npy_and_partition_layout(tmp_path):
    from scipy.io import wavfile
    cfg = wun.get_config("baseline")
    rng = np.random.default_rng(9)
    root = str(tmp_path)
    for part, n in (("train", 2), ("valid", 1)):
        for ti in range(n):
            d = os.path.join(root, part, "track%d" % ti)
            os.makedirs(d)
            voc = rng.uniform(-0.5, 0.5, (2205, 2)).astype(np.float32)
            acc = (rng.uniform(-0.5, 0.5, (2205, 2)) * 32767).astype(np.int16)
            np.save(os.path.join(d, "vocals.npy"), voc)
            wavfile.write(os.path.join(d, "accompaniment.wav"), 22050, acc)
    tr = datasets.load_partition(root, "train", cfg)
    assert len(tr) == 2 and tr[0]["mix"].shape == (2205, 1)                                     # mono downmix
    assert np.allclose(tr[0]["mix"], tr[0]["vocals"] + tr[0]["accompaniment"], atol=1e-6)
    assert np.abs(tr[0]["accompaniment"]).max() <= 1.0
    wavfile.write(os.path.join(root, "bad.wav"), 44100, np.zeros(10, np.int16))
    with pytest.raises(NotImplementedError):
        datasets.load_audio(os.path.join(root, "bad.wav"), expected_sr=22050)


def test_cli_argument_parsing():
    from importlib import import_module
    cli = import_module("wave_u_net_amd.__main__")
    cmd, name, over, opts = cli._parse(["train", "with", "cfg.full_44KHz", "model_config.epoch_it=7",
                                        "model_config.upsampling=linear", "data_root=/x", "experiment_id=3"])
    assert (cmd, name) == ("train", "full_44KHz")
    assert over == {"epoch_it": 7, "upsampling": "linear"} and opts == {"data_root": "/x", "experiment_id": 3}
    with pytest.raises(SystemExit):
        cli._parse(["frobnicate"])


# ---------------------------------------------------------------------------------------------
# the producers against the line-by-line restatement of the reference pipeline (oracle/datasets_np.py)
# ---------------------------------------------------------------------------------------------
class _RecordingRng(object):
    """numpy Generator that logs every random decision the product's pipeline takes, by kind."""

    def __init__(self, seed):
        self.rng = np.random.default_rng(seed)
        self.order, self.pos, self.gain, self.picks = [], [], [], []

    def permutation(self, n):
        r = self.rng.permutation(n)
        self.order.append([int(v) for v in r])
        return r

    def integers(self, lo, hi, size=None, dtype=np.int64):
        r = self.rng.integers(lo, hi, size=size, dtype=dtype)
        if size is None:
            self.picks.append(int(r))
        else:
            self.pos.extend(int(v) for v in r)
        return r

    def uniform(self, lo, hi, size=None):
        r = self.rng.uniform(lo, hi, size=size)
        self.gain.extend(np.atleast_1d(r).astype(np.float32).tolist())
        return r

    def shuffle(self, x):
        self.rng.shuffle(x)


class _Replay(object):
    """The `decisions` object of oracle/datasets_np.py, answering from a recording."""

    def __init__(self, rec):
        self.order, self.pos, self.gains, self.picks = list(rec.order), list(rec.pos), list(rec.gain), list(rec.picks)

    def file_order(self, n):
        o = self.order.pop(0)
        assert sorted(o) == list(range(n))
        return o

    def positions(self, maxval, num):
        out, self.pos = self.pos[:num], self.pos[num:]
        assert len(out) == num and all(0 <= p < maxval for p in out)          # tf.random_uniform(0, maxval)
        return out

    def gain(self):
        g = self.gains.pop(0)
        assert 0.7 <= g <= 1.0
        return g

    def pick(self, buffer_size):
        i = self.picks.pop(0)
        assert 0 <= i < buffer_size
        return i


@pytest.mark.parametrize("config,partition,augment", [("baseline", "train", True), ("full_multi_instrument", "train", True),
                                                      ("baseline_stereo", "train", False), ("full", "valid", False)])
def test_host_pipeline_equals_the_reference_restatement(config, partition, augment):
    """get_dataset() vs oracle/datasets_np.py (Datasets.py:188-216 + Utils.py:26-42 restated line by
    line) under the SAME random decisions: bit-identical batches."""
    from oracle import datasets_np
    cfg = wun.get_config(config, batch_size=4, num_snippets_per_track=3, cache_size=7, augmentation=augment)
    C = 1 if cfg["mono_downmix"] else 2
    t_in, t_out = 90, 40
    tracks = _tracks(cfg, [400, 257, 333], seed=4)
    rec = _RecordingRng(11)
    gen = datasets.get_dataset(cfg, [4, t_in, C], [4, t_out, C], partition, tracks, rng=rec)
    nb = 9 if partition == "train" else 10 ** 6
    got = []
    for b in gen:
        got.append(b)
        if len(got) == nb:
            break
    want = datasets_np.get_dataset(cfg, [4, t_in, C], [4, t_out, C], partition, tracks, _Replay(rec), len(got))
    assert len(want) == len(got) >= 2
    for g, w in zip(got, want):
        assert sorted(g) == sorted(w)
        for k in g:
            assert g[k].dtype == np.float32 and g[k].shape == w[k].shape
            assert np.array_equal(g[k], w[k]), k
    if partition == "train":
        assert len(rec.order) >= 2 and len(rec.picks) > 0          # more than one pass, buffer in use


def test_device_source_delivers_the_host_pipelines_batches():
    """DeviceSnippetSource (gather + gain + sum + crop on the device) == get_dataset(train) for the same
    seed, bit for bit (device "cpu" here; tests/test_validation_gpu.py runs it on the GPU)."""
    import torch
    cfg = wun.get_config("full_multi_instrument", batch_size=4, num_snippets_per_track=3, cache_size=5)
    t_in, t_out = 90, 40
    tracks = _tracks(cfg, [400, 257, 333], seed=6)
    gen = datasets.get_dataset(cfg, [4, t_in, 2], [4, t_out, 2], "train", tracks, seed=21)
    src = datasets.DeviceSnippetSource(cfg, tracks, t_in, t_out, 4, "cpu", seed=21)
    for _ in range(6):
        hb = next(gen)
        mix, targets = src()
        assert torch.equal(mix, torch.from_numpy(hb["mix"]))
        for si, name in enumerate(cfg["source_names"]):
            assert torch.equal(targets[si], torch.from_numpy(hb[name])), name
