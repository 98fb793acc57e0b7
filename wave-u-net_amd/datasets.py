"""The data contract in front of the hot path: the reference's snippet pipeline
(/root/reference/Datasets.py:16-34,76,188-216 and Utils.py:26-42) without TFRecords/tf.data.

A *track* is a dict {source_name: [T, C] float32, ..., "mix": [T, C]} (what Datasets.write_records
stores per song, :57-90).  get_dataset() reproduces the reference's stream of batches
  pad both ends by (input_frames - output_frames)//2 zeros                    (Datasets.py:49,76)
  train: num_snippets_per_track random snippets per track                      (:16-20,201-202)
  valid/test: every snippet at hop = output_frames                             (:22-27,203-204)
  train + augmentation: per-source gain U(0.7, 1.0), mix := sum of sources     (Utils.py:26-36)
  targets centre-cropped to the output length                                  (Utils.py:38-42)
  train: repeat + shuffle buffer of cache_size snippets                        (Datasets.py:211-213)
  batches of batch_size, remainder dropped                                     (:215)
on the host in numpy.  DeviceSnippetSource is the MI355X-first producer for training: every padded
track lives in HBM (a full MUSDB train set at 22 kHz mono fp32 is ~4 GB of 288 GB), and a batch
is one gather + gain + sum + crop on the GPU, so the input side keeps up with a ~10 ms step.

Decoding MUSDB stems / resampling (librosa, musdb, soundfile in the reference, Datasets.py:119-186)
is out of scope: tracks come from WAV (scipy) or .npy files already at model_config["expected_sr"].
"""
import os

import numpy as np


# --------------------------------------------------------------------------------------
# tracks
# --------------------------------------------------------------------------------------
def load_audio(path, mono=False, expected_sr=None):
    """[T, C] float32 from a .wav (PCM16/32 or float) or .npy file (Utils.load without resampling)."""
    if path.endswith(".npy"):
        audio = np.load(path).astype(np.float32)
        sr = expected_sr
    else:
        from scipy.io import wavfile
        sr, raw = wavfile.read(path)
        if raw.dtype == np.int16:
            audio = raw.astype(np.float32) / 32768.0
        elif raw.dtype == np.int32:
            audio = raw.astype(np.float32) / 2147483648.0
        elif raw.dtype == np.uint8:
            audio = (raw.astype(np.float32) - 128.0) / 128.0
        else:
            audio = raw.astype(np.float32)
    if audio.ndim == 1:
        audio = audio[:, None]
    if expected_sr is not None and sr is not None and int(sr) != int(expected_sr):
        raise NotImplementedError("resampling is out of scope: %s is at %s Hz, expected %s" % (path, sr, expected_sr))
    if mono and audio.shape[1] > 1:
        audio = audio.mean(axis=1, keepdims=True)
    return np.ascontiguousarray(audio, dtype=np.float32)


def make_track(sources, model_config, mix=None):
    """Validated track dict from {source_name: array-or-path}.  Mono tracks are duplicated when
    the model is stereo (Datasets.py:64-66); all signals must have equal shape (:79-84); the mix
    defaults to the sum of the sources."""
    mono = bool(model_config["mono_downmix"])
    track = {}
    for key in model_config["source_names"]:
        a = sources[key]
        a = load_audio(a, mono, model_config.get("expected_sr")) if isinstance(a, str) else np.asarray(a, np.float32)
        if a.ndim == 1:
            a = a[:, None]
        if mono and a.shape[1] > 1:
            a = a.mean(axis=1, keepdims=True)
        if not mono and a.shape[1] == 1:
            a = np.tile(a, [1, 2])
        track[key] = np.ascontiguousarray(a, np.float32)
    shapes = {a.shape for a in track.values()}
    assert len(shapes) == 1, "all signals of a track must have the same length and channels"
    if mix is None:
        mix = sum(track[k] for k in model_config["source_names"])
    elif isinstance(mix, str):
        mix = load_audio(mix, mono, model_config.get("expected_sr"))
    mix = np.asarray(mix, np.float32)
    if mix.ndim == 1:
        mix = mix[:, None]
    if not mono and mix.shape[1] == 1:
        mix = np.tile(mix, [1, 2])
    track["mix"] = np.ascontiguousarray(mix, np.float32)
    length, channels = track["mix"].shape
    for a in track.values():
        assert a.shape == (length, channels), "all signals of a track must have the same length and channels"
    return track


def load_track_dir(path, model_config):
    """A directory holding <source_name>.wav|.npy for every source and optionally mix.wav|.npy."""
    def find(stem):
        for ext in (".wav", ".npy"):
            f = os.path.join(path, stem + ext)
            if os.path.exists(f):
                return f
        return None
    srcs = {}
    for key in model_config["source_names"]:
        f = find(key)
        if f is None:
            raise FileNotFoundError("%s: no %s.wav/.npy" % (path, key))
        srcs[key] = f
    return make_track(srcs, model_config, mix=find("mix"))


def load_partition(root, partition, model_config):
    """root/<partition>/<track>/ directories, sorted by name."""
    base = os.path.join(root, partition)
    return [load_track_dir(os.path.join(base, d), model_config)
            for d in sorted(os.listdir(base)) if os.path.isdir(os.path.join(base, d))]


def pad_track(track, pad_frames):
    """Zero padding at both ends (Datasets.py:76)."""
    if pad_frames <= 0:
        return dict(track)
    return {k: np.pad(v, [(pad_frames, pad_frames), (0, 0)], mode="constant") for k, v in track.items()}


# --------------------------------------------------------------------------------------
# snippets
# --------------------------------------------------------------------------------------
def random_positions(length, input_frames, num, rng):
    """tf.random_uniform([num], 0, length - input_frames, int64) (Datasets.py:18)."""
    hi = length - input_frames
    if hi <= 0:
        raise ValueError("track shorter than the network input (%d <= %d)" % (length, input_frames))
    return rng.integers(0, hi, size=num, dtype=np.int64)


def all_positions(length, input_frames, output_frames):
    """tf.range(0, length - input_frames, delta=output_frames) (Datasets.py:24)."""
    return np.arange(0, length - input_frames, output_frames, dtype=np.int64)


def take_snippets_at_pos(track, keys, start_pos, input_frames):
    """{key: [n, input_frames, C]} (Datasets.py:29-34)."""
    return {k: np.stack([track[k][p:p + input_frames, :] for p in start_pos]) if len(start_pos)
            else np.zeros((0, input_frames, track[k].shape[1]), np.float32) for k in keys}


def random_amplify(sample, rng):
    """Per-source scalar gain U(0.7, 1.0); the mix becomes the sum of the amplified sources
    (Utils.py:26-36).  `sample` holds single snippets [T, C]."""
    out = {}
    for key, val in sample.items():
        if key != "mix":
            out[key] = np.float32(rng.uniform(0.7, 1.0)) * val
    out["mix"] = sum(out[k] for k in out)
    return out


def crop_sample(sample, crop_frames):
    """Targets (everything but the mix) lose crop_frames at both ends (Utils.py:38-42)."""
    return {k: (v[crop_frames:-crop_frames, :] if (k != "mix" and crop_frames > 0) else v) for k, v in sample.items()}


def snippet_descriptors(model_config, lengths, input_frames, output_frames, partition, rng):
    """The reference's snippet stream (Datasets.py:188-216) as DESCRIPTORS (track_index, start, gains):
    which snippet of which padded track comes next and, for train + augmentation, the per-source
    gains of Utils.random_amplify (float32 [S], else None).  `lengths` are the padded track lengths.
      train     : endless; per pass the tracks in shuffled order (:195), num_snippets_per_track
                  uniform start positions each (:16-20,201-202), then the shuffle buffer of
                  cache_size elements (:211-213)
      otherwise : one pass, tracks in order, every hop of output_frames (:22-27,203-204)
    Both producers below (host numpy batches, on-GPU gather) consume this one stream, so they
    deliver identical batches for the same seed."""
    train = partition == "train"
    S = len(model_config["source_names"])
    n_tracks = len(lengths)

    def raw():
        while True:
            order = rng.permutation(n_tracks) if train else np.arange(n_tracks)
            for ti in order:
                if train:
                    pos = random_positions(lengths[ti], input_frames, int(model_config["num_snippets_per_track"]), rng)
                else:
                    pos = all_positions(lengths[ti], input_frames, output_frames)
                for p in pos:
                    gains = None
                    if train and model_config["augmentation"]:
                        gains = rng.uniform(0.7, 1.0, size=S).astype(np.float32)
                    yield int(ti), int(p), gains
            if not train:
                return

    def shuffled(stream, buffer_size):
        buf = []
        for s in stream:
            if len(buf) < buffer_size:
                buf.append(s)
                continue
            i = int(rng.integers(0, buffer_size))
            out, buf[i] = buf[i], s
            yield out
        rng.shuffle(buf)                 # (finite streams only; the train stream repeats forever)
        for s in buf:
            yield s

    return shuffled(raw(), int(model_config["cache_size"])) if train else raw()


def materialize(track, keys, source_names, start, input_frames, gains, crop_frames):
    """One snippet {key: [T, C]} from a padded track: cut (Datasets.py:29-34), amplify + re-sum the mix
    (Utils.py:26-36, when gains are given), centre-crop the targets (Utils.py:38-42)."""
    s = {k: track[k][start:start + input_frames, :] for k in keys}
    if gains is not None:
        amp = {k: np.float32(g) * s[k] for k, g in zip(source_names, gains)}
        mix = amp[source_names[0]]
        for k in source_names[1:]:
            mix = mix + amp[k]
        amp["mix"] = mix
        s = amp
    return crop_sample(s, crop_frames)


def get_dataset(model_config, input_shape, output_shape, partition, tracks, seed=1337, rng=None):
    """Generator of batches {source..., "mix"}: mix [B, Tin, C], sources [B, Tout, C]
    (Datasets.get_dataset, Datasets.py:113-218, minus the TFRecord cache).  `tracks` is the list of
    track dicts of this partition.  Train: endless, shuffled; otherwise one pass in order."""
    input_frames, output_frames = int(input_shape[1]), int(output_shape[1])
    assert (input_frames - output_frames) % 2 == 0
    pad = (input_frames - output_frames) // 2
    names = list(model_config["source_names"])
    keys = names + ["mix"]
    batch_size = int(model_config["batch_size"])
    rng = rng if rng is not None else np.random.default_rng(seed)
    padded = [pad_track(t, pad) for t in tracks]
    lengths = [t["mix"].shape[0] for t in padded]
    batch = []
    for ti, pos, gains in snippet_descriptors(model_config, lengths, input_frames, output_frames, partition, rng):
        batch.append(materialize(padded[ti], keys, names, pos, input_frames, gains, pad))
        if len(batch) == batch_size:
            yield {k: np.stack([b[k] for b in batch]) for k in keys}
            batch = []
    # remainder dropped (batch_and_drop_remainder, :215)


def batch_to_device(batch, model_config, device):
    """(mix [B,Tin,C], targets [S,B,Tout,C]) torch tensors in source_names order: the step's inputs."""
    import torch
    mix = torch.from_numpy(np.ascontiguousarray(batch["mix"])).to(device)
    targets = torch.from_numpy(np.stack([batch[k] for k in model_config["source_names"]])).to(device)
    return mix, targets


class DeviceSnippetSource(object):
    """Training batches produced on the GPU from tracks resident in HBM.

    The reference's train pipeline exactly (per pass: shuffled tracks, num_snippets_per_track uniform
    positions per track, per-source gain U(0.7,1.0) and mix = sum when augmentation is on, shuffle
    buffer of cache_size snippets, centre-cropped targets): the snippet stream is the same
    snippet_descriptors() sequence the host pipeline consumes -- only (track, start, gains) triples
    move through the shuffle buffer on the host, a few bytes per snippet -- and a batch is ONE
    gather + gain + sum + crop on the GPU.  For equal seeds the batches are bit-identical to
    get_dataset(..., "train", ...).  Calling the object returns (mix [B,Tin,C], targets [S,B,Tout,C])."""

    def __init__(self, model_config, tracks, input_frames, output_frames, batch_size, device, seed=1337, rng=None):
        import torch
        self.cfg = model_config
        self.names = list(model_config["source_names"])
        self.t_in, self.t_out = int(input_frames), int(output_frames)
        self.pad = (self.t_in - self.t_out) // 2
        self.batch = int(batch_size)
        self.device = torch.device(device)
        starts, lens, cat = [], [], {k: [] for k in self.names + ["mix"]}
        off = 0
        for t in tracks:
            p = pad_track(t, self.pad)
            n = p["mix"].shape[0]
            if n <= self.t_in:
                raise ValueError("track shorter than the network input")
            starts.append(off); lens.append(n); off += n
            for k in cat:
                cat[k].append(p[k])
        self.track_start = np.asarray(starts, dtype=np.int64)
        self.data = {k: torch.from_numpy(np.concatenate(v)).to(self.device) for k, v in cat.items()}
        self.frame = torch.arange(self.t_in, device=self.device, dtype=torch.int64)
        self.stream = snippet_descriptors(model_config, lens, self.t_in, self.t_out, "train",
                                          rng if rng is not None else np.random.default_rng(seed))

    def __call__(self):
        import torch
        B, S = self.batch, len(self.names)
        desc = [next(self.stream) for _ in range(B)]
        base = torch.from_numpy(np.asarray([self.track_start[ti] + pos for ti, pos, _ in desc], dtype=np.int64))
        idx = base.to(self.device)[:, None] + self.frame[None, :]                          # [B, Tin]
        srcs = torch.stack([self.data[k][idx] for k in self.names])                        # [S, B, Tin, C]
        if desc[0][2] is not None:
            gain = torch.from_numpy(np.stack([g for _, _, g in desc], axis=1)).to(self.device)   # [S, B]
            srcs = srcs * gain[:, :, None, None]
            mix = srcs[0]
            for s in range(1, S):                       # same summation order as the host pipeline
                mix = mix + srcs[s]
        else:
            mix = self.data["mix"][idx]
        targets = srcs[:, :, self.pad:self.t_in - self.pad, :] if self.pad > 0 else srcs
        return mix.contiguous(), targets.contiguous()
