"""Tiled full-track inference: the reference's Evaluate.predict_track
(/root/reference/Evaluate.py:82-145) around the MI355X forward pass.

Semantics kept exactly: mono downmix or mono->stereo duplication (:98-104), zero padding of short
inputs (:108-113), symmetric context padding of (input_frames - output_frames)//2 (:121-122),
hops of `output_frames` with the LAST hop re-aligned to the end of the track (:125-128), removal of
the extra padding (:141-143).  MI355X-first difference: hops are evaluated `batch_hops` at a time in
one get_output call instead of one sess.run per hop (identical results, far fewer launches).
Resampling (librosa in the reference, :106) is out of scope: the audio must already be at
model_config["expected_sr"].
"""
import numpy as np
import torch


def predict_track(model_config, separator, mix_audio, mix_sr=None, batch_hops=16):
    """mix_audio: [n_frames, n_channels] float array.  Returns {source_name: float32 [n_frames, C]}."""
    mix_audio = np.asarray(mix_audio, dtype=np.float32)
    assert mix_audio.ndim == 2                                                   # Evaluate.py:97
    if mix_sr is not None and int(mix_sr) != int(model_config["expected_sr"]):
        raise NotImplementedError("resampling is not part of the hot path; provide audio at expected_sr")
    if model_config["mono_downmix"]:
        mix_audio = np.mean(mix_audio, axis=1, keepdims=True)                    # :98-99
    elif mix_audio.shape[1] == 1:
        mix_audio = np.tile(mix_audio, [1, 2])                                   # :101-102

    in_shape, out_shape = separator.get_padding(np.array([1, model_config["num_frames"], 0]))
    input_frames, output_frames = int(in_shape[1]), int(out_shape[1])

    if mix_audio.shape[0] < input_frames:                                        # :108-113
        extra_pad = input_frames - mix_audio.shape[0]
        mix_audio = np.pad(mix_audio, [(0, extra_pad), (0, 0)], mode="constant")
    else:
        extra_pad = 0
    n_frames = mix_audio.shape[0]
    names = list(model_config["source_names"])
    preds = {n: np.zeros(mix_audio.shape, np.float32) for n in names}           # :117

    pad = (input_frames - output_frames) // 2                                    # :121
    padded = np.pad(mix_audio, [(pad, pad), (0, 0)], mode="constant")

    positions = []
    for pos in range(0, n_frames, output_frames):                                # :125-128
        if pos + output_frames > n_frames:
            pos = n_frames - output_frames
        positions.append(pos)

    for k in range(0, len(positions), batch_hops):
        chunk = positions[k:k + batch_hops]
        batch = np.stack([padded[p:p + input_frames, :] for p in chunk])
        outs = separator.get_output(batch, False)                                # training=False: AudioClip active
        for n in names:
            o = outs[n]
            o = o.detach().cpu().numpy() if torch.is_tensor(o) else np.asarray(o)
            for bi, p in enumerate(chunk):
                preds[n][p:p + output_frames] = o[bi]                            # :139
    if extra_pad > 0:                                                            # :141-143
        preds = {n: v[:-extra_pad, :] for n, v in preds.items()}
    return preds


def produce_source_estimates(model_config, load_model, input_path, output_path=None, separator=None):
    """Evaluate.produce_source_estimates (Evaluate.py:160-194): separate one mixture file with a
    checkpoint and write <input file name>_<source>.wav next to it (or into output_path).  WAV/NPY
    input at expected_sr (no resampling, no MP3 decoding here).  Returns {source: [T, C]}."""
    import os
    from scipy.io import wavfile
    from . import datasets
    from .separator import UnetAudioSeparator
    audio = datasets.load_audio(input_path, mono=False, expected_sr=model_config["expected_sr"])
    sep = separator if separator is not None else UnetAudioSeparator(model_config)
    if load_model is not None:
        from .checkpoint import load_checkpoint
        load_checkpoint(sep, load_model, with_optimizer=False)     # .npz or a TensorFlow V2 checkpoint prefix
    preds = predict_track(model_config, sep, audio, model_config["expected_sr"])
    # Evaluate.predict (:59-80): mono models are evaluated on the downmix; estimates are tiled back to
    # the input's channel count
    if model_config["mono_downmix"] and audio.shape[1] > 1:
        preds = {k: np.tile(v, [1, audio.shape[1]]) for k, v in preds.items()}
    folder, name = os.path.split(input_path)
    if output_path is None:
        output_path = folder
    os.makedirs(output_path or ".", exist_ok=True)
    for source_name, source_audio in preds.items():
        wavfile.write(os.path.join(output_path, name) + "_" + source_name + ".wav", int(model_config["expected_sr"]),
                      np.asarray(source_audio, np.float32))
    return preds
