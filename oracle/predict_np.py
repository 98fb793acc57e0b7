"""ORACLE (test infrastructure): line-by-line numpy restatement of the reference's
Evaluate.predict_track (/root/reference/Evaluate.py:82-145) for an arbitrary `run(mix_part)`
callable standing in for sess.run (one hop at a time, exactly like the reference).  No resampling
(librosa is absent): audio is assumed to be at expected_sr."""
import numpy as np


def predict_track_ref(model_config, run, mix_audio, sep_input_shape, sep_output_shape):
    assert len(mix_audio.shape) == 2
    if model_config["mono_downmix"]:
        mix_audio = np.mean(mix_audio, axis=1, keepdims=True)
    else:
        if mix_audio.shape[1] == 1:
            mix_audio = np.tile(mix_audio, [1, 2])
    if mix_audio.shape[0] < sep_input_shape[1]:
        extra_pad = sep_input_shape[1] - mix_audio.shape[0]
        mix_audio = np.pad(mix_audio, [(0, extra_pad), (0, 0)], mode="constant", constant_values=0.0)
    else:
        extra_pad = 0
    source_time_frames = mix_audio.shape[0]
    source_preds = {name: np.zeros(mix_audio.shape, np.float32) for name in model_config["source_names"]}
    input_time_frames = sep_input_shape[1]
    output_time_frames = sep_output_shape[1]
    pad_time_frames = (input_time_frames - output_time_frames) // 2
    mix_audio_padded = np.pad(mix_audio, [(pad_time_frames, pad_time_frames), (0, 0)], mode="constant",
                              constant_values=0.0)
    for source_pos in range(0, source_time_frames, output_time_frames):
        if source_pos + output_time_frames > source_time_frames:
            source_pos = source_time_frames - output_time_frames
        mix_part = mix_audio_padded[source_pos:source_pos + input_time_frames, :]
        mix_part = np.expand_dims(mix_part, axis=0)
        source_parts = run(mix_part)
        for name in model_config["source_names"]:
            source_preds[name][source_pos:source_pos + output_time_frames] = source_parts[name][0, :, :]
    if extra_pad > 0:
        source_preds = {name: source_preds[name][:-extra_pad, :] for name in list(source_preds.keys())}
    return source_preds
