"""ORACLE (test infrastructure -- never imported by the product): numpy restatement of the
reference's snippet pipeline, the data contract in front of the hot path.

Follows /root/reference/Datasets.py:16-34 (snippet cutting), :49,:76 (zero padding by
(input - output)//2 frames), :188-216 (the tf.data graph: shuffled files, flat_map to snippets,
random_amplify on train, crop_sample, repeat + shuffle buffer, batch_and_drop_remainder) and
/root/reference/Utils.py:26-42 (random_amplify, crop_sample), statement by statement.

TensorFlow's random streams cannot be reproduced, so every random DECISION the graph takes is
asked from a `decisions` object instead of an RNG:
    decisions.file_order(n)          -> permutation of range(n)        random.shuffle(records_files)   :195
    decisions.positions(maxval, num) -> num ints in [0, maxval)        tf.random_uniform(..., int64)   :18
    decisions.gain()                 -> float in [0.7, 1.0)            tf.random_uniform([], 0.7, 1.0)  Utils.py:33
    decisions.pick(buffer_size)      -> int in [0, buffer_size)        Dataset.shuffle(buffer_size)    :213
tests/test_datasets.py records the decisions the product's producers took and replays them here:
for equal decisions the batches must be bit-identical.  Parity unpinned by reference fixtures (the
reference ships no dataset tests); this pins the product to a second, line-by-line restatement."""
import numpy as np


def take_snippets_at_pos(sample, keys, start_pos, input_shape):                      # Datasets.py:29-34
    return [{key: sample[key][pos:pos + input_shape[0], :] for key in keys} for pos in start_pos]


def take_random_snippets(sample, keys, input_shape, num_samples, decisions):        # Datasets.py:16-20
    start_pos = decisions.positions(sample["length"] - input_shape[0], num_samples)
    return take_snippets_at_pos(sample, keys, start_pos, input_shape)


def take_all_snippets(sample, keys, input_shape, output_shape):                      # Datasets.py:22-27
    start_pos = range(0, sample["length"] - input_shape[0], output_shape[0])
    return take_snippets_at_pos(sample, keys, start_pos, input_shape)


def random_amplify(sample, decisions):                                               # Utils.py:26-36
    sample = dict(sample)
    for key, val in list(sample.items()):
        if key != "mix":
            sample[key] = np.float32(decisions.gain()) * val
    acc = None
    for key, val in list(sample.items()):                                            # tf.add_n in dict order
        if key != "mix":
            acc = val if acc is None else acc + val
    sample["mix"] = acc
    return sample


def crop_sample(sample, crop_frames):                                                # Utils.py:38-42
    sample = dict(sample)
    for key, val in list(sample.items()):
        if key != "mix" and crop_frames > 0:
            sample[key] = val[crop_frames:-crop_frames, :]
    return sample


def records(tracks, model_config, input_shape, output_shape):
    """What write_records stores per song (Datasets.py:42-90): every signal zero-padded at both ends
    by pad_frames, plus its length."""
    pad_frames = (input_shape[1] - output_shape[1]) // 2                              # :49
    out = []
    for track in tracks:
        rec = {key: np.pad(track[key], [(pad_frames, pad_frames), (0, 0)], mode="constant", constant_values=0.0)
               for key in model_config["source_names"] + ["mix"]}                     # :76
        length = rec["mix"].shape[0]
        for audio in rec.values():
            assert audio.shape[0] == length                                           # :79-84
        rec["length"] = length
        out.append(rec)
    return out


def get_dataset(model_config, input_shape, output_shape, partition, tracks, decisions, max_batches):
    """The first `max_batches` batches of Datasets.get_dataset's stream (Datasets.py:188-216)."""
    keys = model_config["source_names"] + ["mix"]
    recs = records(tracks, model_config, input_shape, output_shape)

    def one_pass():
        order = decisions.file_order(len(recs)) if partition == "train" else range(len(recs))   # :194-196
        for r in order:
            sample = recs[r]
            if partition == "train":                                                  # :200-204
                snips = take_random_snippets(sample, keys, input_shape[1:], model_config["num_snippets_per_track"],
                                             decisions)
            else:
                snips = take_all_snippets(sample, keys, input_shape[1:], output_shape[1:])
            for s in snips:
                if partition == "train" and model_config["augmentation"]:            # :207-208
                    s = random_amplify(s, decisions)
                yield crop_sample(s, (input_shape[1] - output_shape[1]) // 2)         # :211

    def stream():
        if partition != "train":
            for s in one_pass():
                yield s
            return
        def repeated():                                                               # :214 dataset.repeat()
            while True:
                for s in one_pass():
                    yield s
        buf = []                                                                      # :215 shuffle(buffer_size)
        for s in repeated():
            if len(buf) < model_config["cache_size"]:
                buf.append(s)
                continue
            i = decisions.pick(model_config["cache_size"])
            out, buf[i] = buf[i], s
            yield out

    batches, cur = [], []
    for s in stream():
        cur.append(s)
        if len(cur) == model_config["batch_size"]:                                   # :217 batch_and_drop_remainder
            batches.append({k: np.stack([c[k] for c in cur]) for k in keys})
            cur = []
            if len(batches) == max_batches:
                break
    return batches
