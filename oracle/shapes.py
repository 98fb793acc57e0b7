"""ORACLE (test infrastructure, never imported by the product path).

Integer shape logic of the Wave-U-Net hot path, restated from the reference:

  * get_padding      <- /root/reference/Models/UnetAudioSeparator.py:34-83
  * crop_offsets     <- /root/reference/Utils.py:104-123 (centre crop; odd remainder
                        is removed at the END)
  * layer_table      <- shape walk of get_output, UnetAudioSeparator.py:97-142
  * variable_table   <- TF variable creation order implied by UnetAudioSeparator.py:92-142,
                        InterpolationLayer.py:19, OutputLayer.py:8,15

Pinned against the reference's own code: oracle/make_golden.py executes the
reference's get_padding (pure numpy) and stores its answers in
tests/golden/get_padding.json; tests/test_oracle_shapes.py compares.
"""
import math
import numpy as np

BASE_MODEL_CONFIG = {  # /root/reference/Config.py:9-39 (only the keys the hot path reads)
    "batch_size": 16,
    "init_sup_sep_lr": 1e-4,
    "epoch_it": 2000,
    "num_layers": 12,
    "filter_size": 15,
    "merge_filter_size": 5,
    "input_filter_size": 15,
    "output_filter_size": 1,
    "num_initial_filters": 24,
    "num_frames": 16384,
    "expected_sr": 22050,
    "mono_downmix": True,
    "output_type": "direct",
    "output_activation": "tanh",
    "context": False,
    "network": "unet",
    "upsampling": "linear",
    "task": "voice",
    "augmentation": True,
    "raw_audio_loss": True,
    "worse_epochs": 20,
}


def finalize_config(cfg):
    """Derived keys, /root/reference/Config.py:42-50."""
    cfg = dict(cfg)
    if "source_names" not in cfg:
        if cfg["task"] == "multi_instrument":
            cfg["source_names"] = ["bass", "drums", "other", "vocals"]
        elif cfg["task"] == "voice":
            cfg["source_names"] = ["accompaniment", "vocals"]
        else:
            raise NotImplementedError
    cfg["num_sources"] = len(cfg["source_names"])
    cfg["num_channels"] = 1 if cfg["mono_downmix"] else 2
    return cfg


def get_padding(cfg, shape):
    """UnetAudioSeparator.get_padding, UnetAudioSeparator.py:34-83.

    shape = [batch, desired_output_frames, anything]; returns (input_shape, output_shape)
    as lists [B, T, C]."""
    C = 1 if cfg["mono_downmix"] else 2
    if not cfg["context"]:
        return [int(shape[0]), int(shape[1]), C], [int(shape[0]), int(shape[1]), C]   # :83
    L = cfg["num_layers"]
    rem = float(shape[1])                              # :43
    rem = rem - cfg["output_filter_size"] + 1          # :46
    for _ in range(L):                                 # :49-51
        rem = rem + cfg["merge_filter_size"] - 1
        rem = (rem + 1.0) / 2.0
    x = int(math.ceil(rem))                            # :54
    assert x >= 2                                      # :55
    out = x
    inp = x + cfg["filter_size"] - 1                   # :62
    for i in range(L):                                 # :65-73
        out = 2 * out - 1
        out = out - cfg["merge_filter_size"] + 1
        inp = 2 * inp - 1
        if i < L - 1:
            inp = inp + cfg["filter_size"] - 1
        else:
            inp = inp + cfg["input_filter_size"] - 1
    out = out - cfg["output_filter_size"] + 1          # :76
    return [int(shape[0]), inp, C], [int(shape[0]), out, C]


def crop_offsets(t_from, t_to):
    """Utils.crop, Utils.py:112-123: returns (start, end_removed)."""
    diff = t_from - t_to
    assert diff >= 0
    start = diff // 2
    return start, diff - start


def _conv_len(t, k, same):
    return t if same else t - k + 1


def layer_table(cfg, t_in):
    """Walk get_output's shapes (UnetAudioSeparator.py:97-142) for an input of t_in frames.

    Returns dict with per-layer channel counts and lengths."""
    L, F = cfg["num_layers"], cfg["num_initial_filters"]
    Kd, Ku, Ko = cfg["filter_size"], cfg["merge_filter_size"], cfg["output_filter_size"]
    C = 1 if cfg["mono_downmix"] else 2
    same = not cfg["context"]
    down = []
    t, cin = t_in, C
    for i in range(L):
        cout = F + F * i
        t_conv = _conv_len(t, Kd, same)
        assert t_conv >= 1
        t_dec = (t_conv + 1) // 2                       # [:, ::2, :]  :100
        down.append(dict(cin=cin, cout=cout, t_in=t, t_conv=t_conv, t_dec=t_dec))
        t, cin = t_dec, cout
    cb = F + F * L
    t_b = _conv_len(t, Kd, same)
    assert t_b >= 1
    bott = dict(cin=cin, cout=cb, t_in=t, t_conv=t_b)
    up = []
    t, ccur = t_b, cb
    for i in range(L):
        t_up = 2 * t - 1 if cfg["context"] else 2 * t   # :115 / :117 / InterpolationLayer.py:32
        enc = down[L - 1 - i]
        if same:
            assert enc["t_conv"] == t_up                # :121
        start, end = crop_offsets(enc["t_conv"], t_up)
        cout = F + F * (L - i - 1)
        t_conv = _conv_len(t_up, Ku, same)
        assert t_conv >= 1
        up.append(dict(c_skip=enc["cout"], c_cur=ccur, cin=enc["cout"] + ccur, cout=cout,
                       t_cur=t, t_up=t_up, crop_start=start, crop_end=end, t_conv=t_conv))
        t, ccur = t_conv, cout
    in_start, in_end = crop_offsets(t_in, t)            # :127
    t_out = _conv_len(t, Ko, same)
    head = dict(cin=C + ccur, c_in_mix=C, c_feat=ccur, t_feat=t, t_out=t_out,
                in_crop_start=in_start, in_crop_end=in_end)
    if cfg["output_type"] == "difference":
        s2, e2 = crop_offsets(t, t_out)                 # OutputLayer.py:20 (crop of the cropped mix)
        head["mix_crop_start"] = in_start + s2
    return dict(down=down, bottleneck=bott, up=up, head=head, t_in=t_in, t_out=t_out, C=C)


def variable_table(cfg):
    """(name, shape) of every trainable variable in TF creation order."""
    cfg = finalize_config(cfg)
    L, F = cfg["num_layers"], cfg["num_initial_filters"]
    Kd, Ku, Ko = cfg["filter_size"], cfg["merge_filter_size"], cfg["output_filter_size"]
    C = cfg["num_channels"]
    out = []
    n = [0]

    def conv(k, cin, cout):
        name = "separator/conv1d" if n[0] == 0 else "separator/conv1d_%d" % n[0]
        n[0] += 1
        out.append((name + "/kernel", [k, cin, cout]))
        out.append((name + "/bias", [cout]))

    cin = C
    for i in range(L):
        conv(Kd, cin, F + F * i)
        cin = F + F * i
    conv(Kd, cin, F + F * L)
    ccur = F + F * L
    for i in range(L):
        if cfg["upsampling"] == "learned":
            out.append(("separator/interp_%d" % i, [ccur]))
        cskip = F + F * (L - 1 - i)
        conv(Ku, cskip + ccur, F + F * (L - i - 1))
        ccur = F + F * (L - i - 1)
    n_head = cfg["num_sources"] if cfg["output_type"] == "direct" else cfg["num_sources"] - 1
    for _ in range(n_head):
        conv(Ko, C + ccur, C)
    return out


def num_params(cfg):
    return int(sum(int(np.prod(s)) for _, s in variable_table(cfg)))
