"""model_config handling: same keys, same named configs as the reference's sacred
ingredient (/root/reference/Config.py:4-161), as a plain dict (sacred is not needed)."""
import copy

BASE_MODEL_CONFIG = {            # Config.py:9-39
    "model_base_dir": "checkpoints",
    "log_dir": "logs",
    "batch_size": 16,
    "init_sup_sep_lr": 1e-4,
    "epoch_it": 2000,
    "cache_size": 4000,
    "num_workers": 4,
    "num_snippets_per_track": 100,
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

NAMED_CONFIGS = {                # Config.py:52-161 (the Wave-U-Net ones)
    "baseline": {},
    "baseline_diff": {"output_type": "difference"},
    "baseline_context": {"output_type": "difference", "context": True},
    "baseline_stereo": {"output_type": "difference", "context": True, "mono_downmix": False},
    "full": {"output_type": "difference", "context": True, "upsampling": "learned",
             "mono_downmix": False},
    "full_44KHz": {"output_type": "difference", "context": True, "upsampling": "learned",
                   "mono_downmix": False, "expected_sr": 44100},
    "baseline_context_smallfilter_deep": {"output_type": "difference", "context": True,
                                          "num_layers": 14, "duration": 7, "filter_size": 5,
                                          "merge_filter_size": 1},
    "full_multi_instrument": {"output_type": "difference", "context": True,
                              "upsampling": "linear", "mono_downmix": False,
                              "task": "multi_instrument"},
    "baseline_comparison": {"batch_size": 4, "output_type": "difference", "context": True,
                            "num_frames": 768 * 127 + 1024, "duration": 13,
                            "expected_sr": 8192, "num_initial_filters": 34},   # Config.py:123-134
    # BASELINE.json configs[1]: the M1 architecture run with input context (~147k samples in)
    "m1_context": {"context": True},
    # BASELINE.json configs[4]: 16 levels / 48 base channels, stereo, 4 sources, same padding,
    # 589 824-sample excerpts (9 * 2^16; a 16-level context model would need >= 2.1 M input samples)
    "deep_l16_f48": {"num_layers": 16, "num_initial_filters": 48, "mono_downmix": False,
                     "task": "multi_instrument", "output_type": "difference", "num_frames": 589824},
}


def finalize(model_config):
    """Derived keys (Config.py:42-50)."""
    cfg = dict(model_config)
    if cfg["task"] == "multi_instrument":
        cfg.setdefault("source_names", ["bass", "drums", "other", "vocals"])
    elif cfg["task"] == "voice":
        cfg.setdefault("source_names", ["accompaniment", "vocals"])
    else:
        raise NotImplementedError(cfg["task"])
    cfg["num_sources"] = len(cfg["source_names"])
    cfg["num_channels"] = 1 if cfg["mono_downmix"] else 2
    return cfg


def get_config(name="baseline", **overrides):
    cfg = copy.deepcopy(BASE_MODEL_CONFIG)
    cfg.update(NAMED_CONFIGS[name])
    cfg.update(overrides)
    return finalize(cfg)
