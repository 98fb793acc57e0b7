"""ORACLE (test infrastructure): deterministic weights + case list shared by
oracle/make_golden.py (which runs the reference) and the tests (which must NOT need
/root/reference).  Weights are a pure function of (cfg, seed) so that big cases do not
have to be committed."""
import math
import numpy as np

from . import shapes


def golden_params(cfg, seed):
    """[(tf_name, float32 ndarray)] in TF creation order.  Kernels glorot-uniform,
    biases U(-0.1, 0.1) (non-zero on purpose so the bias path is exercised), interp
    vectors U(-1, 1)."""
    rng = np.random.RandomState(seed)
    out = []
    for name, shp in shapes.variable_table(cfg):
        if name.endswith("/bias"):
            val = rng.uniform(-0.1, 0.1, size=shp)
        elif len(shp) == 1:
            val = rng.uniform(-1.0, 1.0, size=shp)
        else:
            rf = int(np.prod(shp[:-2]))
            lim = math.sqrt(6.0 / (shp[-2] * rf + shp[-1] * rf))
            val = rng.uniform(-lim, lim, size=shp)
        out.append((name, val.astype(np.float32)))
    return out


_SMALL = dict(num_layers=3, num_initial_filters=8)

# name -> dict(cfg overrides on BASE_MODEL_CONFIG, batch, frames (desired output frames,
# or the exact length in same-padding mode), seed, training)
GOLDEN_CASES = {
    # reference named configs (Config.py:52-121) at reduced depth/width
    "baseline_small": dict(cfg=dict(_SMALL), batch=2, frames=64, seed=11, training=True),
    "baseline_diff_small": dict(cfg=dict(_SMALL, output_type="difference"), batch=2, frames=64,
                                seed=12, training=True),
    "baseline_context_small": dict(cfg=dict(_SMALL, output_type="difference", context=True),
                                   batch=2, frames=40, seed=13, training=True),
    "baseline_stereo_small": dict(cfg=dict(_SMALL, output_type="difference", context=True,
                                           mono_downmix=False), batch=2, frames=40, seed=14,
                                  training=True),
    "full_small": dict(cfg=dict(_SMALL, output_type="difference", context=True,
                                upsampling="learned", mono_downmix=False), batch=2, frames=40,
                       seed=15, training=True),
    "full_multi_small": dict(cfg=dict(_SMALL, output_type="difference", context=True,
                                      mono_downmix=False, task="multi_instrument"), batch=2,
                             frames=40, seed=16, training=True),
    "learned_same_small": dict(cfg=dict(_SMALL, upsampling="learned"), batch=2, frames=64,
                               seed=17, training=True),
    "linear_act_eval_small": dict(cfg=dict(_SMALL, output_activation="linear",
                                           output_type="difference", context=True),
                                  batch=2, frames=40, seed=18, training=False),
    "linear_act_direct_eval_small": dict(cfg=dict(_SMALL, output_activation="linear"), batch=1,
                                         frames=64, seed=19, training=False),
    "odd_filters_small": dict(cfg=dict(num_layers=4, num_initial_filters=6, filter_size=7,
                                       merge_filter_size=3, input_filter_size=7,
                                       output_filter_size=3, context=True,
                                       output_type="difference", mono_downmix=False,
                                       task="multi_instrument"), batch=1, frames=50, seed=20,
                              training=True),
    "odd_filters_same_small": dict(cfg=dict(num_layers=2, num_initial_filters=5, filter_size=4,
                                            merge_filter_size=2, input_filter_size=4,
                                            output_filter_size=2), batch=1, frames=32, seed=21,
                                   training=True),
    # input_filter_size != filter_size: get_padding sizes layer 0 with input_filter_size
    # (UnetAudioSeparator.py:73) but get_output convolves it with filter_size (:98), so the graph's
    # output length differs from get_padding's answer -- the reference's behaviour, reproduced as is
    "input_filter_mismatch_small": dict(cfg=dict(_SMALL, output_type="difference", context=True,
                                                 filter_size=7, input_filter_size=11, merge_filter_size=3),
                                        batch=2, frames=40, seed=22, training=True),
    # filter_size 1 with context: the transposed stride-2 conv's odd output phase has no taps
    "filter1_context_small": dict(cfg=dict(_SMALL, output_type="difference", context=True, filter_size=1,
                                           input_filter_size=1, merge_filter_size=3, mono_downmix=False),
                                  batch=2, frames=40, seed=23, training=True),
    # baseline_comparison (Config.py:123-134) at reduced depth: 34 initial filters -- layer widths 34, 68, 102, 136 are
    # NOT multiples of 8, so the register-window weight gradient and the DMA-staged conv tiles do not apply and the
    # LDS-tiled fall-back kernels carry the whole network
    "baseline_comparison_small": dict(cfg=dict(num_layers=3, num_initial_filters=34, output_type="difference",
                                               context=True), batch=2, frames=40, seed=24, training=True),
    # full-size M1 (Config.py:15-33), one excerpt
    "M1_full": dict(cfg=dict(), batch=1, frames=16384, seed=31, training=True),
    # full-size M1 architecture with context (BASELINE.json configs[1] shape), one excerpt
    "M1_context_full": dict(cfg=dict(context=True), batch=1, frames=16384, seed=32,
                            training=True),
}
