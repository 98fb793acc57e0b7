"""CPU checks of oracle/bf16_emul.py, the float64 restatement of the training step WITH the bf16 mode's storage roundings
(the sharp checker of tests/test_gpu_bf16.py::test_bf16_step_vs_storage_emulation_*).

(1) With the roundings switched off it is a third, hand-placed backward pass and must agree with the autograd oracle
    (oracle/waveunet_torch.py, float64) to rounding: that pins the graph walk it shares with the quantized form -- the
    decimated / skip-window split of a down level, the two-launch input gradient, the same-padding accumulate.
(2) With the roundings on, every tensor the plan keeps in HBM as bf16 is bf16-representable, and the gradients move away
    from the float64 oracle by the rounding noise of the mode (the figure tests/test_gpu_bf16.py tolerates: 1e-2 .. 1e-1),
    not by more."""
import numpy as np
import pytest
import torch

from oracle import bf16_emul, shapes, waveunet_torch as wt
from oracle.golden_params import GOLDEN_CASES, golden_params

CASES = ["baseline_small", "baseline_diff_small", "baseline_context_small", "baseline_stereo_small", "full_small",
         "full_multi_small", "learned_same_small", "odd_filters_small", "odd_filters_same_small",
         "input_filter_mismatch_small", "filter1_context_small"]


def _setup(name, batch=2):
    case = GOLDEN_CASES[name]
    cfg = shapes.finalize_config(dict(shapes.BASE_MODEL_CONFIG, **case["cfg"]))
    params = golden_params(cfg, case["seed"])
    i, o = shapes.get_padding(cfg, [batch, case["frames"], 0])
    t_out = None
    # (input_filter_size != filter_size: the graph's output length differs from get_padding's answer)
    tp = wt.params_to_torch(params, torch.float64)
    probe = wt.get_output(cfg, tp, torch.zeros(1, i[1], cfg["num_channels"], dtype=torch.float64), True)
    t_out = next(iter(probe.values())).shape[1]
    mix, targets = wt.synthetic_batch(cfg, batch, i[1], t_out if (i[1] - t_out) % 2 == 0 else o[1], seed=case["seed"] + 7)
    if next(iter(targets.values())).shape[1] != t_out:
        rng = np.random.default_rng(case["seed"])
        targets = {k: rng.uniform(-0.3, 0.3, (batch, t_out, cfg["num_channels"])).astype(np.float32) for k in cfg["source_names"]}
    return cfg, params, mix, targets


@pytest.mark.parametrize("name", CASES)
def test_unquantized_emulation_equals_the_autograd_oracle(name):
    cfg, params, mix, targets = _setup(name)
    loss, grads, _ = bf16_emul.train_step(cfg, params, mix, targets, quantize=False)
    oloss, ograds, _ = wt.chunked_train_step(cfg, params, mix, targets, dtype=torch.float64, chunk=mix.shape[0])
    assert abs(loss - oloss) <= 1e-12 * max(1.0, abs(oloss))
    for (n, _), g, og in zip(params, grads, ograds):
        assert g.shape == og.shape, n
        err = (g - og).abs().max().item()
        assert err <= 1e-10 * max(og.abs().max().item(), 1e-30) + 1e-18, (n, err)


def _is_bf16(t):
    return torch.equal(t, t.to(torch.float32).to(torch.bfloat16).to(torch.float64))


@pytest.mark.parametrize("name", ["baseline_small", "baseline_stereo_small", "full_small", "learned_same_small"])
def test_quantized_emulation_rounds_what_the_plan_stores(name):
    cfg, params, mix, targets = _setup(name)
    loss, grads, inter = bf16_emul.train_step(cfg, params, mix, targets)
    for k, v in inter.items():
        if k not in ("outputs", "_scale"):
            assert _is_bf16(v), k
    oloss, ograds, _ = wt.chunked_train_step(cfg, params, mix, targets, dtype=torch.float64, chunk=mix.shape[0])
    assert abs(loss - oloss) <= 2e-2 * abs(oloss)
    worst = 0.0
    for (n, _), g, og in zip(params, grads, ograds):
        if n.endswith("/kernel"):
            worst = max(worst, (g - og).norm().item() / max(og.norm().item(), 1e-30))
    assert 1e-4 < worst < 0.15, worst         # the mode's rounding noise: present, and of the size the GPU tests tolerate


def test_head_on_mfma_rule():
    base = dict(shapes.BASE_MODEL_CONFIG)
    assert not bf16_emul.head_on_mfma(dict(base, mono_downmix=False, output_type="difference", context=True))       # M4: 26 x 2
    assert not bf16_emul.head_on_mfma(dict(base, mono_downmix=False, task="multi_instrument", output_type="difference"))  # 26 x 6
    assert bf16_emul.head_on_mfma(dict(base, num_layers=16, num_initial_filters=48, mono_downmix=False,
                                       task="multi_instrument", output_type="difference"))                           # deep: 50 x 6


@pytest.mark.parametrize("name", ["baseline_stereo_small", "learned_same_small", "full_small"])
def test_layerwise_mode_is_a_fixed_point_on_its_own_tensors(name):
    """forced = the chained run's own stored tensors: every recomputed tensor, the loss and every gradient are identical
    (what the GPU test relies on: a difference in layer-by-layer mode is a difference in ONE launch)."""
    cfg, params, mix, targets = _setup(name)
    loss, grads, inter = bf16_emul.train_step(cfg, params, mix, targets)
    forced = {k: v.clone() for k, v in inter.items() if k not in ("outputs", "_scale")}
    expect = {"bottleneck", "dz_bottleneck"}
    for i in range(cfg["num_layers"]):
        expect |= {"dec%d" % i, "skip%d" % i, "dz_skip%d" % i, "ups%d" % i, "up%d" % i, "dz_up%d" % i, "d_ups%d" % i}
        if cfg["context"]:
            expect.add("dz_dec%d" % i)
    assert set(forced) == expect
    loss2, grads2, inter2 = bf16_emul.train_step(cfg, params, mix, targets, forced=forced)
    assert loss2 == loss
    for k in forced:
        assert torch.equal(inter2[k], inter[k]), k
    for a, b in zip(grads, grads2):
        assert torch.equal(a, b)
    # ... and a perturbed input tensor moves only what reads it
    forced["dec0"] = forced["dec0"] * 1.5
    _, _, inter3 = bf16_emul.train_step(cfg, params, mix, targets, forced=forced)
    assert torch.equal(inter3["dec0"], inter["dec0"])                 # computed from the audio, which did not change
    assert not torch.equal(inter3["dec1"], inter["dec1"])             # reads dec0
    assert torch.equal(inter3["bottleneck"], inter["bottleneck"])     # reads the (given) dec2


def test_wide_head_rounds_only_the_heads_weight_gradient_operands():
    """(C + F) * Sh * C > 256 (the deep variant's head: stereo, 48 filters, 3 trained sources): the plan runs the head's weight
    gradient on the bf16 MFMA kernel, on bf16 copies of the audio and of d(pre-activation) -- everything else of the step is
    the narrow-head computation, so the two emulations may differ in the head kernels' gradients only, and there by bf16
    rounding of the operands (a few 1e-3), not more."""
    over = dict(num_layers=2, num_initial_filters=48, mono_downmix=False, task="multi_instrument", output_type="difference")
    cfg = shapes.finalize_config(dict(shapes.BASE_MODEL_CONFIG, **over))
    assert bf16_emul.head_on_mfma(cfg)
    params = golden_params(cfg, 7)
    mix, targets = wt.synthetic_batch(cfg, 2, 256, 256, seed=8)
    loss_w, grads_w, inter_w = bf16_emul.train_step(cfg, params, mix, targets)
    orig = bf16_emul.head_on_mfma
    bf16_emul.head_on_mfma = lambda c: False
    try:
        loss_n, grads_n, inter_n = bf16_emul.train_step(cfg, params, mix, targets)
    finally:
        bf16_emul.head_on_mfma = orig
    assert loss_w == loss_n
    for k in inter_w:
        if k not in ("outputs", "_scale"):
            assert torch.equal(inter_w[k], inter_n[k]), k
    nvar = len(params)
    head = set(range(nvar - 6, nvar))                                   # three head convs: kernel + bias each
    for i, ((n, _), a, b) in enumerate(zip(params, grads_w, grads_n)):
        if i in head:
            rel = (a - b).norm().item() / max(b.norm().item(), 1e-30)
            assert rel < 2e-2, (n, rel)
            if n.endswith("/kernel"):
                assert rel > 1e-5, (n, rel)                             # ... and the rounding IS there
        else:
            assert torch.equal(a, b), n
