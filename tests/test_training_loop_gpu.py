"""Training-loop shape of the reference (Training.py:24-121) on the GPU: epoch_it steps, global_step
counting, checkpoint with TF variable names + Adam slots, and exact resume."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

import wave_u_net_amd as wun                        # noqa: E402
from wave_u_net_amd import training                 # noqa: E402


def _cfg(tmp, epoch_it):
    return wun.get_config("full", num_layers=3, num_initial_filters=8, num_frames=40, batch_size=4,
                          epoch_it=epoch_it, model_base_dir=os.path.join(tmp, "ckpt"),
                          log_dir=os.path.join(tmp, "logs"), init_sup_sep_lr=1e-3)


def test_train_checkpoint_and_exact_resume(tmp_path, monkeypatch):
    # Bit-exact resume needs identical tilings in every run: the autotuner (which may pick
    # different but equivalent tilings per process) is switched off for this check.
    monkeypatch.setenv("WUN_NO_TUNE", "1")
    tmp = str(tmp_path)
    p3 = training.train(_cfg(tmp, 3), "runA")                       # 3 steps, save
    assert os.path.basename(p3) == "runA-3.npz"                     # <id>-<global_step>, Training.py:113
    ck = np.load(p3)
    names = [k for k in ck.files if k.startswith("separator/")]
    assert "separator/conv1d/kernel" in names and "separator/interp_0" in names
    assert ck["separator/conv1d/kernel"].shape == (15, 2, 8)        # TF layout [K, Cin, Cout]
    assert int(ck["global_step"]) == 3 and ck["adam_m"].shape == ck["adam_v"].shape
    p5 = training.train(_cfg(tmp, 2), "runA", load_model=p3)        # resume for 2 more steps
    assert os.path.basename(p5) == "runA-5.npz"
    p5b = training.train(_cfg(tmp, 5), "runB")                      # uninterrupted 5 steps
    a, b = np.load(p5), np.load(p5b)
    for k in names + ["adam_m", "adam_v"]:
        assert np.array_equal(a[k], b[k]), k                        # deterministic kernels -> bit-exact resume
    # the loss went down on the fixed synthetic batch
    log = [eval(l.replace("NaN", "float('nan')")) for l in open(os.path.join(tmp, "logs", "runB", "train.jsonl"))]
    assert log[-1]["sep_loss"] < log[0]["sep_loss"]
    assert log[-1]["global_step"] == 5


def test_autotuned_runs_agree_to_rounding(tmp_path, monkeypatch):
    """With the autotuner on, two runs may use different tilings / split factors: the trained
    weights then agree to fp32 rounding, not bit for bit."""
    monkeypatch.delenv("WUN_NO_TUNE", raising=False)
    tmp = str(tmp_path)
    pa = training.train(_cfg(tmp, 5), "runT1")
    monkeypatch.setenv("WUN_NO_TUNE", "1")
    pb = training.train(_cfg(tmp, 5), "runT2")
    a, b = np.load(pa), np.load(pb)
    for k in a.files:
        if k.startswith("separator/"):
            assert np.abs(a[k] - b[k]).max() <= 2e-5, k


def test_tuning_table_export_import_roundtrip(tmp_path, monkeypatch):
    """wun_plan_tune_export / _import: a second trainer that imports the table runs the same tilings
    (bit-identical steps) without tuning; a table from another shape is rejected."""
    monkeypatch.delenv("WUN_NO_TUNE", raising=False)
    cache = os.path.join(str(tmp_path), "tune.txt")
    monkeypatch.setenv("WUN_TUNE_CACHE", cache)
    cfg = _cfg(str(tmp_path), 1)
    ta = training.Trainer(cfg)
    src = training.synthetic_source(cfg, ta.batch, ta.t_in, ta.t_out, ta.device)
    mix, targets = src()
    ta.tune(mix, targets)                                   # tunes and writes the table
    text = open(cache).read()
    assert text.startswith("wun-tune 2 order=") and " B=4 " in text and "\ncf " in text and "\nwg " in text and text.endswith("end\n")
    tb = training.Trainer(cfg)
    tb.tune(mix, targets)                                   # imports: no tuning pass
    assert tb.sep.tune_export() == text
    la = [float(ta.step(mix, targets).item()) for _ in range(3)]
    lb = [float(tb.step(mix, targets).item()) for _ in range(3)]
    assert la == lb
    assert torch.equal(ta.sep.params, tb.sep.params)
    other = training.Trainer(dict(cfg, num_frames=72))
    with pytest.raises(ValueError):
        other.sep.tune_import(text)
    # a truncated table (lost tail) and an entry out of range are refused ...
    with pytest.raises(ValueError):
        tb.sep.tune_import(text[:len(text) // 2])
    first_cf = next(l for l in text.splitlines() if l.startswith("cf "))
    with pytest.raises(ValueError):
        tb.sep.tune_import(text.replace(first_cf, "cf 9999 1", 1))
    # ... and entries that are in range but not legal for their launch (split far past the scratch) are ignored
    lines = text.splitlines()
    forged = "\n".join([l if not l.startswith("cf ") else "cf 0 48" for l in lines]) + "\n"
    tb.sep.tune_import(forged)                              # in range, but not legal choices for these launches:
    lc = [float(tb.step(mix, targets).item()) for _ in range(2)]      # ignored at launch, heuristics used instead
    assert all(np.isfinite(lc))


def test_tensorflow_format_checkpoint_resume_and_predict(tmp_path, monkeypatch):
    """model_config["checkpoint_format"] = "tf": train() saves "<dir>/<id>-<step>" as a TensorFlow V2 checkpoint with the
    reference's variable names (model, global_step, separator_solver/<var>/Adam{,_1}, beta powers: Training.py:66,75-77,
    98,113); resuming from it is bit-identical to resuming from the .npz, and the inference-side restore (variables only)
    accepts the prefix as `load_model`."""
    from wave_u_net_amd import tf_checkpoint, checkpoint
    monkeypatch.setenv("WUN_NO_TUNE", "1")
    tmp = str(tmp_path)
    p3 = training.train(dict(_cfg(tmp, 3), checkpoint_format="tf"), "runT")
    assert os.path.basename(p3) == "runT-3" and tf_checkpoint.is_checkpoint(p3)
    ck = tf_checkpoint.read(p3)
    assert ck["separator/conv1d/kernel"].shape == (15, 2, 8) and int(ck["global_step"]) == 3
    assert ck["global_step"].dtype == np.int64
    assert ck["separator_solver/separator/conv1d/kernel/Adam"].shape == (15, 2, 8)
    assert ck["separator_solver/separator/conv1d/kernel/Adam_1"].min() >= 0.0          # second moments
    assert np.isclose(float(ck["separator_solver/beta1_power"]), 0.9 ** 4)
    n_model = sum(1 for k in ck if k.startswith("separator/"))
    assert len(ck) == 3 * n_model + 3                                                   # model + 2 slots each + step + 2 powers
    p3n = training.train(_cfg(tmp, 3), "runN")                                          # same run, .npz
    a = np.load(p3n)
    for k in a.files:
        if k.startswith("separator/"):
            assert np.array_equal(a[k], ck[k]), k
    p5 = training.train(_cfg(tmp, 2), "runT", load_model=p3)                            # resume from the TF checkpoint
    p5n = training.train(_cfg(tmp, 2), "runN", load_model=p3n)
    x, y = np.load(p5), np.load(p5n)
    for k in x.files:
        assert np.array_equal(x[k], y[k]), k
    # inference-side restore (Evaluate.py:55-57): variables only
    sep = wun.UnetAudioSeparator(_cfg(tmp, 1))
    assert checkpoint.load_checkpoint(sep, p3, with_optimizer=False) == 3
    for k, v in sep.variables().items():
        assert np.array_equal(v.cpu().numpy(), ck[k]), k
    assert float(sep.adam_m.abs().max()) == 0.0
    # a checkpoint of another architecture is refused with the missing variable named
    other = wun.UnetAudioSeparator(dict(_cfg(tmp, 1), num_layers=4))
    with pytest.raises(KeyError, match="lacks"):
        checkpoint.load_checkpoint(other, p3)
