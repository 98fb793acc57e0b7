"""GPU parity tests (run with -m gpu on an MI355X): the HIP path, called through the C ABI
(libwun.so via wave_u_net_amd), against the oracle on the same seeded inputs and against
the committed goldens generated from the reference's own graph code.

Tolerances (fp32 path, stated per BASELINE.json north_star "within a stated fp32 tolerance";
every comparison also LOGS the error it observed -- tests/_observed.py -- and the tolerances
below are kept within ~10x of those observations):
single ops OP_TOL * max|ref|; network outputs OUT_TOL absolute (outputs are O(1)); loss LOSS_TOL
relative; gradients GRAD_TOL * max|grad tensor| + 1e-7 per tensor against the FLOAT64 oracle at
every size, including full size (<= 1e-3)."""
import ctypes as C
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import shapes, waveunet_torch as wt
from oracle.golden_params import GOLDEN_CASES, golden_params
from _observed import record

pytestmark = pytest.mark.gpu

# observed on MI355X (gpurun_out/parity_observed.json, round 2): ops <= 1.5e-6, outputs <= 4.8e-7,
# loss <= 1.3e-7, gradients <= 7.1e-5 (the benchmarked B=16 configuration) -- tolerances ~7-13x that
OP_TOL = 2e-5        # x max|ref|
OUT_TOL = 5e-6       # absolute, outputs are O(1)
LOSS_TOL = 2e-6      # relative
GRAD_TOL = 5e-4      # x max|g| per tensor, vs the float64 oracle
GRAD_TOL_FULL_TUNED = 1e-3   # the B = 16 headline plan under its tuned table: max-norm on the LeakyReLU sign floor (see the test)
GRAD_L2_TOL = 2.6e-4 # per conv kernel ||g - ref||_2 / ||ref||_2 at full size vs the FREE float64 oracle: 2x the observed 1.0e-4 .. 1.3e-4
                     # (LeakyReLU branch flips included; ~1e-6 without)
GRAD_TOL_PINNED = 2e-5   # x max|g| per tensor vs the float64 oracle evaluated on the kernels' own LeakyReLU branches (_gpu_pins)

import wave_u_net_amd as wun                      # noqa: E402
from wave_u_net_amd import _lib                   # noqa: E402
from wave_u_net_amd.separator import UnetAudioSeparator   # noqa: E402


@pytest.fixture(scope="module")
def lib():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return _lib.load()


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _cuda(a):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).cuda()


def test_native_library_is_loaded(lib):
    maps = open("/proc/self/maps").read()
    assert "libwun.so" in maps
    assert "gfx950" in torch.cuda.get_device_properties(0).gcnArchName


def test_mfma_lane_layout(lib):
    rng = np.random.default_rng(0)
    a = rng.standard_normal((16, 4)).astype(np.float32)      # asymmetric operands catch transposes
    b = rng.standard_normal((4, 16)).astype(np.float32)
    da, db = _cuda(a), _cuda(b)
    dd = torch.zeros(16, 16, device="cuda")
    _lib.check(lib.wun_op_mfma_probe(da.data_ptr(), db.data_ptr(), dd.data_ptr(), _stream()))
    torch.cuda.synchronize()
    ref = a.astype(np.float64) @ b.astype(np.float64)
    assert np.abs(dd.cpu().numpy() - ref).max() < 1e-5


CONV_CASES = [
    # (B, Cin, Cout, K, T_in, stride, pad_left, same)
    (2, 24, 48, 15, 700, 1, 0, False),
    (2, 24, 48, 15, 701, 2, 0, False),
    (1, 48, 72, 15, 1100, 2, 0, False),
    (2, 1, 24, 15, 1000, 1, 0, False),
    (2, 1, 24, 15, 1001, 2, 0, False),
    (2, 2, 24, 15, 600, 2, 0, False),
    (1, 72, 24, 5, 500, 1, 0, False),
    (2, 40, 24, 5, 300, 1, 2, True),
    (2, 24, 48, 15, 512, 1, 7, True),
    (3, 288, 312, 15, 23, 1, 0, False),
    (2, 264, 288, 15, 59, 2, 0, False),
    (2, 600, 288, 5, 17, 1, 0, False),
    (2, 13, 7, 4, 90, 1, 1, True),
    (2, 13, 7, 7, 91, 2, 0, False),
    (1, 6, 5, 3, 40, 1, 0, False),
    (2, 120, 144, 15, 200, 1, 0, False),
    (2, 168, 176, 15, 100, 1, 0, False),
    (1, 26, 2, 1, 333, 1, 0, False),
    (1, 96, 120, 15, 260, 2, 0, False),
    # few output positions per excerpt, several excerpts: batch-folded (FOLD) tiles
    (16, 40, 72, 15, 39, 1, 0, False),
    (16, 48, 56, 15, 95, 2, 0, False),
    (5, 24, 48, 5, 20, 1, 2, True),
    (16, 96, 96, 15, 110, 1, 0, False),
    (7, 64, 40, 15, 33, 2, 0, False),
]


def _conv_ref(x, w, bias, stride, pad_left, t_out, lrelu):
    K = w.shape[0]
    xx = torch.as_tensor(x, dtype=torch.float64)
    need = (t_out - 1) * stride + K - pad_left
    pad_r = max(0, need - xx.shape[2])
    xx = F.pad(xx, (pad_left, pad_r))
    y = F.conv1d(xx, torch.as_tensor(w, dtype=torch.float64).permute(2, 1, 0),
                 torch.as_tensor(bias, dtype=torch.float64), stride=stride)[:, :, :t_out]
    if lrelu:
        y = torch.maximum(0.2 * y, y)
    return y.numpy()


def _t_out(T, K, stride, same):
    if same:
        return T
    return (T - K) // stride + 1


@pytest.mark.parametrize("case", CONV_CASES, ids=[str(c) for c in CONV_CASES])
def test_op_conv1d_forward(lib, case):
    B, Cin, Cout, K, T, stride, pad, same = case
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31))
    x = rng.uniform(-1, 1, (B, Cin, T)).astype(np.float32)
    w = (rng.uniform(-1, 1, (K, Cin, Cout)) / np.sqrt(K * Cin)).astype(np.float32)
    b = rng.uniform(-0.1, 0.1, Cout).astype(np.float32)
    t_out = _t_out(T, K, stride, same)
    y = torch.full((B, Cout, t_out), float("nan"), device="cuda")
    dx, dw, db_ = _cuda(x), _cuda(w), _cuda(b)
    _lib.check(lib.wun_op_conv1d(dx.data_ptr(), dw.data_ptr(), db_.data_ptr(), y.data_ptr(), B, Cin, Cout,
                                 K, T, t_out, stride, pad, 1, _stream()))
    torch.cuda.synchronize()
    ref = _conv_ref(x, w, b, stride, pad, t_out, True)
    got = y.cpu().numpy()
    assert np.isfinite(got).all()
    err = np.abs(got - ref).max() / max(1.0, np.abs(ref).max())
    record("op_conv1d_forward", str(case), err, OP_TOL)
    assert err <= OP_TOL


FOLD_CASES = [
    (16, 40, 72, 15, 23, 1, 0, False),      # T_out 9: every excerpt in one tile
    (16, 48, 96, 15, 95, 2, 0, False),      # stride 2, T_out 41
    (6, 72, 48, 5, 77, 1, 2, True),         # 'same' padding, T_out 77
    (16, 24, 64, 15, 151, 1, 0, False),     # T_out 137: two excerpts per 128-row tile
    (16, 32, 128, 7, 19, 2, 0, False),      # T_out 7 < 9
]


@pytest.mark.gpu
@pytest.mark.parametrize("case", FOLD_CASES, ids=[str(c) for c in FOLD_CASES])
def test_op_conv1d_every_fold_variant(lib, case):
    """Forward conv + input gradient through every batch-folded tile variant (34..41) and split-K
    factor, against the float64 reference."""
    B, Cin, Cout, K, T, stride, pad, same = case
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31) + 7)
    x = rng.uniform(-1, 1, (B, Cin, T)).astype(np.float32)
    w = (rng.uniform(-1, 1, (K, Cin, Cout)) / np.sqrt(K * Cin)).astype(np.float32)
    b = rng.uniform(-0.1, 0.1, Cout).astype(np.float32)
    t_out = _t_out(T, K, stride, same)
    dz = rng.uniform(-1, 1, (B, Cout, t_out)).astype(np.float32)
    ref = _conv_ref(x, w, b, stride, pad, t_out, True)
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    need = (t_out - 1) * stride + K - pad
    xp = F.pad(xt, (pad, max(0, need - T)))
    yy = F.conv1d(xp, torch.tensor(w, dtype=torch.float64).permute(2, 1, 0), None, stride=stride)[:, :, :t_out]
    (yy * torch.tensor(dz, dtype=torch.float64)).sum().backward()
    ref_dx = xt.grad.numpy()
    dx, dw, db_, dzg = _cuda(x), _cuda(w), _cuda(b), _cuda(dz)
    wts = torch.empty(2 * K * Cin * Cout, device="cuda")
    ran = 0
    try:
        for variant in range(34, 42):
            for ks in (1, 3):
                lib.wun_op_force_conv_variant(variant, ks)
                y = torch.full((B, Cout, t_out), float("nan"), device="cuda")
                rc = lib.wun_op_conv1d(dx.data_ptr(), dw.data_ptr(), db_.data_ptr(), y.data_ptr(), B, Cin, Cout,
                                       K, T, t_out, stride, pad, 1, _stream())
                if rc != 0:
                    continue               # segments do not fit this tile's LDS row: rejected, not miscomputed
                torch.cuda.synchronize()
                got = y.cpu().numpy()
                assert np.isfinite(got).all(), (variant, ks)
                assert np.abs(got - ref).max() <= OP_TOL * max(1.0, np.abs(ref).max()), (variant, ks)
                ran += 1
                gdx = torch.full((B, Cin, T), float("nan"), device="cuda")
                rc = lib.wun_op_conv1d_dgrad(dzg.data_ptr(), dw.data_ptr(), gdx.data_ptr(), wts.data_ptr(), B, Cin,
                                             Cout, K, T, t_out, stride, pad, _stream())
                if rc != 0:
                    continue               # not a legal choice for the transposed shape (N = Cin)
                torch.cuda.synchronize()
                gd = gdx.cpu().numpy()
                assert np.isfinite(gd).all(), (variant, ks)
                assert np.abs(gd - ref_dx).max() <= OP_TOL * max(1.0, np.abs(ref_dx).max()), (variant, ks)
    finally:
        lib.wun_op_force_conv_variant(-1, 0)
    assert ran >= 4


@pytest.mark.parametrize("case", CONV_CASES, ids=[str(c) for c in CONV_CASES])
def test_op_conv1d_wgrad_and_dgrad(lib, case):
    B, Cin, Cout, K, T, stride, pad, same = case
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31) + 1)
    x = rng.uniform(-1, 1, (B, Cin, T)).astype(np.float32)
    w = (rng.uniform(-1, 1, (K, Cin, Cout)) / np.sqrt(K * Cin)).astype(np.float32)
    t_out = _t_out(T, K, stride, same)
    dz = rng.uniform(-1, 1, (B, Cout, t_out)).astype(np.float32)
    # float64 reference through autograd
    xt = torch.tensor(x, dtype=torch.float64, requires_grad=True)
    wtn = torch.tensor(w, dtype=torch.float64, requires_grad=True)
    need = (t_out - 1) * stride + K - pad
    xp = F.pad(xt, (pad, max(0, need - T)))
    y = F.conv1d(xp, wtn.permute(2, 1, 0), None, stride=stride)[:, :, :t_out]
    (y * torch.tensor(dz, dtype=torch.float64)).sum().backward()
    ref_dw, ref_dx = wtn.grad.numpy(), xt.grad.numpy()
    ref_db = dz.astype(np.float64).sum(axis=(0, 2))

    dxg, dwg, dzg = _cuda(x), _cuda(w), _cuda(dz)
    n_scr = lib.wun_op_conv1d_wgrad_scratch(B, Cin, Cout, K, t_out)
    scr = torch.empty(int(n_scr), device="cuda")
    gdw = torch.full((K, Cin, Cout), float("nan"), device="cuda")
    gdb = torch.full((Cout,), float("nan"), device="cuda")
    _lib.check(lib.wun_op_conv1d_wgrad(dxg.data_ptr(), dzg.data_ptr(), gdw.data_ptr(), gdb.data_ptr(),
                                       scr.data_ptr(), B, Cin, Cout, K, T, t_out, stride, pad, _stream()))
    torch.cuda.synchronize()
    ew = np.abs(gdw.cpu().numpy() - ref_dw).max() / max(1.0, np.abs(ref_dw).max())
    eb = np.abs(gdb.cpu().numpy() - ref_db).max() / max(1.0, np.abs(ref_db).max())
    record("op_conv1d_wgrad", str(case), max(ew, eb), OP_TOL)
    assert ew <= OP_TOL and eb <= OP_TOL

    if stride == 2 and pad != 0:
        return
    wts = torch.empty(2 * K * Cin * Cout, device="cuda")
    gdx = torch.full((B, Cin, T), float("nan"), device="cuda")
    _lib.check(lib.wun_op_conv1d_dgrad(dzg.data_ptr(), dwg.data_ptr(), gdx.data_ptr(), wts.data_ptr(), B, Cin,
                                       Cout, K, T, t_out, stride, pad, _stream()))
    torch.cuda.synchronize()
    got = gdx.cpu().numpy()
    assert np.isfinite(got).all()
    err = np.abs(got - ref_dx).max() / max(1.0, np.abs(ref_dx).max())
    record("op_conv1d_dgrad", str(case), err, OP_TOL)
    assert err <= OP_TOL


def _ocfg(case):
    return shapes.finalize_config(dict(shapes.BASE_MODEL_CONFIG, **case["cfg"]))


def _make_sep(case, params):
    cfg = wun.get_config("baseline", **case["cfg"])
    sep = UnetAudioSeparator(cfg, device="cuda:0")
    return sep, cfg


@pytest.mark.parametrize("name", sorted(GOLDEN_CASES))
def test_forward_matches_reference_goldens(lib, name, golden_dir):
    """HIP get_output vs outputs of the reference's own graph code (tests/golden)."""
    case = GOLDEN_CASES[name]
    ocfg = _ocfg(case)
    g = np.load(os.path.join(golden_dir, "fwd_%s.npz" % name))
    params = golden_params(ocfg, case["seed"])
    sep, cfg = _make_sep(case, params)
    mix = torch.from_numpy(g["mix"]).cuda()
    sep._plan(mix.shape[0], mix.shape[1])
    sep._active = sep._plans[(mix.shape[0], mix.shape[1])]
    sep.load_variables(params)
    outs = sep.get_output(mix, case["training"])
    torch.cuda.synchronize()
    assert list(outs.keys()) == ocfg["source_names"]
    for n in ocfg["source_names"]:
        got = outs[n].cpu().numpy()
        ref = g["out_" + n]
        assert got.shape == ref.shape
        assert np.isfinite(got).all(), n
        record("forward_vs_reference_goldens", "%s/%s" % (name, n), np.abs(got - ref).max(), OUT_TOL)
        assert np.abs(got - ref).max() <= OUT_TOL, (n, np.abs(got - ref).max())


STEP_CASES = ["baseline_small", "baseline_diff_small", "baseline_context_small", "baseline_stereo_small",
              "full_small", "full_multi_small", "learned_same_small", "odd_filters_small",
              "odd_filters_same_small", "input_filter_mismatch_small", "filter1_context_small",
              "baseline_comparison_small"]


def _grad_check(sep, tp, ograds, tol=GRAD_TOL, tag="?"):
    """Every gradient tensor against the (float64) oracle: max|got - ref| <= tol * max|ref| + 1e-7."""
    g = sep.gradients()
    worst = []
    for (n, _), og in zip(tp, ograds):
        got = g[n].cpu().double()
        og = og.double()
        scale = max(og.abs().max().item(), 1e-30)
        err = (got - og).abs().max().item()
        worst.append((err / scale, n, err, scale))
        assert torch.isfinite(got).all(), n
    worst.sort(reverse=True)
    record("gradients_vs_float64_oracle", "%s (worst: %s)" % (tag, worst[0][1]), worst[0][0], tol)
    bad = [w for w in worst if w[2] > tol * w[3] + 1e-7]
    assert not bad, bad[:5]
    return worst[0][0]


def _grad_rel_l2(sep, tp, ograds):
    """worst per-tensor ||got - ref||_2 / ||ref||_2 over the conv kernels (the statistic a single LeakyReLU mask flip
    barely moves: it perturbs a few hundred of a tensor's 10^4 .. 10^6 elements)."""
    g = sep.gradients()
    w = (0.0, "")
    for (n, _), og in zip(tp, ograds):
        if not n.endswith("/kernel"):
            continue
        got = g[n].cpu().double(); og = og.double()
        e = ((got - og).norm() / max(og.norm().item(), 1e-30)).item()
        if e > w[0]:
            w = (e, n)
    return w


def _gpu_pins(sep, ocfg):
    """The LeakyReLU branch every conv output of the last get_output(training=True) fell on, read back from the
    workspace (wun_plan_activation), as `pins` for the oracle (oracle/waveunet_torch.py: leaky_relu): name -> (pos, known)
    bool tensors of the layer's full conv-output shape.  Post-activation values keep the pre-activation's sign, so
    pos = (stored value > 0).  A down level keeps two tensors -- the decimated stream (even positions, what the next
    level reads) and the skip window (what crop_and_concat reads).  Since round 6 the decimated stream IS a slice of the
    encoder output, as in the reference (UnetAudioSeparator.py:98-100): the even window positions are written once, by the
    stride-2 launch, into both tensors -- asserted here bit for bit -- so ONE pin per level, "down<i>", covers both (rounds
    1 - 5 computed them twice with different summation orders and needed a separate "down<i>/dec" pin).  Positions the
    kernels never compute (odd outputs outside the crop window, context mode: dead work) stay unpinned."""
    L, same = ocfg["num_layers"], not ocfg["context"]
    Kd = ocfg["filter_size"]
    B = int(sep._active.info.batch)
    t = int(sep._active.info.input_frames)
    assert sep.effective_dtype == "f32"
    pins = {}

    def place(shape, view, t0, tstep):
        pos = torch.zeros(shape, dtype=torch.bool)
        known = torch.zeros(shape, dtype=torch.bool)
        v = view.cpu()
        n = v.shape[2]
        pos[:, :, t0:t0 + n * tstep:tstep] = v > 0
        known[:, :, t0:t0 + n * tstep:tstep] = True
        return pos, known

    for i in range(L):
        t_conv = t if same else t - Kd + 1
        v, t0, ts = sep.activation("skip", i)
        shape = (B, v.shape[1], t_conv)
        pos, known = place(shape, v, t0, ts)
        if not same:
            vd, t0d, tsd = sep.activation("dec", i)
            assert (t0d, tsd) == (0, 2) and vd.shape[2] == (t_conv + 1) // 2 and ts == 1
            # the even absolute positions of the window are the decimated stream's elements: same bits
            e0 = t0 + (t0 & 1)
            n_even = len(range(e0, t0 + v.shape[2], 2))
            if n_even:
                assert torch.equal(v[:, :, e0 - t0::2].cpu(), vd[:, :, e0 // 2:e0 // 2 + n_even].cpu()), \
                    "down level %d: skip window and decimated stream disagree at the shared positions" % i
            pd, kd = place(shape, vd, t0d, tsd)
            pos, known = torch.where(known, pos, pd), known | kd
        pins["down%d" % i] = (pos, known)
        t = (t_conv + 1) // 2
    v, t0, ts = sep.activation("bottleneck")
    pins["bottleneck"] = place(tuple(v.shape), v, t0, ts)
    for j in range(L):
        v, t0, ts = sep.activation("up", j)
        pins["up%d" % j] = place(tuple(v.shape), v, t0, ts)
    return pins


def _count_flips(ocfg, params, mix, pins):
    """(number of pinned LeakyReLU inputs whose float64 value lies on the other side of 0 than the kernels' fp32 value,
    number of pinned inputs) -- evaluated excerpt by excerpt with the FREE float64 oracle."""
    tp = wt.params_to_torch(params, torch.float64, requires_grad=False)
    flips = total = 0
    with torch.no_grad():
        for b in range(mix.shape[0]):
            _, inter = wt.get_output(ocfg, tp, torch.as_tensor(mix[b:b + 1]).double(), True, return_intermediates=True)
            for name, (pos, known) in pins.items():
                y = inter[name.split("/")[0]]
                k = known[b:b + 1]
                flips += int(((y > 0) != pos[b:b + 1])[k].sum())
                total += int(k.sum())
    return flips, total


def _loss_check(loss, oloss, tag):
    rel = abs(float(loss) - float(oloss)) / max(abs(float(oloss)), 1e-3)
    record("loss_vs_float64_oracle", tag, rel, LOSS_TOL)
    assert rel <= LOSS_TOL, (float(loss), float(oloss))


def _out_check(outs, oouts, names, tag):
    worst = max((outs[n].cpu().double() - oouts[n].detach().double()).abs().max().item() for n in names)
    record("outputs_vs_float64_oracle", tag, worst, OUT_TOL)
    assert worst <= OUT_TOL, worst


def _oracle64(ocfg, params, mix, targets, chunk=1):
    """float64 oracle of a whole batch, one chunk of excerpts at a time (bounded host memory)."""
    loss, grads, outs = wt.chunked_train_step(ocfg, params, mix, targets, dtype=torch.float64, chunk=chunk,
                                              want_outputs=True)
    tp = [(n, None) for n, _ in params]
    return loss, grads, outs, tp


@pytest.mark.parametrize("name", STEP_CASES)
def test_train_step_matches_oracle(lib, name):
    """loss, every variable's gradient and one TF-Adam update vs the float64 oracle."""
    case = GOLDEN_CASES[name]
    ocfg = _ocfg(case)
    params = golden_params(ocfg, case["seed"])
    sep, cfg = _make_sep(case, params)
    B = 3
    i, o = shapes.get_padding(ocfg, [B, case["frames"], 0])
    mix, targets = wt.synthetic_batch(ocfg, B, i[1], o[1], seed=case["seed"] + 100)
    sep._plan(B, i[1]); sep._active = sep._plans[(B, i[1])]
    sep.load_variables(params)
    outs = sep.get_output(torch.from_numpy(mix).cuda(), True)
    loss = sep.loss_and_gradients({k: torch.from_numpy(v) for k, v in targets.items()})
    torch.cuda.synchronize()

    tp = wt.params_to_torch(params, torch.float64, requires_grad=True)
    tmix = torch.tensor(mix, dtype=torch.float64)
    ttg = {k: torch.tensor(v, dtype=torch.float64) for k, v in targets.items()}
    oloss, ograds = wt.train_step(ocfg, tp, tmix, ttg)
    oouts = wt.get_output(ocfg, tp, tmix, True)
    _out_check(outs, oouts, ocfg["source_names"], name)
    _loss_check(loss.item(), oloss.item(), name)
    _grad_check(sep, tp, ograds, tag=name)

    # one TF-rule Adam update: the kernel vs the oracle's tf_adam_step fed with the SAME (GPU)
    # gradients, so this isolates the optimizer arithmetic from gradient round-off
    g = sep.gradients()
    gp = [g[n].cpu().double() for n, _ in tp]
    pp = [p.detach().clone() for _, p in tp]
    m = [torch.zeros_like(p) for p in pp]
    v = [torch.zeros_like(p) for p in pp]
    wt.tf_adam_step(pp, gp, m, v, 1, 1e-3)
    sep.adam_step(1e-3)
    torch.cuda.synchronize()
    var = sep.variables()
    for (n, _), p in zip(tp, pp):
        assert (var[n].cpu().double() - p).abs().max().item() <= 2e-6, n
    # second step exercises the moment buffers
    wt.tf_adam_step(pp, gp, m, v, 2, 1e-3)
    sep.adam_step(1e-3)
    torch.cuda.synchronize()
    var = sep.variables()
    for (n, _), p in zip(tp, pp):
        assert (var[n].cpu().double() - p).abs().max().item() <= 4e-6, n


def test_full_size_m1_context_step_vs_oracle(lib):
    """BASELINE.json configs[1] architecture at B=2 (the oracle's fp32 backward of one
    147443-sample excerpt takes seconds): loss + all 54 gradients."""
    ocfg = shapes.finalize_config(dict(shapes.BASE_MODEL_CONFIG, context=True))
    params = golden_params(ocfg, 77)
    sep = UnetAudioSeparator(wun.get_config("m1_context"), device="cuda:0")
    B = 2
    i, o = shapes.get_padding(ocfg, [B, 16384, 0])
    mix, targets = wt.synthetic_batch(ocfg, B, i[1], o[1], seed=78)
    sep._plan(B, i[1]); sep._active = sep._plans[(B, i[1])]
    sep.load_variables(params)
    outs = sep.get_output(torch.from_numpy(mix).cuda(), True)
    loss = sep.loss_and_gradients({k: torch.from_numpy(v) for k, v in targets.items()})
    torch.cuda.synchronize()
    oloss, ograds, oouts, tp = _oracle64(ocfg, params, mix, targets)
    _out_check(outs, oouts, ocfg["source_names"], "M1_context_full_B2")
    _loss_check(loss.item(), oloss, "M1_context_full_B2")
    _grad_check(sep, tp, ograds, tag="M1_context_full_B2")


def test_full_size_m1_same_step_vs_oracle(lib):
    """BASELINE.json configs[0]: M1 as shipped (same padding, 16384 samples), B=2."""
    ocfg = shapes.finalize_config(dict(shapes.BASE_MODEL_CONFIG))
    params = golden_params(ocfg, 79)
    sep = UnetAudioSeparator(wun.get_config("baseline"), device="cuda:0")
    B = 2
    mix, targets = wt.synthetic_batch(ocfg, B, 16384, 16384, seed=80)
    sep._plan(B, 16384); sep._active = sep._plans[(B, 16384)]
    sep.load_variables(params)
    outs = sep.get_output(torch.from_numpy(mix).cuda(), True)
    loss = sep.loss_and_gradients({k: torch.from_numpy(v) for k, v in targets.items()})
    torch.cuda.synchronize()
    oloss, ograds, oouts, tp = _oracle64(ocfg, params, mix, targets)
    _out_check(outs, oouts, ocfg["source_names"], "M1_same_full_B2")
    _loss_check(loss.item(), oloss, "M1_same_full_B2")
    _grad_check(sep, tp, ograds, tag="M1_same_full_B2")


def test_full_batch_properties_m1_context(lib):
    """At BASELINE.json's full size (B=16, 147443 -> 16389) the oracle is too slow for a
    per-element check, so use size-independent properties: (1) determinism -- two runs are
    bit-identical; (2) batch independence -- excerpt b of a B=16 run equals the same excerpt
    run alone (to fp32 rounding: the split-K depth of the deep levels depends on the batch);
    (3) the B=16 gradient is the mean of per-excerpt gradients."""
    ocfg = shapes.finalize_config(dict(shapes.BASE_MODEL_CONFIG, context=True))
    params = golden_params(ocfg, 81)
    sep = UnetAudioSeparator(wun.get_config("m1_context"), device="cuda:0")
    B = 16
    mix, targets = wt.synthetic_batch(ocfg, B, 147443, 16389, seed=82)
    dmix = torch.from_numpy(mix).cuda()
    tg = torch.stack([torch.from_numpy(targets[n]) for n in ocfg["source_names"]]).cuda()
    sep._plan(B, 147443); sep._active = sep._plans[(B, 147443)]
    sep.load_variables(params)
    o1 = torch.stack(list(sep.get_output(dmix, True).values())).clone()
    l1 = sep.loss_and_gradients(tg).clone()
    g1 = sep.grads.clone()
    o2 = torch.stack(list(sep.get_output(dmix, True).values())).clone()
    l2 = sep.loss_and_gradients(tg).clone()
    g2 = sep.grads.clone()
    torch.cuda.synchronize()
    assert torch.equal(o1, o2) and torch.equal(l1, l2) and torch.equal(g1, g2)
    assert torch.isfinite(g1).all() and torch.isfinite(o1).all()
    gsum = torch.zeros_like(g1)
    for b in (0, 7, 15):
        ob = torch.stack(list(sep.get_output(dmix[b:b + 1], True).values()))
        assert (ob[:, 0] - o1[:, b]).abs().max().item() <= 2e-6, b
    for b in range(B):
        sep.get_output(dmix[b:b + 1], True)
        sep.loss_and_gradients(tg[:, b:b + 1])
        gsum += sep.grads
    torch.cuda.synchronize()
    gmean = gsum / B
    scale = g1.abs().max().item()
    assert (gmean - g1).abs().max().item() <= 1e-4 * scale
    # 'training=False' clips only for the linear activation; tanh outputs stay in (-1, 1)
    assert o1.abs().max().item() < 1.0


def test_inference_mode_clip(lib):
    case = GOLDEN_CASES["linear_act_eval_small"]
    ocfg = _ocfg(case)
    params = golden_params(ocfg, case["seed"])
    sep, cfg = _make_sep(case, params)
    i, o = shapes.get_padding(ocfg, [2, case["frames"], 0])
    mix = np.random.default_rng(5).uniform(-3, 3, (2, i[1], 1)).astype(np.float32)
    sep._plan(2, i[1]); sep._active = sep._plans[(2, i[1])]
    sep.load_variables(params)
    ev = sep.get_output(torch.from_numpy(mix).cuda(), False)
    for n in ocfg["source_names"]:
        assert ev[n].abs().max().item() <= 1.0
    tr = sep.get_output(torch.from_numpy(mix).cuda(), True)
    assert max(tr[n].abs().max().item() for n in ocfg["source_names"]) > 1.0


def test_overlapped_allreduce_plumbing_world1(lib):
    """The bucket-event hooks of wun_loss_backward_ex + the overlapped reducer, exercised with a
    1-rank RCCL group (the all-reduce of one rank is the identity, so gradients must be
    bit-identical to the plain path); multi-rank semantics are covered on CPU/gloo."""
    import socket
    import torch.distributed as dist
    from wave_u_net_amd.parallel import OverlappedGradAllReducer
    case = GOLDEN_CASES["full_small"]
    ocfg = _ocfg(case)
    params = golden_params(ocfg, case["seed"])
    sep, cfg = _make_sep(case, params)
    B = 4
    i, o = shapes.get_padding(ocfg, [B, case["frames"], 0])
    mix, targets = wt.synthetic_batch(ocfg, B, i[1], o[1], seed=3)
    tg = torch.stack([torch.from_numpy(targets[n]) for n in ocfg["source_names"]]).cuda()
    plan = sep._plan(B, i[1]); sep._active = plan
    sep.load_variables(params)
    dmix = torch.from_numpy(mix).cuda()
    sep.get_output(dmix, True)
    l0 = sep.loss_and_gradients(tg).clone()
    g0 = sep.grads.clone()
    created = False
    if not dist.is_initialized():
        sock = socket.socket(); sock.bind(("127.0.0.1", 0)); port = sock.getsockname()[1]; sock.close()
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % port, rank=0, world_size=1)
        created = True
    try:
        red = OverlappedGradAllReducer(plan.tensors, plan.info.arena_floats, bucket_mib=0.02, device=sep._dev())
        assert len(red.buckets) >= 3
        for _ in range(3):
            sep.get_output(dmix, True)
            l1 = sep.loss_and_gradients(tg, *red.begin())
            red.launch(sep.grads, force=True)
            red.finish()
            torch.cuda.synchronize()
            assert torch.equal(l1, l0) and torch.equal(sep.grads, g0)
    finally:
        if created:
            dist.destroy_process_group()


FULL_SIZE_CONFIGS = {
    # BASELINE.json configs[2..4] architectures at full size (fp32), one or two excerpts
    "M4_baseline_stereo": dict(output_type="difference", context=True, mono_downmix=False),
    "M5_full_learned": dict(output_type="difference", context=True, upsampling="learned", mono_downmix=False),
    "M6_full_multi_instrument": dict(output_type="difference", context=True, mono_downmix=False,
                                     task="multi_instrument"),
}


@pytest.mark.parametrize("name", sorted(FULL_SIZE_CONFIGS))
def test_full_size_named_configs_step_vs_oracle(lib, name):
    over = FULL_SIZE_CONFIGS[name]
    ocfg = shapes.finalize_config(dict(shapes.BASE_MODEL_CONFIG, **over))
    params = golden_params(ocfg, 91)
    sep = UnetAudioSeparator(wun.get_config("baseline", **over), device="cuda:0")
    B = 2
    i, o = shapes.get_padding(ocfg, [B, 16384, 0])
    assert (i[1], o[1]) == (147443, 16389)
    mix, targets = wt.synthetic_batch(ocfg, B, i[1], o[1], seed=92)
    sep._plan(B, i[1]); sep._active = sep._plans[(B, i[1])]
    sep.load_variables(params)
    outs = sep.get_output(torch.from_numpy(mix).cuda(), True)
    loss = sep.loss_and_gradients({k: torch.from_numpy(v) for k, v in targets.items()})
    torch.cuda.synchronize()
    tmix = torch.from_numpy(mix)
    oloss, ograds, oouts, tp = _oracle64(ocfg, params, mix, targets)
    _out_check(outs, oouts, ocfg["source_names"], name)
    # difference output: the sources add up to the cropped mix (OutputLayer.py:20-22)
    total = sum(outs[n] for n in ocfg["source_names"]).cpu()
    pad = (i[1] - o[1]) // 2
    assert (total - tmix[:, pad:i[1] - pad, :]).abs().max().item() <= 1e-5
    _loss_check(loss.item(), oloss, name)
    _grad_check(sep, tp, ograds, tag=name)


def test_deep_variant_16_levels_48_filters(lib):
    """BASELINE.json configs[4] architecture: 16 levels, 48 base channels, stereo, 4 sources,
    same padding, 589824-sample excerpt (9 * 2^16), 92.45 M parameters -- one excerpt, fp32:
    forward vs the oracle, loss, and EVERY gradient tensor (heuristic tilings; the tuned plan is checked
    against the float64 oracle by test_deep_variant_tuned_all_gradients_vs_float64)."""
    over = dict(num_layers=16, num_initial_filters=48, mono_downmix=False, task="multi_instrument",
                output_type="difference")
    ocfg = shapes.finalize_config(dict(shapes.BASE_MODEL_CONFIG, **over))
    assert shapes.num_params(ocfg) == 92452098
    params = golden_params(ocfg, 93)
    sep = UnetAudioSeparator(wun.get_config("baseline", **over), device="cuda:0")
    T = 589824
    mix, targets = wt.synthetic_batch(ocfg, 1, T, T, seed=94)
    sep._plan(1, T); sep._active = sep._plans[(1, T)]
    sep.load_variables(params)
    outs = sep.get_output(torch.from_numpy(mix).cuda(), True)
    loss = sep.loss_and_gradients({k: torch.from_numpy(v) for k, v in targets.items()})
    torch.cuda.synchronize()
    assert torch.isfinite(sep.grads).all()
    # the one comparison against the FLOAT32 oracle: its float64 backward of this 92 M-parameter,
    # 590 k-sample graph needs tens of GB of host memory; the fp32 oracle's own rounding is part of
    # the observed error here, hence the wider gradient tolerance
    tp = wt.params_to_torch(params, torch.float32, requires_grad=True)
    oloss, ograds = wt.train_step(ocfg, tp, torch.from_numpy(mix), {k: torch.from_numpy(v) for k, v in targets.items()})
    oouts = wt.get_output(ocfg, tp, torch.from_numpy(mix), True)
    _out_check(outs, oouts, ocfg["source_names"], "deep_l16_f48 (fp32 oracle)")
    _loss_check(loss.item(), oloss.item(), "deep_l16_f48 (fp32 oracle)")
    _grad_check(sep, tp, ograds, tol=5e-3, tag="deep_l16_f48 (fp32 oracle)")


def test_deep_variant_tuned_all_gradients_vs_float64(lib):
    """BASELINE.json configs[4] architecture (16 levels, 48 base channels, stereo, 4 sources, same padding) with the
    AUTOTUNED plan (wun_plan_tune -- the tilings the deep figures of DESIGN.md section 6 run), every one of its
    gradient tensors against the FLOAT64 oracle at the normal gradient tolerance.  The excerpt is 2 * 2^16 samples
    instead of 9 * 2^16 so that the float64 autograd graph stays within a few GB of host memory (same-padding
    model: every level, tile shape family and split-K path of the full-size plan is exercised; only the number of
    time tiles per launch differs), batch 2, oracle one excerpt at a time."""
    over = dict(num_layers=16, num_initial_filters=48, mono_downmix=False, task="multi_instrument",
                output_type="difference")
    ocfg = shapes.finalize_config(dict(shapes.BASE_MODEL_CONFIG, **over))
    params = golden_params(ocfg, 95)
    sep = UnetAudioSeparator(wun.get_config("baseline", **over), device="cuda:0")
    B, T = 2, 2 * 65536
    mix, targets = wt.synthetic_batch(ocfg, B, T, T, seed=96)
    sep._plan(B, T); sep._active = sep._plans[(B, T)]
    sep.load_variables(params)
    dmix = torch.from_numpy(mix).cuda()
    tg = {k: torch.from_numpy(v) for k, v in targets.items()}
    sep.tune(dmix, tg)
    assert sep.tune_export().startswith("wun-tune 2 ")
    outs = sep.get_output(dmix, True)
    loss = sep.loss_and_gradients(tg)
    torch.cuda.synchronize()
    oloss, ograds, oouts, tp = _oracle64(ocfg, params, mix, targets)
    assert len(tp) == len(ograds) == len(sep._active.tensors)
    _out_check(outs, oouts, ocfg["source_names"], "deep_l16_f48 tuned (float64 oracle)")
    _loss_check(loss.item(), oloss, "deep_l16_f48 tuned (float64 oracle)")
    _grad_check(sep, tp, ograds, tag="deep_l16_f48 tuned (float64 oracle)")


@pytest.mark.parametrize("name", ["baseline_context_small", "full_small", "baseline_small"])
def test_leaky_relu_tie_digital_silence(lib, name):
    """LeakyReLU at exactly 0 (Utils.py:79-80, tf.maximum(0.2x, x): TensorFlow's _MaximumGrad gives the tie to the
    first argument => gradient 0.2).  Digitally silent input segments with zero biases make exact-zero
    pre-activations the COMMON case at step 0 on real stems (Datasets.py:188-216 feeds MUSDB vocals): excerpt 0 is
    entirely silent, excerpt 1 is silent in its first half -- with zero biases every conv output over a silent
    stretch is exactly 0.0, so the tie rule decides the weight gradients of every layer above.  Outputs, loss and
    all gradients vs the float64 oracle (whose LeakyReLU implements the TF rule); a tie handled the torch way
    (0.6) or the 'natural' way (1.0) fails this test by orders of magnitude."""
    case = GOLDEN_CASES[name]
    ocfg = _ocfg(case)
    params = [(n, (np.zeros_like(v) if n.endswith("/bias") else v)) for n, v in golden_params(ocfg, case["seed"])]
    sep, cfg = _make_sep(case, params)
    B = 3
    i, o = shapes.get_padding(ocfg, [B, case["frames"], 0])
    mix, targets = wt.synthetic_batch(ocfg, B, i[1], o[1], seed=case["seed"] + 300)
    mix = mix.copy(); targets = {k: v.copy() for k, v in targets.items()}
    pad = (i[1] - o[1]) // 2
    mix[0] = 0.0
    mix[1, :i[1] // 2] = 0.0
    for k in targets:                                    # the sources of a silent mix are silent (Utils.py:35)
        targets[k][0] = 0.0
        targets[k][1, :max(0, i[1] // 2 - pad)] = 0.0
    sep._plan(B, i[1]); sep._active = sep._plans[(B, i[1])]
    sep.load_variables(params)
    outs = sep.get_output(torch.from_numpy(mix).cuda(), True)
    loss = sep.loss_and_gradients({k: torch.from_numpy(v) for k, v in targets.items()})
    torch.cuda.synchronize()
    tp = wt.params_to_torch(params, torch.float64, requires_grad=True)
    tmix = torch.tensor(mix, dtype=torch.float64)
    ttg = {k: torch.tensor(v, dtype=torch.float64) for k, v in targets.items()}
    oloss, ograds = wt.train_step(ocfg, tp, tmix, ttg)
    oouts = wt.get_output(ocfg, tp, tmix, True)
    # the construction really produces ties: the first conv's pre-activation of the silent excerpt is exactly 0
    k0, b0 = tp[0][1].detach(), tp[1][1].detach()
    pre = wt.conv1d_tf(tmix[:1].permute(0, 2, 1), k0, b0, not ocfg["context"])
    assert (pre == 0).all()
    _out_check(outs, oouts, ocfg["source_names"], "lrelu_tie_" + name)
    _loss_check(loss.item(), oloss.item(), "lrelu_tie_" + name)
    _grad_check(sep, tp, ograds, tag="lrelu_tie_" + name)
    # sensitivity of the construction: with the tie split the torch way the oracle itself moves by far more than
    # the tolerance, i.e. this test can tell the rules apart
    tp2 = wt.params_to_torch(params, torch.float64, requires_grad=True)
    old = wt.leaky_relu
    try:
        wt.leaky_relu = lambda x, pin=None: torch.maximum(0.2 * x, x)
        _, g_split = wt.train_step(ocfg, tp2, tmix, ttg)
    finally:
        wt.leaky_relu = old
    moved = max(((a - b).abs().max() / b.abs().max().clamp_min(1e-30)).item() for a, b in zip(g_split, ograds))
    assert moved > 20 * GRAD_TOL, moved          # observed 0.05 .. 0.10 of max|g| vs GRAD_TOL 5e-4


@pytest.mark.parametrize("name", ["full_multi_small", "baseline_small"])
def test_autotuned_plan_keeps_parity(lib, name):
    """wun_plan_tune only changes tilings / split factors: results stay within the fp32 tolerances
    and remain bit-deterministic afterwards."""
    case = GOLDEN_CASES[name]
    ocfg = _ocfg(case)
    params = golden_params(ocfg, case["seed"])
    sep, cfg = _make_sep(case, params)
    B = 3
    i, o = shapes.get_padding(ocfg, [B, case["frames"], 0])
    mix, targets = wt.synthetic_batch(ocfg, B, i[1], o[1], seed=case["seed"] + 7)
    sep._plan(B, i[1]); sep._active = sep._plans[(B, i[1])]
    sep.load_variables(params)
    dmix = torch.from_numpy(mix).cuda()
    tg = {k: torch.from_numpy(v) for k, v in targets.items()}
    sep.tune(dmix, tg)
    outs = sep.get_output(dmix, True)
    loss = sep.loss_and_gradients(tg).clone()
    g1 = sep.grads.clone()
    tp = wt.params_to_torch(params, torch.float64, requires_grad=True)
    oloss, ograds = wt.train_step(ocfg, tp, torch.tensor(mix, dtype=torch.float64),
                                  {k: torch.tensor(v, dtype=torch.float64) for k, v in targets.items()})
    _loss_check(loss.item(), oloss.item(), "autotuned_" + name)
    _grad_check(sep, tp, ograds, tag="autotuned_" + name)
    sep.get_output(dmix, True)
    l2 = sep.loss_and_gradients(tg)
    assert torch.equal(l2, loss) and torch.equal(sep.grads, g1)


def test_autotuned_full_size_m1_context(lib):
    ocfg = shapes.finalize_config(dict(shapes.BASE_MODEL_CONFIG, context=True))
    params = golden_params(ocfg, 77)
    sep = UnetAudioSeparator(wun.get_config("m1_context"), device="cuda:0")
    B = 2
    i, o = shapes.get_padding(ocfg, [B, 16384, 0])
    mix, targets = wt.synthetic_batch(ocfg, B, i[1], o[1], seed=78)
    sep._plan(B, i[1]); sep._active = sep._plans[(B, i[1])]
    sep.load_variables(params)
    tg = {k: torch.from_numpy(v) for k, v in targets.items()}
    sep.tune(torch.from_numpy(mix).cuda(), tg)
    sep.get_output(torch.from_numpy(mix).cuda(), True)
    loss = sep.loss_and_gradients(tg)
    torch.cuda.synchronize()
    oloss, ograds, oouts, tp = _oracle64(ocfg, params, mix, targets)
    _loss_check(loss.item(), oloss, "autotuned_M1_context_full_B2")
    _grad_check(sep, tp, ograds, tag="autotuned_M1_context_full_B2")


def test_tuned_plan_error_is_the_leaky_relu_sign_floor(lib):
    """Where an autotuned plan sits 1e-4 .. 5e-4 (of max|g|) from float64 on some weight gradients, the cause is ONE (or a
    few) LeakyReLU input within float32 rounding of zero whose sign -- hence a factor 5 on that element's gradient --
    depends on the summation order of the forward conv that produced it (profiles/round4_wsdiff_single_mask_flip.txt:
    one element of the workspace differs by exactly 5x between two plans; profiles/round4_lrelu_flip_study_B2.json: the
    torch-CPU float32 oracle jumps by 2e-4 under 3e-7 input perturbations).  A kernel bug of the same size (a dropped
    boundary position) would NOT disappear when the masks are pinned.  So, on the full-size M1 + context plan:
      * tuned backward kernels (input + weight gradients) on the HEURISTIC forward pass (tuned table with its forward
        entries reset): every gradient tensor within 2e-5 of float64 -- the tuned backward kernels are exact;
      * the fully tuned plan: outputs within OUT_TOL (its forward kernels differ from the heuristic ones by rounding
        only), gradients within GRAD_TOL."""
    ocfg = shapes.finalize_config(dict(shapes.BASE_MODEL_CONFIG, context=True))
    params = golden_params(ocfg, 77)
    B = 2
    i, o = shapes.get_padding(ocfg, [B, 16384, 0])
    mix, targets = wt.synthetic_batch(ocfg, B, i[1], o[1], seed=78)
    tg = {k: torch.from_numpy(v) for k, v in targets.items()}
    dmix = torch.from_numpy(mix).cuda()
    oloss, ograds, oouts, tp = _oracle64(ocfg, params, mix, targets)
    names = ocfg["source_names"]

    def fresh():
        sep = UnetAudioSeparator(wun.get_config("m1_context"), device="cuda:0")
        sep._plan(B, i[1]); sep._active = sep._plans[(B, i[1])]
        sep.load_variables(params)
        return sep

    sep = fresh()
    sep.tune(dmix, tg)
    table = sep.tune_export()
    lines = table.strip().split("\n")
    fwd_reset = lines[0] + "\n" + "\n".join("cf -1 0" if ln.startswith("cf ") else ln for ln in lines[1:]) + "\n"

    def worst_grad_err(s_):
        g = s_.gradients()
        w = 0.0
        for (n, _), og in zip(tp, ograds):
            got = g[n].cpu().double(); og = og.double()
            w = max(w, (got - og).abs().max().item() / max(og.abs().max().item(), 1e-30))
        return w

    sep2 = fresh()
    sep2.get_output(dmix, True)
    sep2.tune_import(fwd_reset)
    sep2.get_output(dmix, True)
    sep2.loss_and_gradients(tg)
    torch.cuda.synchronize()
    e_bwd = worst_grad_err(sep2)
    record("gradients_vs_float64_oracle", "tuned backward kernels on the heuristic forward pass, M1_context_B2", e_bwd, 2e-5)
    assert e_bwd <= 2e-5, e_bwd

    outs = sep.get_output(dmix, True)
    sep.loss_and_gradients(tg)
    torch.cuda.synchronize()
    _out_check(outs, oouts, names, "fully_tuned_M1_context_B2")
    e_all = worst_grad_err(sep)
    record("gradients_vs_float64_oracle", "fully tuned plan, M1_context_B2 (LeakyReLU sign floor)", e_all, GRAD_TOL)
    assert e_all <= GRAD_TOL, e_all


def test_benchmarked_configuration_b16_tuned_vs_oracle(lib):
    """EXACTLY what bench.py times -- BASELINE.json configs[1]: M1 with context, batch 16,
    147443 -> 16389 samples, the Trainer's seed-1337 weights, the synthetic_source(seed 1337) batch
    and the tilings bench.py runs (the committed table profiles/round6_tune_table.txt, handed to
    Trainer.tune as read-only text, when it matches this build; a fresh wun_plan_tune otherwise) -- compared element
    by element with the FLOAT64 oracle: outputs, loss and all 54 gradient tensors of the whole batch
    (the oracle runs one excerpt at a time and averages: the loss is a mean over excerpts)."""
    from wave_u_net_amd.training import Trainer, synthetic_source
    cfg = wun.get_config("m1_context")
    table = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "round6_tune_table.txt")
    text = open(table).read() if os.path.exists(table) else None
    before = text
    tr = Trainer(cfg, batch_size=16)
    assert (tr.t_in, tr.t_out) == (147443, 16389)
    mix, targets = synthetic_source(cfg, 16, tr.t_in, tr.t_out, tr.device, seed=1337)()
    tr.tune(mix, targets, pinned_table=text)          # the committed table is handed over as text: read-only
    assert tr.tune_source in ("pinned", "cache", "autotuned")
    if text is not None:
        assert open(table).read() == before           # never rewritten
    sep = tr.sep
    assert sep.tune_export().startswith("wun-tune 2 ")               # the plan really runs tuned tilings
    outs = sep.get_output(mix, True)
    loss = sep.loss_and_gradients(targets)
    torch.cuda.synchronize()
    names = cfg["source_names"]
    var = sep.variables()
    params = [(n, var[n].detach().cpu().numpy()) for n, _, _ in sep._active.tensors]
    ocfg = shapes.finalize_config(dict(shapes.BASE_MODEL_CONFIG, context=True))
    hmix = mix.cpu().numpy()
    htg = {n: targets[k].cpu().numpy() for k, n in enumerate(names)}
    oloss, ograds, oouts, tp = _oracle64(ocfg, params, hmix, htg)
    _out_check(outs, oouts, names, "bench_config_B16_tuned")
    _loss_check(loss.item(), oloss, "bench_config_B16_tuned")
    # Max-norm: the fully tuned plan sits ON the LeakyReLU sign floor (one pre-activation within float32
    # rounding of zero flips a derivative 1 <-> 0.2; 3e-4 .. 5.1e-4 of max|g| observed over this round's tables, the fp32
    # CPU oracle itself shows 2e-4 under 3e-7 input perturbations at B = 2, and B = 16 has 8x the pre-activations), so the
    # max-norm bound here is SURVEY section 7's 1e-3, not GRAD_TOL (observed 5.1e-4 under the pinned table AND 5.3e-4 with
    # the heuristic forward kernels: at B = 16 both forward passes contain borderline pre-activations); the relative L2
    # error per conv kernel is recorded and bounded too (1.0e-4 .. 1.3e-4 observed; a wrong tap / dropped position /
    # missing split moves it by 1e-2 .. 1).  What shows that the kernels themselves are exact is the B = 2 test above,
    # where no pre-activation is borderline: 6e-7.
    e_free = _grad_check(sep, tp, ograds, tol=GRAD_TOL_FULL_TUNED, tag="bench_config_B16_tuned")
    l2, which = _grad_rel_l2(sep, tp, ograds)
    record("gradients_rel_l2_vs_float64_oracle", "bench_config_B16_tuned (worst: %s)" % which, l2, GRAD_L2_TOL)
    assert l2 <= GRAD_L2_TOL, (l2, which)
    # ... and what is left when the float64 oracle takes the SAME LeakyReLU branches as the kernels did (their masks
    # read back from the workspace): the arithmetic alone.  If the 1e-4 .. 5e-4 above were anything but branch flips --
    # a B-dependent defect, a split-K order issue in the batch-folded tiles that only engage at B = 16 -- it would
    # still be here; bound 2e-5.
    pins = _gpu_pins(sep, ocfg)
    flips = _count_flips(ocfg, params, hmix, pins)
    ploss, pgrads, pouts = wt.chunked_train_step(ocfg, params, hmix, htg, dtype=torch.float64, chunk=1, want_outputs=True, pins=pins)
    _out_check(outs, pouts, names, "bench_config_B16_tuned (branch-pinned oracle)")
    _loss_check(loss.item(), ploss, "bench_config_B16_tuned (branch-pinned oracle)")
    e_pin = _grad_check(sep, tp, pgrads, tol=GRAD_TOL_PINNED, tag="bench_config_B16_tuned (branch-pinned float64 oracle; %d of %d "
                        "LeakyReLU inputs on the other side of 0 in float64)" % flips)
    l2p, whichp = _grad_rel_l2(sep, tp, pgrads)
    record("gradients_rel_l2_vs_float64_oracle", "bench_config_B16_tuned (branch-pinned; worst: %s)" % whichp, l2p, GRAD_TOL_PINNED)
    assert l2p <= GRAD_TOL_PINNED and e_pin <= GRAD_TOL_PINNED, (e_pin, l2p, whichp)
    # The free comparison's bound follows from the flip count: with NO flipped branch the free oracle
    # IS the pinned one and must meet the pinned bound; every flipped branch may move the gradients upstream of it by a few
    # 1e-4 of max|g| (one x5 in one dz element: profiles/round4_wsdiff_single_mask_flip.txt), up to SURVEY section 7's 1e-3.
    tol_free = min(GRAD_TOL_FULL_TUNED, GRAD_TOL_PINNED + 2.5e-4 * flips[0])
    record("gradients_vs_float64_oracle", "bench_config_B16_tuned (free oracle; bound from %d flipped branches)" % flips[0], e_free, tol_free)
    assert e_free <= tol_free, (e_free, tol_free, flips)
    # ... and the same table with its forward entries reset (heuristic forward kernels, tuned backward kernels): other
    # masks, same bounds
    if text is not None and tr.tune_source == "pinned":
        lines = text.strip().split("\n")
        sep.tune_import(lines[0] + "\n" + "\n".join("cf -1 0" if ln.startswith("cf ") else ln for ln in lines[1:]) + "\n")
        sep.get_output(mix, True)
        sep.loss_and_gradients(targets)
        torch.cuda.synchronize()
        _grad_check(sep, tp, ograds, tol=GRAD_TOL_FULL_TUNED, tag="bench_config_B16_tuned_backward_on_heuristic_forward")
        l2b, whichb = _grad_rel_l2(sep, tp, ograds)
        record("gradients_rel_l2_vs_float64_oracle", "bench_config_B16_tuned_backward_on_heuristic_forward (worst: %s)" % whichb,
               l2b, GRAD_L2_TOL)
        assert l2b <= GRAD_L2_TOL, (l2b, whichb)
