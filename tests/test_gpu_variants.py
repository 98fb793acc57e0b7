"""Forced sweeps over EVERY tile variant of the two MFMA kernels (run with -m gpu on an MI355X).

The autotuner picks tilings by speed alone, so each of them has to be correct on its own:
  * conv_mfma_kernel: all variants of the table x split-K {1, 3} x the launch kinds the plan uses --
    stride 1, the stride-2 de-interleaving loader, two virtual sources (crop_and_concat) with
    accumulate + LeakyReLU mask, strided/offset output (transposed stride-2 conv, one phase) and the
    fused two-phase transposed stride-2 conv -- through the C ABI test hooks
    (wun_op_force_conv_variant; a choice the dispatcher would never make for a launch is REJECTED
    by the library, not computed);
  * wgrad_mfma_kernel<MTW, NW>: every instantiated geometry x split count {auto, 1, 3}
    (wun_op_force_wgrad_variant).
Every comparison is against a float64 torch-CPU reference of the same op; the last tests assert
that no variant / geometry was left unexercised."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from _observed import record

pytestmark = pytest.mark.gpu

from wave_u_net_amd import _lib                   # noqa: E402

OP_TOL = 2e-5          # x max|ref|: fp32 MFMA chain over K*Cin <= 3000 products vs float64 (observed: see DESIGN.md)

_RAN_CONV = {}         # variant -> set of launch kinds it was checked under
_RAN_WGRAD = set()     # (mtw, nw)


@pytest.fixture(scope="module")
def lib():
    assert torch.cuda.is_available(), "GPU tests need an MI355X"
    return _lib.load()


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _cuda(a):
    return torch.as_tensor(np.ascontiguousarray(a), dtype=torch.float32).cuda()


def _conv64(x, w, bias, stride, pad_left, t_out):
    """float64 cross-correlation y[b][co][q] = bias + sum w[k][ci][co] x[b][ci][q*stride + k - pad_left]."""
    K = w.shape[0]
    xx = torch.as_tensor(x, dtype=torch.float64)
    need = (t_out - 1) * stride + K - pad_left
    xx = F.pad(xx, (pad_left, max(0, need - xx.shape[2])))
    b = None if bias is None else torch.as_tensor(bias, dtype=torch.float64)
    return F.conv1d(xx, torch.as_tensor(w, dtype=torch.float64).permute(2, 1, 0), b, stride=stride)[:, :, :t_out]


def _t_out(T, K, stride, same):
    return T if same else (T - K) // stride + 1


# (B, Cin, Cout, K, T, same)  -- chosen so that, between them, every tile variant is a legal choice:
# N = 48 admits 32/48/64/80-column tiles, N = 96 admits 96/128-column tiles, T >= 400 admits 384-row
# tiles, the 1-/2-channel cases admit the 4-channel-chunk audio-input tiles, the B = 16 short cases
# the batch-folded tiles.  (Variants 42..55 are a RETIRED index range -- round 4's register-window conv tiles, removed in
# round 5; the range stays reserved so the in-workgroup split-K tiles keep their numbers in committed tuning tables.)
RETIRED_VARIANTS = range(42, 56)
SWEEP_CASES = [
    (2, 24, 48, 15, 800, False),
    (2, 40, 96, 5, 420, True),
    (2, 72, 80, 15, 430, False),
    (2, 1, 24, 15, 900, False),
    (2, 2, 24, 15, 300, False),
    (16, 40, 48, 15, 39, False),
    (16, 48, 96, 15, 95, False),
    (6, 72, 48, 5, 77, True),
    (16, 24, 64, 15, 151, False),
    # 16-byte aligned output rows (lengths whose outputs are multiples of 4)
    (2, 24, 48, 15, 814, False),
    (2, 48, 96, 15, 814, False),
    (2, 48, 80, 5, 404, False),
    # the in-workgroup split-K tiles (conv_mfma_kernel<..., KG = 3>: whole 8-channel chunks in multiples of 3): short
    # rows, wide layers -- what the deep levels look like
    (16, 72, 48, 5, 40, False),
    (16, 144, 64, 15, 60, False),
    (4, 72, 96, 5, 100, False),
]


def _sweep(lib, kind, launch, check, ks_list=(1, 3)):
    """Run `launch()` under every (variant, ksplit); rejected choices are skipped."""
    nvar = lib.wun_op_num_conv_variants()
    ran = 0
    try:
        for v in range(nvar):
            for ks in ks_list:
                lib.wun_op_force_conv_variant(v, ks)
                rc = launch()
                if rc != 0:
                    continue
                torch.cuda.synchronize()
                check(v, ks)
                _RAN_CONV.setdefault(v, set()).add(kind)
                ran += 1
    finally:
        lib.wun_op_force_conv_variant(-1, 0)
    return ran


@pytest.mark.parametrize("case", SWEEP_CASES, ids=[str(c) for c in SWEEP_CASES])
@pytest.mark.parametrize("stride", [1, 2])
def test_conv_every_variant_forward(lib, case, stride):
    B, Cin, Cout, K, T, same = case
    if stride == 2 and same:
        pytest.skip("the plan only uses the stride-2 loader with valid padding")
    rng = np.random.default_rng(abs(hash((case, stride))) % (2 ** 31))
    x = rng.uniform(-1, 1, (B, Cin, T)).astype(np.float32)
    w = (rng.uniform(-1, 1, (K, Cin, Cout)) / np.sqrt(K * Cin)).astype(np.float32)
    b = rng.uniform(-0.1, 0.1, Cout).astype(np.float32)
    pad = (K - 1) // 2 if same else 0
    t_out = _t_out(T, K, stride, same)
    ref = _conv64(x, w, b, stride, pad, t_out)
    ref = torch.maximum(0.2 * ref, ref).numpy()
    scale = max(1.0, np.abs(ref).max())
    dx, dw, db_ = _cuda(x), _cuda(w), _cuda(b)
    y = torch.empty((B, Cout, t_out), device="cuda")
    worst = [0.0]

    def launch():
        y.fill_(float("nan"))
        return lib.wun_op_conv1d(dx.data_ptr(), dw.data_ptr(), db_.data_ptr(), y.data_ptr(), B, Cin, Cout, K, T,
                                 t_out, stride, pad, 1, _stream())

    def check(v, ks):
        got = y.cpu().numpy()
        assert np.isfinite(got).all(), (v, ks)
        err = np.abs(got - ref).max() / scale
        worst[0] = max(worst[0], err)
        assert err <= OP_TOL, (v, ks, err)

    ran = _sweep(lib, "stride%d" % stride, launch, check)
    assert ran >= 2
    record("conv_every_variant_forward", "stride%d %s" % (stride, case), worst[0], OP_TOL)


@pytest.mark.parametrize("case", SWEEP_CASES, ids=[str(c) for c in SWEEP_CASES])
def test_conv_every_variant_two_sources_accumulate_mask(lib, case):
    """The up-path launches: virtual concat of two sources; and the gradient-side epilogue: multiply
    by the LeakyReLU derivative of the stored forward activation, accumulate into the destination."""
    B, Cin, Cout, K, T, same = case
    if Cin < 8:
        pytest.skip("two sources need >= 8 input channels")
    c0 = (Cin // 2 + 3) // 4 * 4
    c1 = Cin - c0
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31) + 3)
    x = rng.uniform(-1, 1, (B, Cin, T)).astype(np.float32)
    w = (rng.uniform(-1, 1, (K, Cin, Cout)) / np.sqrt(K * Cin)).astype(np.float32)
    pad = (K - 1) // 2 if same else 0
    t_out = _t_out(T, K, 1, same)
    fwd = rng.uniform(-1, 1, (B, Cout, t_out)).astype(np.float32)          # "forward activation" -> mask
    base = rng.uniform(-1, 1, (B, Cout, t_out)).astype(np.float32)         # destination content before the launch
    ref = _conv64(x, w, None, 1, pad, t_out).numpy() * np.where(fwd > 0, 1.0, 0.2) + base.astype(np.float64)
    scale = max(1.0, np.abs(ref).max())
    x0, x1 = _cuda(x[:, :c0]), _cuda(x[:, c0:])
    dw, dmask, dbase = _cuda(w), _cuda(fwd), _cuda(base)
    y = torch.empty((B, Cout, t_out), device="cuda")
    worst = [0.0]

    def launch():
        y.copy_(dbase)
        return lib.wun_op_conv1d_ex(x0.data_ptr(), c0, x1.data_ptr(), c1, dw.data_ptr(), None, y.data_ptr(),
                                    dmask.data_ptr(), B, Cout, K, T, t_out, t_out, 1, pad, 0, 1, 1, 0, _stream())

    def check(v, ks):
        got = y.cpu().numpy()
        assert np.isfinite(got).all(), (v, ks)
        err = np.abs(got - ref).max() / scale
        worst[0] = max(worst[0], err)
        assert err <= OP_TOL, (v, ks, err)

    ran = _sweep(lib, "two_source_accum_mask", launch, check)
    assert ran >= 2
    record("conv_every_variant_two_sources", str(case), worst[0], OP_TOL)


@pytest.mark.parametrize("case", SWEEP_CASES[:3] + SWEEP_CASES[5:7], ids=[str(c) for c in SWEEP_CASES[:3] + SWEEP_CASES[5:7]])
def test_conv_every_variant_strided_output(lib, case):
    """One output phase of a transposed stride-2 conv / the same-padding decimation gradient: outputs
    land at y[ooff + 2q] and accumulate there."""
    B, Cin, Cout, K, T, same = case
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31) + 5)
    x = rng.uniform(-1, 1, (B, Cin, T)).astype(np.float32)
    w = (rng.uniform(-1, 1, (K, Cin, Cout)) / np.sqrt(K * Cin)).astype(np.float32)
    pad = (K - 1) // 2 if same else 0
    t_out = _t_out(T, K, 1, same)
    t_y = 2 * t_out + 1
    base = rng.uniform(-1, 1, (B, Cout, t_y)).astype(np.float32)
    ref = base.astype(np.float64).copy()
    ref[:, :, 1:1 + 2 * t_out:2] += _conv64(x, w, None, 1, pad, t_out).numpy()
    scale = max(1.0, np.abs(ref).max())
    dx, dw, dbase = _cuda(x), _cuda(w), _cuda(base)
    y = torch.empty((B, Cout, t_y), device="cuda")
    worst = [0.0]

    def launch():
        y.copy_(dbase)
        return lib.wun_op_conv1d_ex(dx.data_ptr(), Cin, None, 0, dw.data_ptr(), None, y.data_ptr(), None, B, Cout, K,
                                    T, t_out, t_y, 1, pad, 0, 1, 2, 1, _stream())

    def check(v, ks):
        got = y.cpu().numpy()
        assert np.isfinite(got).all(), (v, ks)
        err = np.abs(got - ref).max() / scale
        worst[0] = max(worst[0], err)
        assert err <= OP_TOL, (v, ks, err)

    ran = _sweep(lib, "strided_output", launch, check)
    assert ran >= 2
    record("conv_every_variant_strided_output", str(case), worst[0], OP_TOL)


_COPY_CASES = SWEEP_CASES[:4] + SWEEP_CASES[5:7] + SWEEP_CASES[12:14]


@pytest.mark.parametrize("case", _COPY_CASES, ids=[str(c) for c in _COPY_CASES])
@pytest.mark.parametrize("mode", ["expand_stride2", "parity_split_masked", "acc_window"])
def test_conv_every_variant_secondary_copies(lib, case, mode):
    """Round 6 (a context plan computes every conv output ONCE: the decimated stream is a slice of the encoder output,
    UnetAudioSeparator.py:98-100) -- the epilogue features behind it, on every tile variant x split-K through the
    wun_op_set_conv_copies hook:
      expand_stride2      a stride-2 conv that also writes output q at copy0[2q - lo] inside a window (the decimating launch
                          writing the even positions of the skip window): bit-equal to the main output there, untouched elsewhere;
      parity_split_masked a stride-1 launch with a LeakyReLU-derivative mask whose EVEN outputs are also stored compactly in
                          copy0 and whose ODD outputs in copy1 (an up level's input gradient splitting the skip window's
                          gradient by parity): both bit-equal to the main output;
      acc_window          accumulate only inside [lo, lo + len) of the row, store elsewhere (the transposed conv that fills
                          a gradient row which already holds the even half of the window's gradient)."""
    B, Cin, Cout, K, T, same = case
    rng = np.random.default_rng(abs(hash((case, mode))) % (2 ** 31) + 11)
    x = rng.uniform(-1, 1, (B, Cin, T)).astype(np.float32)
    w = (rng.uniform(-1, 1, (K, Cin, Cout)) / np.sqrt(K * Cin)).astype(np.float32)
    stride = 2 if mode == "expand_stride2" else 1
    if stride == 2 and same:
        pytest.skip("the plan only uses the stride-2 loader with valid padding")
    pad = (K - 1) // 2 if same else 0
    t_out = _t_out(T, K, stride, same)
    if t_out < 6:
        pytest.skip("row too short for a window")
    conv = _conv64(x, w, None, stride, pad, t_out).numpy()
    dx, dw = _cuda(x), _cuda(w)
    y = torch.empty((B, Cout, t_out), device="cuda")
    worst = [0.0]
    POISON = -7.25
    if mode == "expand_stride2":
        lo, ln = 2 * (t_out // 4) + 1, max(3, t_out // 2) | 1          # an odd window start, odd length
        c0 = torch.empty((B, Cout, ln), device="cuda")
        ref = np.maximum(0.2 * conv, conv)
        args = (c0.data_ptr(), ln, 1, lo, ln, None, 0, 0, 0)
        acc_flag, mask_ptr, lrelu = 0, None, 1
    elif mode == "parity_split_masked":
        fwd = rng.uniform(-1, 1, (B, Cout, t_out)).astype(np.float32)
        dmask = _cuda(fwd)
        ref = conv * np.where(fwd > 0, 1.0, 0.2)
        ne, no = (t_out + 1) // 2, t_out // 2
        c0 = torch.empty((B, Cout, ne), device="cuda")
        c1 = torch.empty((B, Cout, max(no, 1)), device="cuda")
        args = (c0.data_ptr(), ne, 0, 0, 0, c1.data_ptr(), max(no, 1), 0, 0)
        acc_flag, mask_ptr, lrelu = 0, dmask.data_ptr(), 0
    else:
        base = rng.uniform(-1, 1, (B, Cout, t_out)).astype(np.float32)
        dbase = _cuda(base)
        lo, ln = t_out // 3 + 1, max(2, t_out // 3)
        ref = conv.copy()
        ref[:, :, lo:lo + ln] += base[:, :, lo:lo + ln].astype(np.float64)
        args = (None, 0, 0, 0, 0, None, 0, lo, ln)
        acc_flag, mask_ptr, lrelu = 1, None, 0
    scale = max(1.0, np.abs(ref).max())

    def launch():
        if mode == "acc_window":
            y.copy_(dbase)
        else:
            y.fill_(float("nan"))
            c0.fill_(POISON)
            if mode == "parity_split_masked":
                c1.fill_(POISON)
        _lib.check(lib.wun_op_set_conv_copies(*args))
        try:
            return lib.wun_op_conv1d_ex(dx.data_ptr(), Cin, None, 0, dw.data_ptr(), None, y.data_ptr(), mask_ptr, B, Cout, K,
                                        T, t_out, t_out, stride, pad, lrelu, acc_flag, 1, 0, _stream())
        finally:
            lib.wun_op_set_conv_copies(None, 0, 0, 0, 0, None, 0, 0, 0)

    def check(v, ks):
        got = y.cpu().numpy()
        assert np.isfinite(got).all(), (v, ks)
        err = np.abs(got - ref).max() / scale
        worst[0] = max(worst[0], err)
        assert err <= OP_TOL, (v, ks, err)
        if mode == "expand_stride2":
            g0 = c0.cpu().numpy()
            exp = np.full(g0.shape, POISON, dtype=np.float32)
            for q in range(t_out):
                pos = 2 * q - lo
                if 0 <= pos < ln:
                    exp[:, :, pos] = got[:, :, q]
            assert np.array_equal(g0, exp), (v, ks)                       # same bits where written, untouched elsewhere
        elif mode == "parity_split_masked":
            assert np.array_equal(c0.cpu().numpy()[:, :, :ne], got[:, :, 0::2]), (v, ks)
            if no:
                assert np.array_equal(c1.cpu().numpy()[:, :, :no], got[:, :, 1::2]), (v, ks)

    ran = _sweep(lib, "copies_" + mode, launch, check)
    assert ran >= 2
    record("conv_every_variant_secondary_copies", "%s %s" % (mode, case), worst[0], OP_TOL)


PHASE2_CASES = [
    # (B, Cin, Cout, K, T_in): input gradient of a stride-2 valid conv; Cin (the GEMM N) decides which
    # fused two-phase tiles (32/64/96 columns = 16/32/48 channels x 2 phases) are legal
    # rows long enough that the op entry takes the FUSED launch (>= 64 natural workgroups; shorter rows fall back to
    # two strided single-phase launches, which the last case keeps covered): 24 / 72 exercise the packed 24-channel
    # tiles (NW == 3, lane exchange in the epilogue), odd and even T the scalar / vector store paths
    (2, 24, 48, 15, 73715),
    (2, 72, 96, 15, 20004),
    (2, 48, 72, 15, 24001),
    (3, 32, 40, 15, 16100),
    (2, 96, 24, 7, 12000),
    (2, 24, 48, 15, 1400),
]


@pytest.mark.parametrize("case", PHASE2_CASES, ids=[str(c) for c in PHASE2_CASES])
def test_conv_every_variant_fused_two_phase_dgrad(lib, case):
    B, Cin, Cout, K, T = case
    t_out = (T - K) // 2 + 1
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31) + 9)
    w = (rng.uniform(-1, 1, (K, Cin, Cout)) / np.sqrt(K * Cin)).astype(np.float32)
    dz = rng.uniform(-1, 1, (B, Cout, t_out)).astype(np.float32)
    xt = torch.zeros((B, Cin, T), dtype=torch.float64, requires_grad=True)
    yy = F.conv1d(xt, torch.tensor(w, dtype=torch.float64).permute(2, 1, 0), None, stride=2)
    (yy * torch.tensor(dz, dtype=torch.float64)).sum().backward()
    ref = xt.grad.numpy()
    scale = max(1.0, np.abs(ref).max())
    dw, dzg = _cuda(w), _cuda(dz)
    wts = torch.empty(2 * K * Cin * Cout + 64, device="cuda")
    gdx = torch.empty((B, Cin, T), device="cuda")
    worst = [0.0]

    def launch():
        gdx.fill_(float("nan"))
        return lib.wun_op_conv1d_dgrad(dzg.data_ptr(), dw.data_ptr(), gdx.data_ptr(), wts.data_ptr(), B, Cin, Cout, K,
                                       T, t_out, 2, 0, _stream())

    def check(v, ks):
        got = gdx.cpu().numpy()
        assert np.isfinite(got).all(), v
        err = np.abs(got - ref).max() / scale
        worst[0] = max(worst[0], err)
        assert err <= OP_TOL, (v, err)

    ran = _sweep(lib, "fused_two_phase", launch, check, ks_list=(1,))
    assert ran >= 2
    record("conv_every_variant_fused_two_phase", str(case), worst[0], OP_TOL)


DMA_CASES = [
    # (B, C0, C1, Cout, K, T, pad_left): rows 16-byte aligned (T % 4 == 0), channel counts multiples of 8, so the
    # stride-1 launches take the DMA staging path; pad_left 0..3 walks the sub-vector shift, pad_left > 0 and the
    # overhanging last tile exercise the zero fix of the edge tiles, C1 > 0 the second source
    (2, 24, 0, 48, 15, 800, 0), (2, 16, 24, 40, 5, 420, 2), (3, 72, 0, 80, 15, 432, 7), (2, 8, 8, 24, 15, 1000, 1),
    (2, 40, 0, 96, 5, 404, 3), (4, 32, 16, 56, 9, 640, 4),
]


@pytest.mark.parametrize("case", DMA_CASES, ids=[str(c) for c in DMA_CASES])
def test_conv_dma_staging_equals_register_staging(lib, case, monkeypatch):
    """The DMA instantiations (global -> LDS directly, sub-vector shift folded into the operand offset, edge tiles zeroed
    after landing) against the register-staged instantiations of the SAME tile (WUN_NO_DMA=1): the MFMA stream is the
    same, so the results must agree BIT FOR BIT -- for every variant x split-K the dispatcher accepts -- and both match
    the float64 reference."""
    B, C0, C1, Cout, K, T, pad = case
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31) + 21)
    x0 = rng.uniform(-1, 1, (B, C0, T)).astype(np.float32)
    x1 = rng.uniform(-1, 1, (B, max(C1, 1), T)).astype(np.float32)
    w = (rng.uniform(-1, 1, (K, C0 + C1, Cout)) / np.sqrt(K * (C0 + C1))).astype(np.float32)
    bias = rng.uniform(-0.1, 0.1, Cout).astype(np.float32)
    t_out = T - K + 1 + min(pad, K - 1)                 # some zero padding on the left, valid on the right
    xcat = np.concatenate([x0, x1[:, :C1]], axis=1) if C1 > 0 else x0
    ref = _conv64(xcat, w, bias, 1, pad, t_out)
    ref = torch.maximum(0.2 * ref, ref).numpy()
    scale = max(1.0, np.abs(ref).max())
    d0, d1, dw, db_ = _cuda(x0), _cuda(x1), _cuda(w), _cuda(bias)
    y = torch.empty((B, Cout, t_out), device="cuda")
    nvar = lib.wun_op_num_conv_variants()
    compared = 0
    try:
        for v in range(nvar):
            for ks in (1, 3):
                lib.wun_op_force_conv_variant(v, ks)
                outs = []
                for no_dma in (False, True):
                    if no_dma:
                        monkeypatch.setenv("WUN_NO_DMA", "1")
                    else:
                        monkeypatch.delenv("WUN_NO_DMA", raising=False)
                    y.fill_(float("nan"))
                    rc = lib.wun_op_conv1d_ex(d0.data_ptr(), C0, d1.data_ptr() if C1 > 0 else None, C1, dw.data_ptr(),
                                              db_.data_ptr(), y.data_ptr(), None, B, Cout, K, T, t_out, t_out, 1, pad, 1, 0,
                                              1, 0, _stream())
                    if rc != 0:
                        outs = []
                        break
                    torch.cuda.synchronize()
                    outs.append(y.cpu().numpy().copy())
                if not outs:
                    continue
                assert np.isfinite(outs[0]).all(), (v, ks)
                assert np.array_equal(outs[0], outs[1]), (v, ks, np.abs(outs[0] - outs[1]).max())
                assert np.abs(outs[0] - ref).max() / scale <= OP_TOL, (v, ks)
                compared += 1
    finally:
        lib.wun_op_force_conv_variant(-1, 0)
        monkeypatch.delenv("WUN_NO_DMA", raising=False)
    assert compared >= 4


def test_every_conv_variant_was_exercised(lib):
    nvar = lib.wun_op_num_conv_variants()
    assert not (set(RETIRED_VARIANTS) & set(_RAN_CONV)), "a retired variant index was launched"
    missing = [v for v in range(nvar) if v not in _RAN_CONV and v not in RETIRED_VARIANTS]
    assert not missing, "conv tile variants never checked: %s" % missing
    kinds = set().union(*_RAN_CONV.values())
    assert kinds == {"stride1", "stride2", "two_source_accum_mask", "strided_output", "fused_two_phase",
                     "copies_expand_stride2", "copies_parity_split_masked", "copies_acc_window"}, kinds
    # the round-6 epilogue features (secondary copies / windowed accumulate) ran on the plain, batch-folded and in-workgroup
    # split-K tile families alike
    for kind in ("copies_expand_stride2", "copies_parity_split_masked", "copies_acc_window"):
        vs = sorted(v for v, k in _RAN_CONV.items() if kind in k)
        assert len(vs) >= 20 and any(34 <= v <= 41 for v in vs) and any(v >= 56 for v in vs), (kind, vs)
    # every variant that can serve the fused two-phase launch (one wave column, even number of column
    # tiles, 8-channel chunks, not batch-folded) was checked there
    fused = sorted(v for v, k in _RAN_CONV.items() if "fused_two_phase" in k)
    assert len(fused) >= 10, fused
    print("[parity] conv variants x kinds:", {v: sorted(k) for v, k in sorted(_RAN_CONV.items())})


WGRAD_CASES = [
    # (B, Cin, Cout, K, T, stride, pad_left, same)
    (2, 24, 48, 15, 700, 1, 0, False),
    (2, 24, 80, 15, 701, 2, 0, False),
    (2, 40, 24, 5, 300, 1, 2, True),
    (16, 48, 56, 15, 95, 2, 0, False),
    (3, 64, 72, 15, 23, 1, 0, False),
]
WGRAD_GEOMS = [(m, n) for m in (1, 2, 4) for n in (1, 2, 3, 4, 5)] + [(6, 1), (6, 2), (6, 3)]


@pytest.mark.parametrize("case", WGRAD_CASES, ids=[str(c) for c in WGRAD_CASES])
def test_wgrad_every_geometry_and_split(lib, case):
    B, Cin, Cout, K, T, stride, pad, same = case
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31) + 11)
    x = rng.uniform(-1, 1, (B, Cin, T)).astype(np.float32)
    t_out = _t_out(T, K, stride, same)
    dz = rng.uniform(-1, 1, (B, Cout, t_out)).astype(np.float32)
    xt = torch.tensor(x, dtype=torch.float64)
    wtn = torch.zeros((K, Cin, Cout), dtype=torch.float64, requires_grad=True)
    need = (t_out - 1) * stride + K - pad
    xp = F.pad(xt, (pad, max(0, need - T)))
    y = F.conv1d(xp, wtn.permute(2, 1, 0), None, stride=stride)[:, :, :t_out]
    (y * torch.tensor(dz, dtype=torch.float64)).sum().backward()
    ref_dw = wtn.grad.numpy()
    ref_db = dz.astype(np.float64).sum(axis=(0, 2))
    sw, sb = max(1.0, np.abs(ref_dw).max()), max(1.0, np.abs(ref_db).max())
    dxg, dzg = _cuda(x), _cuda(dz)
    ran, worst = 0, 0.0
    try:
        for mtw, nw in WGRAD_GEOMS:
            for ns in (0, 1, 3):
                lib.wun_op_force_wgrad_variant(mtw, nw, ns)
                scr = torch.empty(int(lib.wun_op_conv1d_wgrad_scratch(B, Cin, Cout, K, t_out)), device="cuda")
                gdw = torch.full((K, Cin, Cout), float("nan"), device="cuda")
                gdb = torch.full((Cout,), float("nan"), device="cuda")
                rc = lib.wun_op_conv1d_wgrad(dxg.data_ptr(), dzg.data_ptr(), gdw.data_ptr(), gdb.data_ptr(),
                                             scr.data_ptr(), B, Cin, Cout, K, T, t_out, stride, pad, _stream())
                if rc == -2:
                    continue               # geometry not available for this shape: refused, not miscomputed
                _lib.check(rc)
                torch.cuda.synchronize()
                ew = np.abs(gdw.cpu().numpy() - ref_dw).max() / sw
                eb = np.abs(gdb.cpu().numpy() - ref_db).max() / sb
                assert ew <= OP_TOL and eb <= OP_TOL, (mtw, nw, ns, ew, eb)
                worst = max(worst, ew, eb)
                _RAN_WGRAD.add((mtw, nw))
                ran += 1
    finally:
        lib.wun_op_force_wgrad_variant(0, 0, 0)
    assert ran >= 6
    record("wgrad_every_geometry_and_split", str(case), worst, OP_TOL)


def test_every_wgrad_geometry_was_exercised(lib):
    missing = [g for g in WGRAD_GEOMS if g not in _RAN_WGRAD]
    assert not missing, "wgrad_mfma_kernel<MTW, NW> instantiations never checked: %s" % missing


# ---- register-window weight-gradient kernel (wun_wgrad_win.hip): every instantiation <K, NW, S> x split counts ----
WIN_WGRAD_CASES = [
    # (B, Cin, Cout, K, T, stride, pad_left, same)
    (2, 24, 48, 15, 1400, 1, 0, False),
    (2, 24, 80, 15, 1401, 2, 0, False),
    (16, 48, 56, 15, 95, 2, 0, False),
    (3, 64, 72, 15, 23, 1, 7, True),            # same padding: zero columns on both sides of every row
    (2, 40, 24, 5, 300, 1, 2, True),
    (2, 72, 100, 5, 517, 1, 0, False),
]
_RAN_WIN = set()


@pytest.mark.parametrize("case", WIN_WGRAD_CASES, ids=[str(c) for c in WIN_WGRAD_CASES])
def test_window_wgrad_every_instantiation_and_split(lib, case):
    """wun_op_set_wgrad_win(1): the same operator on wgrad_win_kernel<K, NW, S> (+ wgrad_win_reduce_kernel) for every column
    tile count, with the split count chosen by the library, 1, and a target grid of 512 workgroups."""
    B, Cin, Cout, K, T, stride, pad, same = case
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31) + 13)
    x = rng.uniform(-1, 1, (B, Cin, T)).astype(np.float32)
    t_out = _t_out(T, K, stride, same)
    dz = rng.uniform(-1, 1, (B, Cout, t_out)).astype(np.float32)
    xt = torch.tensor(x, dtype=torch.float64)
    wtn = torch.zeros((K, Cin, Cout), dtype=torch.float64, requires_grad=True)
    need = (t_out - 1) * stride + K - pad
    xp = F.pad(xt, (pad, max(0, need - T)))
    y = F.conv1d(xp, wtn.permute(2, 1, 0), None, stride=stride)[:, :, :t_out]
    (y * torch.tensor(dz, dtype=torch.float64)).sum().backward()
    ref_dw = wtn.grad.numpy()
    ref_db = dz.astype(np.float64).sum(axis=(0, 2))
    sw, sb = max(1.0, np.abs(ref_dw).max()), max(1.0, np.abs(ref_db).max())
    dxg, dzg = _cuda(x), _cuda(dz)
    ran, worst = 0, 0.0
    _lib.check(lib.wun_op_set_wgrad_win(1))
    try:
        for nw in range(1, 7 if K == 5 else 6):
            for ns in (0, 1, -512):
                lib.wun_op_force_wgrad_variant(1, nw, ns)
                scr = torch.empty(int(lib.wun_op_conv1d_wgrad_scratch(B, Cin, Cout, K, t_out)), device="cuda")
                gdw = torch.full((K, Cin, Cout), float("nan"), device="cuda")
                gdb = torch.full((Cout,), float("nan"), device="cuda")
                rc = lib.wun_op_conv1d_wgrad(dxg.data_ptr(), dzg.data_ptr(), gdw.data_ptr(), gdb.data_ptr(),
                                             scr.data_ptr(), B, Cin, Cout, K, T, t_out, stride, pad, _stream())
                _lib.check(rc)
                torch.cuda.synchronize()
                ew = np.abs(gdw.cpu().numpy() - ref_dw).max() / sw
                eb = np.abs(gdb.cpu().numpy() - ref_db).max() / sb
                assert ew <= OP_TOL and eb <= OP_TOL, (nw, ns, ew, eb)
                worst = max(worst, ew, eb)
                _RAN_WIN.add((K, min(nw, (Cout + 15) // 16), stride))
                ran += 1
    finally:
        lib.wun_op_force_wgrad_variant(0, 0, 0)
        lib.wun_op_set_wgrad_win(0)
    assert ran >= 15
    record("window_wgrad_every_instantiation_and_split", str(case), worst, OP_TOL)


def test_window_wgrad_refuses_what_it_does_not_serve(lib):
    """Tap counts other than 15 / 5 (and K = 15 with a channel count that is not a multiple of 8) are refused with
    WUN_ERR_UNSUPPORTED while the hook is on -- never computed by another kernel behind the caller's back."""
    _lib.check(lib.wun_op_set_wgrad_win(1))
    try:
        for (Cin, K) in ((24, 9), (20, 15)):
            B, Cout, T = 2, 32, 200
            t_out = T - K + 1
            x = torch.zeros(B, Cin, T, device="cuda"); dz = torch.zeros(B, Cout, t_out, device="cuda")
            dw = torch.zeros(K, Cin, Cout, device="cuda"); db = torch.zeros(Cout, device="cuda")
            scr = torch.empty(max(int(lib.wun_op_conv1d_wgrad_scratch(B, Cin, Cout, K, t_out)), 1), device="cuda")
            rc = lib.wun_op_conv1d_wgrad(x.data_ptr(), dz.data_ptr(), dw.data_ptr(), db.data_ptr(), scr.data_ptr(),
                                         B, Cin, Cout, K, T, t_out, 1, 0, _stream())
            assert rc == -2, (Cin, K, rc)
    finally:
        lib.wun_op_set_wgrad_win(0)


# ---- the direct-reduction weight gradients of the narrow layers (wun_narrow.hip) as single operators ----
NARROW_CASES = [
    # (B, Cin, Cout, K, T, stride, pad_left, same)
    (3, 1, 24, 15, 5000, 2, 0, False),          # mono audio-input conv, decimated positions: streaming form, ragged last unit
    (3, 1, 24, 15, 1303, 1, 0, False),          # ... its skip-window positions
    (2, 1, 24, 15, 777, 1, 7, True),            # same padding: zero samples on both sides
    (2, 1, 20, 9, 600, 1, 0, False),            # fewer rows than the waves hold, taps < 15
    (2, 2, 24, 15, 900, 2, 0, False),           # stereo input: the LDS-staged form
    (2, 25, 2, 1, 1000, 1, 0, False),           # the output head's shape (25 feature channels -> 2 samples)
    (2, 26, 4, 3, 500, 1, 1, True),
]


@pytest.mark.parametrize("case", NARROW_CASES, ids=[str(c) for c in NARROW_CASES])
def test_narrow_wgrad_operator(lib, case):
    B, Cin, Cout, K, T, stride, pad, same = case
    rng = np.random.default_rng(abs(hash(case)) % (2 ** 31) + 17)
    x = rng.uniform(-1, 1, (B, Cin, T)).astype(np.float32)
    t_out = _t_out(T, K, stride, same)
    dz = rng.uniform(-1, 1, (B, Cout, t_out)).astype(np.float32)
    xt = torch.tensor(x, dtype=torch.float64)
    wtn = torch.zeros((K, Cin, Cout), dtype=torch.float64, requires_grad=True)
    need = (t_out - 1) * stride + K - pad
    xp = F.pad(xt, (pad, max(0, need - T)))
    y = F.conv1d(xp, wtn.permute(2, 1, 0), None, stride=stride)[:, :, :t_out]
    (y * torch.tensor(dz, dtype=torch.float64)).sum().backward()
    ref_dw = wtn.grad.numpy()
    ref_db = dz.astype(np.float64).sum(axis=(0, 2))
    sw, sb = max(1.0, np.abs(ref_dw).max()), max(1.0, np.abs(ref_db).max())
    dxg, dzg = _cuda(x), _cuda(dz)
    _lib.check(lib.wun_op_set_wgrad_narrow(1))
    try:
        scr = torch.empty(int(lib.wun_op_conv1d_wgrad_scratch(B, Cin, Cout, K, t_out)), device="cuda")
        gdw = torch.full((K, Cin, Cout), float("nan"), device="cuda")
        gdb = torch.full((Cout,), float("nan"), device="cuda")
        _lib.check(lib.wun_op_conv1d_wgrad(dxg.data_ptr(), dzg.data_ptr(), gdw.data_ptr(), gdb.data_ptr(), scr.data_ptr(),
                                           B, Cin, Cout, K, T, t_out, stride, pad, _stream()))
        torch.cuda.synchronize()
    finally:
        lib.wun_op_set_wgrad_narrow(0)
    ew = np.abs(gdw.cpu().numpy() - ref_dw).max() / sw
    eb = np.abs(gdb.cpu().numpy() - ref_db).max() / sb
    record("narrow_wgrad_operator", str(case), max(ew, eb), OP_TOL)
    assert ew <= OP_TOL and eb <= OP_TOL, (ew, eb)
