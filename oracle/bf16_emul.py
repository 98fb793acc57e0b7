"""ORACLE (test infrastructure): the Wave-U-Net training step with the bf16 mode's STORAGE POINTS restated.

NOT the product; only tests/ may import it.  The float64 oracle (oracle/waveunet_torch.py) is what the reference computes
(/root/reference/Models/UnetAudioSeparator.py:85-144, Training.py:50-63); the bf16 mode of the HIP path
(wun_config.compute_dtype = 1, BASELINE.json configs[2], [4]) deliberately deviates from it by rounding every activation
and activation-gradient tensor to bfloat16 where it is stored, which puts 1e-2 .. 1e-1 of rounding noise on the
gradients -- a tolerance wide enough to hide a real defect of a few per cent.  This file closes that gap: the SAME graph,
written out layer by layer with hand-placed adjoints (as oracle/backward_np.py), in float64, with a round-to-nearest-even
bfloat16 conversion at exactly the points where the HIP plan stores a tensor in HBM:

  forward   every conv output after bias + LeakyReLU (down levels, bottleneck, up levels), the 2x-upsampled tensor;
            conv weights of every conv with >= 8 input channels (the packed bf16 MFMA operand image); the audio, the
            biases, the head's weights and outputs stay fp32
  backward  d(pre-activation) of every conv (LeakyReLU derivative applied BEFORE the store, read from the sign of the stored
            activation), d(upsampled tensor); a down level's input gradient is stored after the decimated part and stored
            AGAIN after the skip-window part was added inside the window (two kernel launches); same-padding: the input
            gradient is added onto the skip gradient already stored at the even positions; weight / bias / interpolation
            gradients are fp32 sums of products of the stored (rounded) operands; the head's d(pre-activation) stays fp32
            (wide heads that run on the bf16 MFMA weight-gradient kernel -- (C + F) * Sh * C > 256 -- round it and the audio)

so that the HIP path can be compared with it at fp32-accumulation error plus rare one-ulp bf16 flips (an fp32 sum that
lands within 1e-7 of a rounding boundary) instead of at the rounding noise of the mode itself.  With `quantize=False`
every rounding is the identity and the result must equal the float64 oracle exactly (tests/test_bf16_emul.py) -- that
pins the hand-placed adjoints; the rounding points themselves are a restatement of wave-u-net_amd/csrc (wun_plan.hip:
wun_forward / wun_loss_backward_ex, wun_bf16.hip epilogue), cited inline.

    loss, grads, inter = train_step(cfg, params, mix, targets)                  # chained
    loss, grads, inter = train_step(cfg, params, mix, targets, forced=gpu_tensors)   # layer by layer (see train_step)
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import shapes


def bf16_round(x):
    """fp32 value -> nearest-even bfloat16, as float64 (what a `st<bf16>` of an fp32 register stores)."""
    return x.to(torch.float32).to(torch.bfloat16).to(torch.float64)


def fp32_round(x):
    return x.to(torch.float32).to(torch.float64)


class _Q:
    """The two roundings, or identities (quantize=False)."""

    def __init__(self, on):
        self.on = on

    def a(self, x):            # a tensor stored in HBM as bf16
        return bf16_round(x) if self.on else x

    def f(self, x):            # a tensor stored in HBM as fp32
        return fp32_round(x) if self.on else x


def _conv(x, w_kcn, b, same, stride=1):
    """tf.layers.conv1d (UnetAudioSeparator.py:98): cross-correlation, kernel [K, Cin, Cout]; `stride` = 2 is the conv
    followed by [:, ::2, :] (:100)."""
    K = w_kcn.shape[0]
    if same:
        left = (K - 1) // 2
        x = F.pad(x, (left, K - 1 - left))
    return F.conv1d(x, w_kcn.permute(2, 1, 0), b, stride=stride)


def _conv_adjoint(x, w_kcn, dz, same, stride=1, need_dx=True):
    """(dx or None, dw, db) of _conv for the output gradient dz -- through autograd of this ONE op (exact float64)."""
    xr = x.detach().clone().requires_grad_(need_dx)
    wr = w_kcn.detach().clone().requires_grad_(True)
    with torch.enable_grad():
        y = _conv(xr, wr, None, same, stride)
        assert y.shape == dz.shape, (tuple(y.shape), tuple(dz.shape))
        g = torch.autograd.grad(y, [xr, wr] if need_dx else [wr], dz)
    dx, dw = (g[0], g[1]) if need_dx else (None, g[0])
    return dx, dw, dz.sum(dim=(0, 2))


def _mask(stored_act):
    """LeakyReLU derivative as the kernels take it: 1 where the STORED post-activation value is > 0, else 0.2
    (wun_bf16.hip epilogue, upsample_bwd_vec_kernel; Utils.py:79-80 with TF's tie rule)."""
    return torch.where(stored_act > 0, torch.ones_like(stored_act), torch.full_like(stored_act, 0.2))


def _lrelu(pre):
    return torch.maximum(0.2 * pre, pre)


def _upsample(x, w, context, q):
    """upsample_vec_kernel: even outputs copy, odd outputs 0.5 (x[j] + x[j+1]) / sg x[j] + (1 - sg) x[j+1] in fp32
    (UnetAudioSeparator.py:109-118, InterpolationLayer.py:19-39)."""
    n = x.shape[2]
    if w is None:
        if context:
            mid = 0.5 * q.f(x[:, :, :-1] + x[:, :, 1:])
        else:
            xr = torch.cat([x[:, :, 1:], x[:, :, -1:]], dim=2)        # legacy bilinear clamps
            mid = 0.5 * q.f(x + xr)
    else:
        sg = q.f(torch.sigmoid(w)).view(1, -1, 1)
        xr = x[:, :, 1:] if context else F.pad(x[:, :, 1:], (0, 1))   # SAME: one zero on the right
        x0 = x[:, :, :-1] if context else x
        mid = q.f(q.f(sg * x0) + q.f(q.f(1.0 - sg) * xr))
    tup = 2 * n - 1 if context else 2 * n
    out = x.new_zeros(x.shape[0], x.shape[1], tup)
    out[:, :, 0::2] = x
    out[:, :, 1::2] = mid
    return out


def _upsample_adjoint(x, w, context, dy, q):
    """upsample_bwd_vec_kernel + interp_grad_kernel: g[i] = dy[2i] + wa dy[2i+1] + wb dy[2i-1] (that order, fp32);
    dw[c] = sg (1 - sg) sum dy[2i+1] (x[i] - x[i+1])."""
    n = x.shape[2]
    tup = dy.shape[2]
    if w is None:
        wa = wb = 0.5
    else:
        wa = q.f(torch.sigmoid(w)).view(1, -1, 1)
        wb = q.f(1.0 - wa)
    g = dy[:, :, 0::2].clone()
    odd = dy[:, :, 1::2]                                              # n - 1 (context) or n (same) mid samples
    if context:
        g[:, :, :-1] = q.f(g[:, :, :-1] + q.f(wa * odd))
        g[:, :, 1:] = q.f(g[:, :, 1:] + q.f(wb * odd))
    else:
        wgt = wa
        if w is None:                                                 # same-mode legacy clamp: out[2n-1] = x[n-1]
            wgt = torch.full((1, 1, n), 0.5, dtype=dy.dtype)
            wgt[:, :, -1] = 1.0
        g = q.f(g + q.f(wgt * odd))
        g[:, :, 1:] = q.f(g[:, :, 1:] + q.f(wb * odd[:, :, :-1]))
    dw = None
    if w is not None:
        sg = torch.sigmoid(w)
        xr = x[:, :, 1:] if context else F.pad(x[:, :, 1:], (0, 1))
        x0 = x[:, :, :-1] if context else x
        nmid = tup // 2
        dw = (odd[:, :, :nmid] * (x0 - xr)[:, :, :nmid]).sum(dim=(0, 2)) * sg * (1.0 - sg)
    return g, dw


def head_on_mfma(cfg):
    """wun_plan_create: the output layer's weight gradient runs on the bf16 MFMA kernel (bf16 copies of the audio and of
    d(pre-activation)) when the direct-reduction kernel cannot hold its (input channel, output row) pairs."""
    cfg = shapes.finalize_config(cfg)
    C, Fi = cfg["num_channels"], cfg["num_initial_filters"]
    Sh = cfg["num_sources"] - (1 if cfg["output_type"] == "difference" else 0)
    return (C + Fi) * Sh * C > 256 and C % 2 == 0


def train_step(cfg, params, mix_btc, targets, quantize=True, forced=None):
    """Forward + loss + hand-placed reverse pass with the bf16 plan's storage roundings.
    Returns (loss float, [gradient tensors (float64) in variable order], {name: tensor as COMPUTED here}).

    Stored tensors and their names (NCW): "dec<i>" / "skip<i>" (down level i: the decimated stream; the window the skip
    connection crops -- the whole row with same padding), "bottleneck", "ups<j>" / "up<j>" (up level j: its upsampled
    input, its output); "dz_up<j>", "d_ups<j>", "dz_skip<i>", "dz_dec<i>" (context), "dz_bottleneck" (the gradients
    w.r.t. the pre-activations / the upsampled tensor) -- wun_plan_activation's kinds 0 - 9.

    forced: {name: tensor}.  LAYER-BY-LAYER mode: every tensor named there is still computed (and returned) but REPLACED
    by the given one for everything downstream, so each computed tensor -- and each weight gradient -- is a function of
    the given tensors its producing launch read.  Comparing computed with given is then sharp at every depth; chained
    (forced=None) the rare one-ulp flips of one layer seed flips in the next (the rounding of a value that moved by d
    crosses a boundary with probability d / ulp and then moves it by a whole ulp: rms sqrt(d ulp) > d), and after four or
    five stored tensors two valid executions differ by the rounding noise of the mode itself."""
    cfg = shapes.finalize_config(cfg)
    q = _Q(quantize)
    L, same, ctx = cfg["num_layers"], not cfg["context"], bool(cfg["context"])
    names = cfg["source_names"]
    P = [torch.as_tensor(np.asarray(v), dtype=torch.float64) for _, v in params]
    G = [None] * len(P)
    it = iter(range(len(P)))
    x_in = torch.as_tensor(np.asarray(mix_btc), dtype=torch.float64).permute(0, 2, 1).contiguous()
    inter = {}

    inter["_scale"] = {}

    def store(name, value, scale=None):
        # scale: magnitude of the operands of a store that ADDS onto an earlier store (the earlier one may itself have landed
        # on either side of a rounding boundary: the sum is then off by an ulp of the OPERAND, many ulps of a cancelling sum)
        inter[name] = value
        if scale is not None:
            inter["_scale"][name] = scale
        if forced is not None and name in forced:
            f = torch.as_tensor(forced[name]).to(torch.float64)
            assert f.shape == value.shape, (name, tuple(f.shape), tuple(value.shape))
            return f
        return value

    def wq(w, cin):
        # conv_dispatch (wun_plan.hip): fewer than 8 input channels = the audio-input conv, a direct fp32 conv;
        # everything else reads the packed bf16 weight image
        return q.a(w) if cin >= 8 else w

    # ---------------- forward (wun_forward) ----------------
    cur = x_in
    tape = []
    acts, decs = [], []
    for i in range(L):                                              # UnetAudioSeparator.py:97-100
        ik, ib = next(it), next(it)
        w = wq(P[ik], cur.shape[1])
        act = q.a(_lrelu(_conv(cur, w, P[ib], same)))               # full rate; the plan keeps the two views below
        tape.append((ik, ib, cur, w))
        acts.append(act)
        cur = store("dec%d" % i, act[:, :, ::2])
        decs.append(cur)
    ik, ib = next(it), next(it)                                     # :102
    w = wq(P[ik], cur.shape[1])
    bott_tape = (ik, ib, cur, w)
    cur = store("bottleneck", q.a(_lrelu(_conv(cur, w, P[ib], same))))
    up_tape = []
    skcs = [None] * L
    for j in range(L):                                              # :107-125
        i = L - 1 - j
        iw = next(it) if cfg["upsampling"] == "learned" else None
        ups = store("ups%d" % j, q.a(_upsample(cur, None if iw is None else P[iw], ctx, q)))
        s0, s1 = shapes.crop_offsets(acts[i].shape[2], ups.shape[2])
        skc = store("skip%d" % i, acts[i][:, :, s0:acts[i].shape[2] - s1])
        skcs[i] = (s0, skc)
        cat = torch.cat([skc, ups], dim=1)                          # Utils.py:23-24: [skip, current]
        ik, ib = next(it), next(it)
        w = wq(P[ik], cat.shape[1])
        up_tape.append((ik, ib, cat, w, iw, cur, skc.shape[1]))
        cur = store("up%d" % j, q.a(_lrelu(_conv(cat, w, P[ib], same))))
    c0, c1 = shapes.crop_offsets(x_in.shape[2], cur.shape[2])       # :127
    xin_c = x_in[:, :, c0:x_in.shape[2] - c1]
    feat = torch.cat([xin_c, cur], dim=1)
    tanh_act = cfg["output_activation"] == "tanh"
    direct = cfg["output_type"] == "direct"
    heads, outs = [], {}
    total = 0.0
    for nme in (names if direct else names[:-1]):                   # OutputLayer.py:5-9 / 11-23 (fp32 weights)
        ik, ib = next(it), next(it)
        pre = _conv(feat, P[ik], P[ib], same)
        o = q.f(torch.tanh(pre) if tanh_act else pre)               # training: AudioClip is the identity
        heads.append((ik, ib, o))
        outs[nme] = o
        total = total + o
    if not direct:
        d0, d1 = shapes.crop_offsets(xin_c.shape[2], total.shape[2])
        outs[names[-1]] = q.f(xin_c[:, :, d0:xin_c.shape[2] - d1] - total)
    S = len(names)
    loss = 0.0
    dout = {}
    for nme in names:                                               # Training.py:50-63
        tgt = torch.as_tensor(np.asarray(targets[nme]), dtype=torch.float64).permute(0, 2, 1)
        diff = outs[nme] - tgt
        loss += float((diff ** 2).mean()) / S
        dout[nme] = 2.0 * diff / (diff.numel() * S)
    inter["outputs"] = {n: v.permute(0, 2, 1) for n, v in outs.items()}

    # ---------------- reverse (wun_loss_backward_ex) ----------------
    wide = quantize and head_on_mfma(cfg)
    nC = xin_c.shape[1]
    dfeat = torch.zeros_like(feat)
    for k, (ik, ib, o) in enumerate(heads):
        g = dout[names[k]].clone()
        if not direct:
            g = g - dout[names[-1]]                                 # last = cropped mix - sum(others)
        if tanh_act:
            g = g * (1.0 - o * o)
        g = q.f(g)                                                  # d(pre-activation) of the head: fp32 in HBM
        dx, _, _ = _conv_adjoint(feat, P[ik], g, same)              # head_dfeat_kernel: fp32 weights x fp32 gradient
        dfeat += dx
        if wide:
            xh = torch.cat([q.a(xin_c), feat[:, nC:, :]], dim=1)
            _, G[ik], G[ib] = _conv_adjoint(xh, P[ik], q.a(g), same, need_dx=False)
        else:
            _, G[ik], G[ib] = _conv_adjoint(feat, P[ik], g, same, need_dx=False)
    dz = store("dz_up%d" % (L - 1), q.a(_mask(cur) * dfeat[:, nC:, :]))
    dz_skip = [None] * L
    for j in range(L - 1, -1, -1):
        ik, ib, cat, w, iw, xprev, cskip = up_tape[j]
        i = L - 1 - j
        dcat, G[ik], G[ib] = _conv_adjoint(cat, w, dz, same)
        first = q.a(_mask(cat[:, :cskip, :]) * dcat[:, :cskip, :])  # up-conv epilogue, destination 0 (mask of skip[i])
        # (same padding: the level below later adds its input gradient at the even positions -- only the sum is observable)
        dz_skip[i] = first if same else store("dz_skip%d" % i, first)
        d_ups = store("d_ups%d" % j, q.a(dcat[:, cskip:, :]))       # destination 1: no mask
        gprev, dw = _upsample_adjoint(xprev, None if iw is None else P[iw], ctx, d_ups, q)
        if iw is not None:
            G[iw] = dw
        dz = store("dz_bottleneck" if j == 0 else "dz_up%d" % (j - 1), q.a(_mask(xprev) * gprev))
    ik, ib, xb, w = bott_tape                                       # bottleneck
    dcur, G[ik], G[ib] = _conv_adjoint(xb, w, dz, same)
    if same:
        # the input gradient of the level below lands on the even positions of dz_skip[i], which already holds the skip
        # connection's share (conv epilogue: mask, + old, one store)
        for i in range(L - 1, -1, -1):
            ik, ib, xi, w = tape[i]
            acc = dz_skip[i].clone()
            m = _mask(skcs[i][1])[:, :, ::2]
            mag = acc.abs()
            mag[:, :, ::2] = torch.maximum(mag[:, :, ::2], (m * dcur).abs())
            acc[:, :, ::2] = q.a(m * dcur + acc[:, :, ::2])
            dzs = store("dz_skip%d" % i, acc, mag)
            dcur, G[ik], G[ib] = _conv_adjoint(xi, w, dzs, True, need_dx=(i > 0))
        assert all(g is not None for g in G)
        return loss, G, inter
    dzd = store("dz_dec%d" % (L - 1), q.a(_mask(decs[L - 1]) * dcur))
    for i in range(L - 1, -1, -1):
        ik, ib, xi, w = tape[i]
        s0, dzs = skcs[i][0], dz_skip[i]
        K = w.shape[0]
        tc = dzs.shape[2]
        xwin = xi[:, :, s0:s0 + tc + K - 1]
        g1, dw1, db1 = _conv_adjoint(xi, w, dzd, False, stride=2, need_dx=(i > 0))      # decimated positions
        g2, dw2, db2 = _conv_adjoint(xwin, w, dzs, False, need_dx=(i > 0))              # skip-window positions
        G[ik], G[ib] = dw1 + dw2, db1 + db2
        if i > 0:
            m = _mask(decs[i - 1])
            st = q.a(m * g1)                                        # launch 1: fused two-phase transposed conv, stored
            mag = st.abs()
            win = slice(s0, s0 + tc + K - 1)
            mag[:, :, win] = torch.maximum(mag[:, :, win], (m[:, :, win] * g2).abs())
            st[:, :, win] = q.a(m[:, :, win] * g2 + st[:, :, win])  # launch 2: F_ACCUM inside the skip window
            dzd = store("dz_dec%d" % (i - 1), st, mag)
    assert all(g is not None for g in G)
    return loss, G, inter
