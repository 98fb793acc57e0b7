"""ORACLE (test infrastructure): torch-CPU restatement of the Wave-U-Net hot path.

NOT the product.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg
may import this file.  It is the checker the HIP path is compared with, and the timed
"reference CPU path" (TensorFlow 1.8 itself cannot be installed here).

PARITY PINNING: the reference ships no tests / golden vectors / checkpoints for this
path ("parity unpinned" by the reference's own tests).  What this oracle IS pinned
against: (1) get_padding answers produced by executing the reference's own
get_padding; (2) forward outputs produced by executing the reference's own, unmodified
get_output graph code on the numpy TF-1.8 op shim (oracle/tf1_shim) -- see
oracle/make_golden.py and tests/golden/.  The per-op TF-1.8 arithmetic in the shim is
itself a restatement of TF's published semantics, so the pin is structural
(graph wiring, shapes, crop/concat order, interleave, head) plus an independent second
implementation of the arithmetic, not a bit-level TensorFlow comparison.

Each function cites the reference lines it follows (paths relative to /root/reference).
"""
import math
import numpy as np
import torch
import torch.nn.functional as F

from . import shapes


# ---------------------------------------------------------------------------
# parameters
# ---------------------------------------------------------------------------
def init_params(cfg, seed=1337):
    """Glorot-uniform kernels, zero biases, glorot interp vectors (TF defaults of
    tf.layers.conv1d / tf.get_variable; UnetAudioSeparator.py:98,102,123,
    InterpolationLayer.py:19, OutputLayer.py:8,15).  Returns [(name, float32 ndarray)]
    in TF creation order."""
    rng = np.random.default_rng(seed)
    out = []
    for name, shp in shapes.variable_table(cfg):
        if name.endswith("/bias"):
            val = np.zeros(shp, dtype=np.float32)
        else:
            if len(shp) == 1:
                fan_in = fan_out = shp[0]
            else:
                rf = int(np.prod(shp[:-2]))
                fan_in, fan_out = shp[-2] * rf, shp[-1] * rf
            lim = math.sqrt(6.0 / (fan_in + fan_out))
            val = rng.uniform(-lim, lim, size=shp).astype(np.float32)
        out.append((name, val))
    return out


def params_to_torch(params, dtype=torch.float32, requires_grad=False):
    return [(n, torch.tensor(np.asarray(v), dtype=dtype).requires_grad_(requires_grad))
            for n, v in params]


# ---------------------------------------------------------------------------
# ops (all on NCW tensors [B, C, T]; the public functions take/return [B, T, C])
# ---------------------------------------------------------------------------
class _TfLeakyReLU(torch.autograd.Function):
    """Utils.LeakyReLU, Utils.py:79-80: tf.maximum(alpha * x, x) with alpha = 0.2, INCLUDING TensorFlow's
    gradient at the tie.  TF-1.8's _MaximumGrad (math_grad.py, _MaximumMinimumGrad) routes the incoming gradient
    to the FIRST argument where `greater_equal(first, second)` holds and to the second elsewhere; here the first
    argument is alpha*x, so d/dx = alpha wherever alpha*x >= x, i.e. for x <= 0 -- 0.2 at x == 0 exactly
    (torch.maximum would split the tie 0.5/0.5 -> 0.6).  Exact zeros are the common case on real stems: digital
    silence through zero-initialised biases (Datasets.py:188-216 feeds MUSDB vocals)."""

    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return torch.maximum(0.2 * x, x)

    @staticmethod
    def backward(ctx, g):
        (x,) = ctx.saved_tensors
        return g * torch.where(0.2 * x >= x, torch.full_like(x, 0.2), torch.ones_like(x))


class _PinnedLeakyReLU(torch.autograd.Function):
    """LeakyReLU whose branch is PRESCRIBED: y = x where `pos`, 0.2 x elsewhere; dy/dx = 1 / 0.2 accordingly.  Identical
    to _TfLeakyReLU wherever pos == (x > 0).  Used by the parity tests to evaluate the float64 oracle on the branch
    decisions the fp32 kernels took: a pre-activation within fp32 rounding of 0 (|x| ~ 1e-8) can fall on either side of
    0 in the two precisions, and that one 1-vs-0.2 factor moves every gradient upstream by up to a few 1e-4 of max|g|
     -- with the branches pinned the comparison measures the arithmetic alone.  The forward values
    differ from the free oracle's by at most 0.8 |x| at the flipped elements, i.e. by fp32 rounding."""

    @staticmethod
    def forward(ctx, x, pos):
        ctx.save_for_backward(pos)
        return torch.where(pos, x, 0.2 * x)

    @staticmethod
    def backward(ctx, g):
        (pos,) = ctx.saved_tensors
        return g * torch.where(pos, torch.ones_like(g), torch.full_like(g, 0.2)), None


def leaky_relu(x, pin=None):
    """Utils.LeakyReLU, Utils.py:79-80: max(0.2*x, x); gradient 0.2 at x == 0 as in TensorFlow.
    pin = (pos, known) bool tensors of x's shape: where `known`, the branch is taken from `pos` instead of from x."""
    if pin is None:
        return _TfLeakyReLU.apply(x)
    pos, known = pin
    return _PinnedLeakyReLU.apply(x, torch.where(known, pos, x > 0))


def conv1d_tf(x, kernel, bias, same):
    """tf.layers.conv1d: cross-correlation, kernel [K, Cin, Cout]; 'same' pads
    (K-1)//2 left, rest right (UnetAudioSeparator.py:98)."""
    K = kernel.shape[0]
    if same:
        left = (K - 1) // 2
        x = F.pad(x, (left, K - 1 - left))
    w = kernel.permute(2, 1, 0)  # [Cout, Cin, K]; torch conv1d is cross-correlation too
    return F.conv1d(x, w, bias)


def crop(x, t_to):
    """Utils.crop, Utils.py:104-123."""
    start, end = shapes.crop_offsets(x.shape[2], t_to)
    if start == 0 and end == 0:
        return x
    return x[:, :, start:x.shape[2] - end]


def upsample_linear(x, context):
    """tf.image.resize_bilinear as called at UnetAudioSeparator.py:115 (context:
    align_corners=True, n -> 2n-1) and :117 (legacy align_corners=False, n -> 2n,
    last sample clamps).  Hand-rolled: torch's interpolate uses half-pixel centres."""
    n = x.shape[2]
    mid = 0.5 * (x[:, :, :-1] + x[:, :, 1:])
    if context:
        out = x.new_zeros(x.shape[0], x.shape[1], 2 * n - 1)
        out[:, :, 0::2] = x
        out[:, :, 1::2] = mid
    else:
        out = x.new_zeros(x.shape[0], x.shape[1], 2 * n)
        out[:, :, 0::2] = x
        out[:, :, 1:-1:2] = mid
        out[:, :, -1] = x[:, :, -1]
    return out


def upsample_learned(x, w, context):
    """InterpolationLayer.learned_interpolation_layer, InterpolationLayer.py:4-40:
    mid = sigmoid(w)*x[t] + (1-sigmoid(w))*x[t+1]; 'same' pads one zero on the right."""
    n = x.shape[2]
    a = torch.sigmoid(w).view(1, -1, 1)
    if context:
        mid = a * x[:, :, :-1] + (1.0 - a) * x[:, :, 1:]
        out = x.new_zeros(x.shape[0], x.shape[1], 2 * n - 1)
        out[:, :, 0::2] = x
        out[:, :, 1::2] = mid
    else:
        xr = F.pad(x, (0, 1))
        mid = a * xr[:, :, :-1] + (1.0 - a) * xr[:, :, 1:]
        out = x.new_zeros(x.shape[0], x.shape[1], 2 * n)
        out[:, :, 0::2] = x
        out[:, :, 1::2] = mid
    return out


def audio_clip(x, training):
    """Utils.AudioClip, Utils.py:82-92."""
    return x if training else torch.clamp(x, -1.0, 1.0)


# ---------------------------------------------------------------------------
# forward / loss / optimizer
# ---------------------------------------------------------------------------
def get_output(cfg, tparams, mix_btc, training, return_intermediates=False, pins=None):
    """UnetAudioSeparator.get_output, UnetAudioSeparator.py:85-144.

    mix_btc: torch [B, T, C]; tparams: list of (name, tensor) in TF creation order.
    Returns dict source_name -> [B, Tout, C].
    pins: optional {"down<i>" | "down<i>/dec" | "bottleneck" | "up<i>": (pos, known)} -- prescribed LeakyReLU branches
    (leaky_relu); "down<i>/dec" pins the copy of down level i's output that the decimation reads separately."""
    cfg = shapes.finalize_config(cfg)
    L = cfg["num_layers"]
    same = not cfg["context"]
    it = iter(tparams)

    def nxt():
        return next(it)[1]

    def pin(name):
        return None if pins is None else pins.get(name)

    x_in = mix_btc.permute(0, 2, 1)           # NCW
    cur = x_in
    enc = []
    inter = {}
    for i in range(L):                         # :97-100
        k, b = nxt(), nxt()
        pre = conv1d_tf(cur, k, b, same)
        cur = leaky_relu(pre, pin("down%d" % i))
        enc.append(cur)
        inter["down%d" % i] = cur
        if pin("down%d/dec" % i) is not None:
            # the decimated stream pinned separately from the skip tensor (two kernel launches with different
            # summation orders produce them; tests/test_gpu_parity.py:_gpu_pins)
            cur = leaky_relu(pre, pin("down%d/dec" % i))
        cur = cur[:, :, ::2]
    k, b = nxt(), nxt()                        # :102
    cur = leaky_relu(conv1d_tf(cur, k, b, same), pin("bottleneck"))
    inter["bottleneck"] = cur
    for i in range(L):                         # :107-125
        if cfg["upsampling"] == "learned":
            cur = upsample_learned(cur, nxt(), cfg["context"])
        else:
            cur = upsample_linear(cur, cfg["context"])
        skip = enc[-i - 1]
        assert skip.shape[2] == cur.shape[2] or cfg["context"]          # :121
        cur = torch.cat([crop(skip, cur.shape[2]), cur], dim=1)        # Utils.py:23-24
        k, b = nxt(), nxt()
        cur = leaky_relu(conv1d_tf(cur, k, b, same), pin("up%d" % i))
        inter["up%d" % i] = cur
    feat = torch.cat([crop(x_in, cur.shape[2]), cur], dim=1)            # :127

    if cfg["output_activation"] == "tanh":      # :131-136
        act = torch.tanh
    elif cfg["output_activation"] == "linear":
        act = lambda v: audio_clip(v, training)
    else:
        raise NotImplementedError

    outs = {}
    names = cfg["source_names"]
    if cfg["output_type"] == "direct":          # OutputLayer.py:5-9
        for nme in names:
            k, b = nxt(), nxt()
            outs[nme] = act(conv1d_tf(feat, k, b, same))
    elif cfg["output_type"] == "difference":    # :141 + OutputLayer.py:11-23
        cropped_in = crop(x_in, feat.shape[2])
        total = 0
        for nme in names[:-1]:
            k, b = nxt(), nxt()
            o = act(conv1d_tf(feat, k, b, same))
            outs[nme] = o
            total = total + o
        last = crop(cropped_in, total.shape[2]) - total
        outs[names[-1]] = audio_clip(last, training)
    else:
        raise NotImplementedError
    res = {n: v.permute(0, 2, 1) for n, v in outs.items()}
    if return_intermediates:
        return res, inter
    return res


def separator_loss(cfg, outputs, targets):
    """Training.py:50-63: sum over sources of mean((real-est)^2), / num_sources."""
    cfg = shapes.finalize_config(cfg)
    loss = 0
    for n in cfg["source_names"]:
        loss = loss + torch.mean((targets[n] - outputs[n]) ** 2)
    return loss / float(cfg["num_sources"])


def tf_adam_step(params, grads, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """tf.train.AdamOptimizer (Training.py:77) update rule, `step` is 1-based:
    lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m,v moments; theta -= lr_t*m/(sqrt(v)+eps)
    (epsilon added OUTSIDE the bias correction -- not torch.optim.Adam).  In place on
    lists of tensors."""
    lr_t = lr * math.sqrt(1.0 - beta2 ** step) / (1.0 - beta1 ** step)
    with torch.no_grad():
        for p, g, mm, vv in zip(params, grads, m, v):
            mm.mul_(beta1).add_(g, alpha=1.0 - beta1)
            vv.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
            p.sub_(lr_t * mm / (vv.sqrt() + eps))


def train_step(cfg, tparams, mix_btc, targets, m=None, v=None, step=1, lr=1e-4):
    """One `sess.run([separator_solver, ...])` of Training.py:105: forward, loss,
    backward, (optional) Adam.  Returns (loss, grads list)."""
    for _, p in tparams:
        if p.grad is not None:
            p.grad = None
    outs = get_output(cfg, tparams, mix_btc, True)
    loss = separator_loss(cfg, outs, targets)
    loss.backward()
    grads = [p.grad for _, p in tparams]
    if m is not None:
        tf_adam_step([p for _, p in tparams], grads, m, v, step, lr)
    return loss.detach(), grads


# ---------------------------------------------------------------------------
# synthetic data: band-limited noise sources, mix = sum of sources
# (Utils.random_amplify re-sums the mix, Utils.py:35), targets centre-cropped
# (Utils.crop_sample, Utils.py:38-42; Datasets.py:207)
# ---------------------------------------------------------------------------
def synthetic_batch(cfg, batch, t_in, t_out, seed=1337):
    cfg = shapes.finalize_config(cfg)
    S, C = cfg["num_sources"], cfg["num_channels"]
    rng = np.random.default_rng(seed)
    srcs = []
    kern = np.ones(9, dtype=np.float64) / 9.0
    for _ in range(S):
        w = rng.uniform(-1.0, 1.0, size=(batch, t_in + 8, C))
        sm = np.zeros((batch, t_in, C))
        for k in range(9):
            sm += kern[k] * w[:, k:k + t_in, :]
        sm *= (0.9 / S) / max(1e-9, np.abs(sm).max())
        srcs.append(sm.astype(np.float32))
    mix = np.sum(np.stack(srcs, 0), axis=0).astype(np.float32)
    pad = (t_in - t_out) // 2
    assert (t_in - t_out) % 2 == 0
    targets = {n: (s[:, pad:t_in - pad, :] if pad > 0 else s).copy()
               for n, s in zip(cfg["source_names"], srcs)}
    return mix, targets


def chunked_train_step(cfg, named_params, mix_btc, targets, dtype=torch.float32, chunk=1, want_outputs=False,
                       timings=None, pins=None):
    """Loss and gradients of a whole batch, computed `chunk` excerpts at a time so that a full-size
    batch (16 x 147443 samples) never holds more than one chunk's autograd graph in host memory.
    The loss is a mean over excerpts (Training.py:62), so batch loss / gradient = mean of the
    chunk losses / gradients (equal chunk sizes are required).  named_params: [(tf_name, ndarray)].
    Returns (loss float, [grad tensors in variable order], {source: [B,Tout,C]} or None).
    `timings`, if a list, receives the wall time of each chunk's forward+backward.
    pins: prescribed LeakyReLU branches of the whole batch (get_output), sliced per chunk."""
    import time
    cfg = shapes.finalize_config(cfg)
    B = mix_btc.shape[0]
    assert B % chunk == 0, (B, chunk)
    tp = params_to_torch(named_params, dtype, requires_grad=True)
    acc, loss_sum, outs = None, 0.0, ({n: [] for n in cfg["source_names"]} if want_outputs else None)
    for lo in range(0, B, chunk):
        t0 = time.time()
        tmix = torch.as_tensor(mix_btc[lo:lo + chunk]).to(dtype)
        ttg = {k: torch.as_tensor(v[lo:lo + chunk]).to(dtype) for k, v in targets.items()}
        for _, p in tp:
            p.grad = None
        cp = None if pins is None else {k: (v[0][lo:lo + chunk], v[1][lo:lo + chunk]) for k, v in pins.items()}
        o = get_output(cfg, tp, tmix, True, pins=cp)
        loss = separator_loss(cfg, o, ttg)
        loss.backward()
        if timings is not None:
            timings.append(time.time() - t0)
        loss_sum += float(loss.detach())
        g = [p.grad.detach().double() for _, p in tp]
        acc = g if acc is None else [a + b for a, b in zip(acc, g)]
        if want_outputs:
            for n in cfg["source_names"]:
                outs[n].append(o[n].detach())
    n = B // chunk
    if want_outputs:
        outs = {k: torch.cat(v, dim=0) for k, v in outs.items()}
    return loss_sum / n, [a / n for a in acc], outs
