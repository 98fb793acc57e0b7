"""Host-side mirror of the reference's separator object for the hot path.

Same constructor / get_padding / get_output surface as
/root/reference/Models/UnetAudioSeparator.py:9-144 (what Training.py:29-47, Test.py:15-34 and
Evaluate.py:28-47 call), driving the gfx950 kernels in libwun.so through the C ABI of
include/wun.h.  Because there is no tf.gradients here, the object additionally exposes
`loss_and_gradients` (Training.py:50-63 + the backward implied by :77) and `adam_step`
(tf.train.AdamOptimizer, Training.py:77).

torch is used for device memory, streams and torch.distributed only.
"""
import ctypes as C
import math

import numpy as np
import torch

from . import _lib
from .config import finalize

_UPS = {"linear": 0, "learned": 1}
_OUT = {"direct": 0, "difference": 1}
_ACT = {"tanh": 0, "linear": 1}
_DTYPE = {"f32": 0, "bf16": 1}      # extension key model_config["compute_dtype"] (no reference counterpart)


def _wun_config(cfg):
    if cfg.get("compute_dtype", "f32") not in _DTYPE:
        raise NotImplementedError("compute_dtype=%r" % (cfg["compute_dtype"],))
    for key, table in (("upsampling", _UPS), ("output_type", _OUT), ("output_activation", _ACT)):
        if cfg[key] not in table:
            raise NotImplementedError("%s=%r" % (key, cfg[key]))   # UnetAudioSeparator.py:136,144
    return _lib.WunConfig(
        cfg["num_layers"], cfg["num_initial_filters"], cfg["filter_size"], cfg["merge_filter_size"],
        cfg["input_filter_size"], cfg["output_filter_size"], _UPS[cfg["upsampling"]],
        _OUT[cfg["output_type"]], 1 if cfg["context"] else 0, len(cfg["source_names"]),
        1 if cfg["mono_downmix"] else 2, _ACT[cfg["output_activation"]], _DTYPE[cfg.get("compute_dtype", "f32")],
        1 if cfg.get("exclusive_streams", False) else 0)      # extension key: scheduling hint (wun.h), set by the Trainer


class _Plan(object):
    def __init__(self, lib, wcfg, batch, frames):
        self.lib = lib
        self.handle = C.c_void_p()
        _lib.check(lib.wun_plan_create(C.byref(wcfg), batch, frames, C.byref(self.handle)))
        if wcfg.exclusive_streams:
            if torch.distributed.is_available() and torch.distributed.is_initialized():
                import warnings
                warnings.warn("wun: a plan with exclusive_streams=1 (lowest-priority side streams) was created while a "
                              "process group is initialised -- measured ~40 % slower beside a communication stream; pass "
                              "model_config['exclusive_streams'] = False", RuntimeWarning, stacklevel=3)
            _lib.LOW_PRIORITY_PLANS["created"] += 1
        self.info = _lib.WunPlanInfo()
        _lib.check(lib.wun_plan_query(self.handle, C.byref(self.info)))
        if int(self.info.compute_dtype_effective) != int(wcfg.compute_dtype):
            # (num_initial_filters % 8 != 0, a tap-less conv phase, rows beyond the bf16 kernels' offsets: include/wun.h)
            import warnings
            warnings.warn("wun: compute_dtype='bf16' was requested but this configuration does not qualify for the bf16 "
                          "mode -- the plan runs the exact-fp32 kernels (wun_plan_info.compute_dtype_effective = 0)",
                          RuntimeWarning, stacklevel=3)
        self.tensors = []
        for i in range(self.info.num_tensors):
            ti = _lib.WunTensorInfo()
            _lib.check(lib.wun_plan_tensor(self.handle, i, C.byref(ti)))
            self.tensors.append((ti.name.decode(), int(ti.offset),
                                 tuple(int(ti.shape[k]) for k in range(ti.ndim))))

    def __del__(self):
        try:
            if self.handle:
                self.lib.wun_plan_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class UnetAudioSeparator(object):
    """U-Net separator network on raw waveforms (UnetAudioSeparator.py:9-13)."""

    def __init__(self, model_config, device=None, seed=1337):
        cfg = finalize(model_config) if "source_names" not in model_config else dict(model_config)
        self.model_config = cfg
        self.num_layers = cfg["num_layers"]                      # UnetAudioSeparator.py:20-32
        self.num_initial_filters = cfg["num_initial_filters"]
        self.filter_size = cfg["filter_size"]
        self.merge_filter_size = cfg["merge_filter_size"]
        self.input_filter_size = cfg["input_filter_size"]
        self.output_filter_size = cfg["output_filter_size"]
        self.upsampling = cfg["upsampling"]
        self.output_type = cfg["output_type"]
        self.context = cfg["context"]
        self.padding = "valid" if cfg["context"] else "same"
        self.source_names = list(cfg["source_names"])
        self.num_channels = 1 if cfg["mono_downmix"] else 2
        self.output_activation = cfg["output_activation"]

        self._lib = _lib.load()                                   # raises if libwun.so is missing
        self._wcfg = _wun_config(cfg)
        self._seed = seed
        self._device = torch.device(device) if device is not None else None
        self._plans = {}
        self._active = None          # plan of the last get_output
        self.params = None           # flat float32 arena (TF creation order)
        self.grads = self.adam_m = self.adam_v = None
        self.global_step = 0
        self._ws = {}
        self._outs = {}
        self._last_mix = None

    # ------------------------------------------------------------------ shapes
    def get_padding(self, shape):
        """UnetAudioSeparator.py:34-83.  shape = [batch, desired_output_frames, *]."""
        fin, fout = C.c_int64(), C.c_int64()
        _lib.check(self._lib.wun_get_padding(C.byref(self._wcfg), int(shape[1]), C.byref(fin),
                                             C.byref(fout)))
        b = int(shape[0])
        return (np.array([b, fin.value, self.num_channels], dtype=np.int64),
                np.array([b, fout.value, self.num_channels], dtype=np.int64))

    # ------------------------------------------------------------------ variables
    def _plan(self, batch, frames):
        key = (int(batch), int(frames))
        if key not in self._plans:
            self._plans[key] = _Plan(self._lib, self._wcfg, key[0], key[1])
        return self._plans[key]

    def _dev(self):
        if self._device is None:
            if not torch.cuda.is_available():
                raise RuntimeError("wave-u-net_amd needs an MI355X (no CPU fallback)")
            self._device = torch.device("cuda", torch.cuda.current_device())
        return self._device

    def variable_table(self, batch=1, frames=None):
        """[(tf_name, offset, shape)] in TF creation order."""
        if frames is None:
            frames = int(self.get_padding([batch, 1 if self.context else 2 ** self.num_layers, 0])[0][1])
        return list(self._plan(batch, frames).tensors)

    def _ensure_variables(self, plan):
        if self.params is not None:
            return
        n = int(plan.info.arena_floats)
        host = np.zeros(n, dtype=np.float32)
        rng = np.random.default_rng(self._seed)
        for name, off, shp in plan.tensors:              # glorot-uniform / zero bias (TF defaults)
            size = int(np.prod(shp))
            if name.endswith("/bias"):
                continue
            if len(shp) == 1:
                fan_in = fan_out = shp[0]
            else:
                rf = int(np.prod(shp[:-2]))
                fan_in, fan_out = shp[-2] * rf, shp[-1] * rf
            lim = math.sqrt(6.0 / (fan_in + fan_out))
            host[off:off + size] = rng.uniform(-lim, lim, size=size).astype(np.float32)
        dev = self._dev()
        self.params = torch.from_numpy(host).to(dev)
        self.grads = torch.zeros(n, dtype=torch.float32, device=dev)
        self.adam_m = torch.zeros(n, dtype=torch.float32, device=dev)
        self.adam_v = torch.zeros(n, dtype=torch.float32, device=dev)

    def _any_plan(self):
        """The variable table is the same for every (batch, frames): use any plan, creating a
        minimal one if the separator has not been run yet."""
        if self._active is not None:
            return self._active
        if not self._plans:
            self.variable_table()
        return next(iter(self._plans.values()))

    def variables(self):
        """dict tf_name -> tensor view into the flat arena."""
        plan = self._any_plan()
        self._ensure_variables(plan)
        return {name: self.params[off:off + int(np.prod(shp))].view(*shp)
                for name, off, shp in plan.tensors}

    def gradients(self):
        plan = self._any_plan()
        self._ensure_variables(plan)
        return {name: self.grads[off:off + int(np.prod(shp))].view(*shp)
                for name, off, shp in plan.tensors}

    def load_variables(self, named):
        """named: dict or list of (tf_name, array)."""
        items = named.items() if isinstance(named, dict) else named
        plan = self._any_plan()
        self._ensure_variables(plan)
        index = {name: (off, shp) for name, off, shp in plan.tensors}
        for name, val in items:
            off, shp = index[name]
            t = torch.as_tensor(np.asarray(val), dtype=torch.float32).reshape(-1)
            assert t.numel() == int(np.prod(shp)), (name, t.shape, shp)
            self.params[off:off + t.numel()].copy_(t)

    # ------------------------------------------------------------------ forward
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self._dev()).cuda_stream)

    def get_output(self, input, training, return_spectrogram=False, reuse=True):
        """UnetAudioSeparator.py:85-144.  input: [batch, num_samples, num_channels] float32
        (torch tensor on the GPU, or anything np.asarray accepts).  Returns a dict
        source_name -> [batch, num_out_samples, num_channels] torch tensor (views of one
        buffer that is overwritten by the next call with the same shape)."""
        dev = self._dev()
        if not torch.is_tensor(input):
            input = torch.as_tensor(np.asarray(input, dtype=np.float32))
        mix = input.to(device=dev, dtype=torch.float32).contiguous()
        if mix.dim() != 3 or mix.shape[2] != self.num_channels:
            raise ValueError("input must be [batch, samples, %d]" % self.num_channels)
        plan = self._plan(mix.shape[0], mix.shape[1])
        self._ensure_variables(plan)
        key = (mix.shape[0], mix.shape[1])
        if key not in self._ws:
            self._ws[key] = torch.empty(int(plan.info.workspace_floats), dtype=torch.float32, device=dev)
            self._outs[key] = torch.empty((len(self.source_names), mix.shape[0],
                                           int(plan.info.output_frames), self.num_channels),
                                          dtype=torch.float32, device=dev)
        ws, outs = self._ws[key], self._outs[key]
        _lib.check(self._lib.wun_forward(plan.handle, self.params.data_ptr(), mix.data_ptr(),
                                         ws.data_ptr(), outs.data_ptr(), 1 if training else 0,
                                         self._stream()))
        self._active, self._last_mix, self._last_key = plan, mix, key
        self._last_training = bool(training)
        return {name: outs[i] for i, name in enumerate(self.source_names)}

    # ------------------------------------------------------------------ training step pieces
    def loss_and_gradients(self, targets, bucket_starts=None, bucket_events=None):
        """MSE loss averaged over sources (Training.py:50-63) and its gradient w.r.t. every
        separator variable.  targets: dict source_name -> [B, Tout, C] or a stacked
        [S, B, Tout, C] tensor.  Must follow get_output(training=True).  Returns the loss as
        a 0-dim GPU tensor (no host sync).

        bucket_starts / bucket_events (optional, data parallel): arena offsets in descending
        order and one torch.cuda.Event per bucket; event k is recorded as soon as all gradients
        at offsets >= bucket_starts[k] are final (see include/wun.h, wun_loss_backward_ex)."""
        if self._active is None or not self._last_training:
            raise RuntimeError("call get_output(..., training=True) first")
        dev = self._dev()
        if isinstance(targets, dict):
            tg = torch.stack([torch.as_tensor(targets[n]).to(dev, torch.float32) for n in self.source_names])
        else:
            tg = targets.to(dev, torch.float32)
        tg = tg.contiguous()
        outs = self._outs[self._last_key]
        if tuple(tg.shape) != tuple(outs.shape):
            raise ValueError("targets shape %s != outputs shape %s" % (tuple(tg.shape), tuple(outs.shape)))
        loss = torch.empty((), dtype=torch.float32, device=dev)
        nb = len(bucket_starts) if bucket_starts else 0
        starts = (C.c_int64 * max(nb, 1))(*([int(x) for x in bucket_starts] if nb else [0]))
        events = (C.c_void_p * max(nb, 1))(*([int(e.cuda_event) for e in bucket_events] if nb else [0]))
        _lib.check(self._lib.wun_loss_backward_ex(
            self._active.handle, self.params.data_ptr(), self._last_mix.data_ptr(),
            self._ws[self._last_key].data_ptr(), outs.data_ptr(), tg.data_ptr(),
            self.grads.data_ptr(), loss.data_ptr(), self._stream(), starts, events, nb))
        return loss

    def tune(self, input, targets):
        """Autotune the kernels of this (batch, length) plan on real buffers: one forward +
        backward with per-launch timing of candidate tilings (wun_plan_tune).  Returns the loss."""
        dev = self._dev()
        mix = torch.as_tensor(input).to(device=dev, dtype=torch.float32).contiguous()
        self.get_output(mix, True)                         # allocates plan / workspace / outputs
        if isinstance(targets, dict):
            tg = torch.stack([torch.as_tensor(targets[n]).to(dev, torch.float32) for n in self.source_names])
        else:
            tg = targets.to(dev, torch.float32)
        tg = tg.contiguous()
        loss = torch.empty((), dtype=torch.float32, device=dev)
        _lib.check(self._lib.wun_plan_tune(
            self._active.handle, self.params.data_ptr(), mix.data_ptr(), self._ws[self._last_key].data_ptr(),
            self._outs[self._last_key].data_ptr(), tg.data_ptr(), self.grads.data_ptr(), loss.data_ptr(),
            self._stream()))
        return loss

    def tune_export(self):
        """The tuned per-launch choices of the active plan as text (wun_plan_tune_export)."""
        buf = C.create_string_buffer(1 << 16)
        _lib.check(self._lib.wun_plan_tune_export(self._active.handle, buf, len(buf)))
        return buf.value.decode()

    def tune_import(self, text):
        """Reuse choices exported by a plan of the same config / batch / length (ValueError otherwise)."""
        _lib.check(self._lib.wun_plan_tune_import(self._active.handle, text.encode()))

    def adam_step(self, lr, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=1.0):
        """tf.train.AdamOptimizer(learning_rate=lr) update (Training.py:77) + global_step += 1."""
        self.global_step += 1
        _lib.check(self._lib.wun_adam_step(
            self._active.handle, self.params.data_ptr(), self.grads.data_ptr(),
            self.adam_m.data_ptr(), self.adam_v.data_ptr(), self.global_step, lr, beta1, beta2, eps,
            grad_scale, self._stream()))

    def activation(self, kind, index=0):
        """(tensor view [B, C, frames], t0, tstep) of a forward activation kept in the workspace of the last
        get_output(training=True): kind "dec" / "skip" (down level `index`), "bottleneck", "up" (up conv `index`);
        element j of a row is the post-activation conv output at position t0 + j * tstep (wun_plan_activation).
        After loss_and_gradients also "ups" (upsampled input of up conv `index`), "dz_up", "d_ups", "dz_skip", "dz_dec",
        "dz_bottleneck": the gradient tensors the backward pass left in the workspace (wun.h, kinds 4 - 9)."""
        info = _lib.WunActivationInfo()
        kinds = {"dec": 0, "skip": 1, "bottleneck": 2, "up": 3, "ups": 4, "dz_up": 5, "d_ups": 6, "dz_skip": 7, "dz_dec": 8,
                 "dz_bottleneck": 9}
        _lib.check(self._lib.wun_plan_activation(self._active.handle, kinds[kind], int(index), C.byref(info)))
        B = int(self._active.info.batch)
        ws = self._ws[self._last_key]
        if info.elem_bytes == 2:                                   # bf16 mode: activations live in HBM as bfloat16
            flat = ws[info.offset:].view(torch.bfloat16)[:B * info.batch_stride]
        else:
            flat = ws[info.offset:info.offset + B * info.batch_stride]
        view = flat.view(B, int(info.channels), int(info.pitch))[:, :, :int(info.frames)]
        return view, int(info.t0), int(info.tstep)

    def plan_info(self):
        return self._active.info if self._active is not None else None

    @property
    def effective_dtype(self):
        """'f32' or 'bf16': the arithmetic the active plan really runs (wun_plan_info.compute_dtype_effective) -- a config
        that asks for the bf16 mode without qualifying for it gets the exact-fp32 plan."""
        info = self.plan_info()
        if info is None:
            return None
        return "bf16" if int(info.compute_dtype_effective) == 1 else "f32"
