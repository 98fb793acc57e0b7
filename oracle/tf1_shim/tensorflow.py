"""numpy stand-in for the handful of TensorFlow-1.8 ops the Wave-U-Net reference calls.

TEST INFRASTRUCTURE ONLY (see oracle/README.md).  TensorFlow is not installable in
this environment, so the reference cannot be run as shipped.  This module lets the
reference's *own, unmodified* graph-building code
(/root/reference/Models/UnetAudioSeparator.py, InterpolationLayer.py, OutputLayer.py,
Utils.py) execute eagerly on numpy float64 arrays: `oracle/make_golden.py` puts this
directory first on sys.path, imports the reference modules and calls
`UnetAudioSeparator.get_padding / get_output`.  That pins the model *structure*
(layer order, channel counts, crop/concat order, interleave/gather logic, output
head wiring) to the reference source itself; only the per-op arithmetic below is a
restatement -- of TF-1.8's published op semantics (tensorflow-gpu==1.8.0,
/root/reference/requirements.txt:3), not of anything in /root/reference.

Op semantics encoded here (the "parity traps" of SURVEY.md section 8c):
  * tf.layers.conv1d: NWC input, kernel [K, Cin, Cout], cross-correlation (no tap
    flip), bias add, then activation.  padding='same' (stride 1) pads (K-1)//2 zeros
    left and K-1-(K-1)//2 right.  Variables are created glorot-uniform / zero-bias
    under names conv1d, conv1d_1, ... in call order inside the variable scope.
  * tf.image.resize_bilinear: the TF1 *legacy* kernel (no half-pixel centres):
    scale = in/out, or (in-1)/(out-1) when align_corners and out > 1;
    src = dst*scale; lo = floor(src); hi = min(lo+1, in-1); lerp by src-lo.
  * tf.nn.conv2d: NHWC, filter [fh, fw, Cin, Cout], cross-correlation; SAME pads
    total = f-1 with the extra element at the END (right/bottom).
  * tf.get_variable default initializer = glorot_uniform; for a 1-D shape [F] TF's
    fan computation gives fan_in = fan_out = F, i.e. U(+-sqrt(3/F)).
"""
import contextlib
import numpy as np

float32 = np.float32
float64 = np.float64
int64 = np.int64


class _Shape(object):
    def __init__(self, shp):
        self._s = [int(x) for x in shp]

    def as_list(self):
        return list(self._s)


class Tensor(np.ndarray):
    """ndarray with the two TF-Tensor methods the reference uses."""

    def get_shape(self):
        return _Shape(self.shape)


def _t(x):
    return np.asarray(x, dtype=np.float64).view(Tensor)


# ---------------------------------------------------------------------------
# variable store (creation order == TF creation order; names are TF names)
# ---------------------------------------------------------------------------
class _Store(object):
    def __init__(self):
        self.reset()

    def reset(self, seed=0, preset=None):
        self.scope = []
        self.vars = []          # list of (name, ndarray) in creation order
        self.layer_counts = {}  # per-scope unique-name counters for tf.layers
        self.rng = np.random.RandomState(seed)
        self.preset = dict(preset) if preset else None

    def full(self, name):
        return "/".join(self.scope + [name])

    def create(self, name, shape, kind):
        full = self.full(name)
        if self.preset is not None:
            val = np.asarray(self.preset[full], dtype=np.float64)
            assert list(val.shape) == list(shape), (full, val.shape, shape)
        elif kind == "zeros":
            val = np.zeros(shape, dtype=np.float64)
        else:  # glorot uniform, TF's _compute_fans
            if len(shape) == 1:
                fan_in = fan_out = shape[0]
            elif len(shape) == 2:
                fan_in, fan_out = shape
            else:
                rf = int(np.prod(shape[:-2]))
                fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
            lim = np.sqrt(6.0 / (fan_in + fan_out))
            # draw in float32 so that goldens hold exactly representable fp32 weights
            val = self.rng.uniform(-lim, lim, size=shape).astype(np.float32).astype(np.float64)
        self.vars.append((full, val))
        return _t(val)


_STORE = _Store()


def shim_reset(seed=0, preset=None):
    _STORE.reset(seed, preset)


def shim_variables():
    return list(_STORE.vars)


@contextlib.contextmanager
def variable_scope(name, reuse=None):
    _STORE.scope.append(name)
    try:
        yield
    finally:
        _STORE.scope.pop()


def get_variable(name, shape=None, dtype=None, initializer=None, trainable=True):
    return _STORE.create(name, list(shape), "glorot")


# ---------------------------------------------------------------------------
# ops
# ---------------------------------------------------------------------------
def _same_pad(k):
    left = (k - 1) // 2
    return left, k - 1 - left


class _Layers(object):
    @staticmethod
    def conv1d(inputs, filters, kernel_size, strides=1, activation=None, padding="valid"):
        assert strides == 1
        x = np.asarray(inputs, dtype=np.float64)
        B, T, Cin = x.shape
        scope_key = "/".join(_STORE.scope)
        n = _STORE.layer_counts.get(scope_key, 0)
        _STORE.layer_counts[scope_key] = n + 1
        lname = "conv1d" if n == 0 else "conv1d_%d" % n
        _STORE.scope.append(lname)
        try:
            w = np.asarray(_STORE.create("kernel", [kernel_size, Cin, filters], "glorot"))
            b = np.asarray(_STORE.create("bias", [filters], "zeros"))
        finally:
            _STORE.scope.pop()
        if padding.lower() == "same":
            l, r = _same_pad(kernel_size)
            x = np.pad(x, [(0, 0), (l, r), (0, 0)], mode="constant")
        else:
            assert padding.lower() == "valid"
        Tout = x.shape[1] - kernel_size + 1
        assert Tout >= 1
        y = np.zeros((B, Tout, filters), dtype=np.float64)
        for k in range(kernel_size):
            y += np.matmul(x[:, k:k + Tout, :], w[k])
        y = _t(y + b)
        if activation is not None:
            y = activation(y)
        return _t(y)


layers = _Layers()


class _Image(object):
    @staticmethod
    def resize_bilinear(images, size, align_corners=False):
        x = np.asarray(images, dtype=np.float64)
        B, H, W, C = x.shape
        oh, ow = int(size[0]), int(size[1])

        def axis_interp(arr, axis, n_in, n_out):
            if align_corners and n_out > 1:
                scale = (n_in - 1) / float(n_out - 1)
            else:
                scale = n_in / float(n_out)
            src = np.arange(n_out, dtype=np.float64) * scale
            lo = np.floor(src).astype(np.int64)
            hi = np.minimum(lo + 1, n_in - 1)
            fr = src - lo
            a = np.take(arr, lo, axis=axis)
            b = np.take(arr, hi, axis=axis)
            shp = [1] * arr.ndim
            shp[axis] = n_out
            fr = fr.reshape(shp)
            return a + (b - a) * fr

        y = axis_interp(x, 1, H, oh)
        y = axis_interp(y, 2, W, ow)
        return _t(y)


image = _Image()


class _NN(object):
    @staticmethod
    def sigmoid(x):
        return _t(1.0 / (1.0 + np.exp(-np.asarray(x, dtype=np.float64))))

    @staticmethod
    def conv2d(input, filter, strides, padding):
        assert list(strides) == [1, 1, 1, 1]
        x = np.asarray(input, dtype=np.float64)
        f = np.asarray(filter, dtype=np.float64)
        fh, fw, Cin, Cout = f.shape
        if padding == "SAME":
            ph, pw = fh - 1, fw - 1
            x = np.pad(x, [(0, 0), (ph // 2, ph - ph // 2), (pw // 2, pw - pw // 2), (0, 0)],
                       mode="constant")
        else:
            assert padding == "VALID"
        B, H, W, _ = x.shape
        oh, ow = H - fh + 1, W - fw + 1
        y = np.zeros((B, oh, ow, Cout), dtype=np.float64)
        for i in range(fh):
            for j in range(fw):
                y += np.matmul(x[:, i:i + oh, j:j + ow, :], f[i, j])
        return _t(y)


nn = _NN()
sigmoid = _NN.sigmoid


def diag(v):
    return _t(np.diag(np.asarray(v)))


def expand_dims(x, axis):
    return _t(np.expand_dims(np.asarray(x), axis))


def squeeze(x, axis=None):
    return _t(np.squeeze(np.asarray(x), axis=axis))


def concat(values, axis):
    return _t(np.concatenate([np.asarray(v) for v in values], axis=axis))


def transpose(x, perm):
    return _t(np.transpose(np.asarray(x), perm))


def gather(params, indices):
    return _t(np.asarray(params)[np.asarray(indices, dtype=np.int64)])


def maximum(a, b):
    return _t(np.maximum(np.asarray(a), np.asarray(b)))


def minimum(a, b):
    return _t(np.minimum(np.asarray(a), np.asarray(b)))


def tanh(x):
    return _t(np.tanh(np.asarray(x)))


def trainable_variables():
    class _V(object):
        def __init__(self, n, v):
            self.name = n + ":0"
            self._v = v

        def get_shape(self):
            return _Shape(self._v.shape)
    return [_V(n, v) for n, v in _STORE.vars]
