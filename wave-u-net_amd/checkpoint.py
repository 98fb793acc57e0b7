"""Checkpoints of a `UnetAudioSeparator`: this package's `.npz` files and the reference's TensorFlow V2 checkpoints.

Reference: `tf.train.Saver(tf.global_variables(), write_version=SaverDef.V2)` saves every variable once per `train()`
call and restores them by name (`Training.py:92-98,113`, `Evaluate.py:55-57`).  The variable set of the reference graph:

  separator/conv1d[_n]/{kernel,bias}, separator/interp_<i>     the model (TF layouts: kernel [K, Cin, Cout])
  global_step                                                   int64 scalar (`Training.py:66`)
  separator_solver/<variable>/Adam, .../Adam_1                   Adam m and v slots: the optimizer is built inside
  separator_solver/beta1_power, separator_solver/beta2_power     tf.variable_scope("separator_solver") (`Training.py:75-77`)

`load_checkpoint` accepts either format (a path ending in `.npz`, or a V2 prefix whose `<prefix>.index` exists) --
so the published weights (`README.md:110-111`) load as they are -- and `save_checkpoint` writes either, so a model
trained here can be restored by the reference's `Saver`.  The TensorFlow format is handled by `tf_checkpoint.py`
(no TensorFlow needed).
"""
import numpy as np
import torch

from . import tf_checkpoint

SOLVER_SCOPE = "separator_solver"


def _tensors(separator):
    plan = separator._any_plan()
    separator._ensure_variables(plan)
    return [(name, off, tuple(shp)) for name, off, shp in plan.tensors]


def _fill(arena, off, shp, value, name):
    t = torch.as_tensor(np.asarray(value), dtype=torch.float32).reshape(-1)
    n = int(np.prod(shp)) if len(shp) else 1
    if t.numel() != n:
        raise ValueError("%s: checkpoint has %d elements, the model %d %s" % (name, t.numel(), n, shp))
    arena[off:off + n].copy_(t)


def load_checkpoint(separator, path, with_optimizer=True):
    """Restore variables (and, if present and asked for, the Adam slots) into `separator`; returns global_step.
    Missing model variables are an error, as with `Saver.restore`."""
    path = str(path)
    tensors = _tensors(separator)
    if tf_checkpoint.is_checkpoint(path):
        ck = tf_checkpoint.read(path)
        missing = [n for n, _, _ in tensors if n not in ck]
        if missing:
            raise KeyError("checkpoint %s lacks %d variables of this model, e.g. %s" % (path, len(missing), missing[0]))
        for name, off, shp in tensors:
            _fill(separator.params, off, shp, ck[name], name)
        slots = all(("%s/%s/Adam" % (SOLVER_SCOPE, n)) in ck and ("%s/%s/Adam_1" % (SOLVER_SCOPE, n)) in ck
                    for n, _, _ in tensors)
        if with_optimizer and slots:
            for name, off, shp in tensors:
                _fill(separator.adam_m, off, shp, ck["%s/%s/Adam" % (SOLVER_SCOPE, name)], name + "/Adam")
                _fill(separator.adam_v, off, shp, ck["%s/%s/Adam_1" % (SOLVER_SCOPE, name)], name + "/Adam_1")
        step = int(ck["global_step"]) if "global_step" in ck else 0
    else:
        state = np.load(path)
        separator.load_variables({k: state[k] for k in state.files if k.startswith("separator/")})
        if with_optimizer and "adam_m" in state.files:
            separator.adam_m.copy_(torch.from_numpy(state["adam_m"]))
            separator.adam_v.copy_(torch.from_numpy(state["adam_v"]))
        step = int(state["global_step"]) if "global_step" in state.files else 0
    separator.global_step = step
    return step


def save_checkpoint(separator, path, fmt=None, beta1=0.9, beta2=0.999):
    """Write the separator's variables, Adam slots and global_step.  fmt "npz" (default for paths ending in .npz) or
    "tf" (a V2 checkpoint prefix the reference restores).  Returns the path."""
    path = str(path)
    if fmt is None:
        fmt = "npz" if path.endswith(".npz") else "tf"
    variables = {n: v.detach().cpu().numpy() for n, v in separator.variables().items()}
    step = int(separator.global_step)
    if fmt == "npz":
        arrays = dict(variables)
        arrays["adam_m"] = separator.adam_m.cpu().numpy()
        arrays["adam_v"] = separator.adam_v.cpu().numpy()
        arrays["global_step"] = np.int64(step)
        np.savez(path, **arrays)
        return path if path.endswith(".npz") else path + ".npz"
    if fmt != "tf":
        raise ValueError("fmt must be 'npz' or 'tf'")
    out = dict(variables)
    out["global_step"] = np.asarray(step, dtype=np.int64)
    m, v = separator.adam_m.cpu().numpy(), separator.adam_v.cpu().numpy()
    for name, off, shp in _tensors(separator):
        n = int(np.prod(shp)) if len(shp) else 1
        out["%s/%s/Adam" % (SOLVER_SCOPE, name)] = m[off:off + n].reshape(shp)
        out["%s/%s/Adam_1" % (SOLVER_SCOPE, name)] = v[off:off + n].reshape(shp)
    # TF's Adam keeps beta ** (number of updates + 1) in two accumulator variables
    out["%s/beta1_power" % SOLVER_SCOPE] = np.asarray(beta1 ** (step + 1), dtype=np.float32)
    out["%s/beta2_power" % SOLVER_SCOPE] = np.asarray(beta2 ** (step + 1), dtype=np.float32)
    return tf_checkpoint.write(path, out)
