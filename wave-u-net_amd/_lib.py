"""ctypes binding of libwun.so (include/wun.h).  Fails loudly when the HIP library is
missing -- there is NO CPU fallback in the product path."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, os.environ.get("WUN_LIB", "libwun.so"))


class WunConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in (
        "num_layers", "num_initial_filters", "filter_size", "merge_filter_size",
        "input_filter_size", "output_filter_size", "upsampling", "output_type", "context",
        "num_sources", "num_channels", "output_activation", "compute_dtype", "exclusive_streams")]


class WunPlanInfo(C.Structure):
    _fields_ = [("batch", C.c_int64), ("input_frames", C.c_int64), ("output_frames", C.c_int64),
                ("num_params", C.c_int64), ("arena_floats", C.c_int64),
                ("workspace_floats", C.c_int64), ("num_tensors", C.c_int64),
                ("num_outputs", C.c_int64), ("fwd_flops", C.c_double), ("bwd_flops", C.c_double),
                ("fwd_flops_dense", C.c_double), ("fwd_flops_unique", C.c_double),
                ("bwd_flops_unique", C.c_double), ("compute_dtype_effective", C.c_int64)]


class WunActivationInfo(C.Structure):
    _fields_ = [(n, C.c_int64) for n in ("offset", "batch_stride", "pitch", "channels", "frames", "t0", "tstep", "elem_bytes")]


class WunTensorInfo(C.Structure):
    _fields_ = [("name", C.c_char * 64), ("offset", C.c_int64), ("ndim", C.c_int32),
                ("shape", C.c_int64 * 4)]


_P = C.c_void_p
_SIGS = {
    "wun_get_padding": (C.c_int, [C.POINTER(WunConfig), C.c_int64, C.POINTER(C.c_int64),
                                  C.POINTER(C.c_int64)]),
    "wun_plan_create": (C.c_int, [C.POINTER(WunConfig), C.c_int64, C.c_int64, C.POINTER(_P)]),
    "wun_plan_destroy": (None, [_P]),
    "wun_plan_query": (C.c_int, [_P, C.POINTER(WunPlanInfo)]),
    "wun_plan_tensor": (C.c_int, [_P, C.c_int64, C.POINTER(WunTensorInfo)]),
    "wun_plan_activation": (C.c_int, [_P, C.c_int32, C.c_int32, C.POINTER(WunActivationInfo)]),
    "wun_forward": (C.c_int, [_P, _P, _P, _P, _P, C.c_int, _P]),
    "wun_loss_backward": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "wun_loss_backward_ex": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, C.POINTER(C.c_int64), C.POINTER(C.c_void_p), C.c_int32]),
    "wun_plan_tune": (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "wun_plan_tune_export": (C.c_int, [_P, C.c_char_p, C.c_int64]),
    "wun_plan_tune_import": (C.c_int, [_P, C.c_char_p]),
    "wun_adam_step": (C.c_int, [_P, _P, _P, _P, _P, C.c_int64, C.c_float, C.c_float, C.c_float,
                                C.c_float, C.c_float, _P]),
    "wun_op_conv1d": (C.c_int, [_P, _P, _P, _P] + [C.c_int] * 9 + [_P]),
    "wun_op_conv1d_wgrad_scratch": (C.c_int64, [C.c_int] * 5),
    "wun_op_conv1d_wgrad": (C.c_int, [_P, _P, _P, _P, _P] + [C.c_int] * 8 + [_P]),
    "wun_op_conv1d_dgrad": (C.c_int, [_P, _P, _P, _P] + [C.c_int] * 8 + [_P]),
    "wun_op_mfma_probe": (C.c_int, [_P, _P, _P, _P]),
    "wun_op_mfma_bf16_probe": (C.c_int, [_P, _P, _P, _P]),
    "wun_op_conv1d_bf16_scratch": (C.c_int64, [C.c_int] * 3),
    "wun_op_conv1d_bf16": (C.c_int, [_P, _P, _P, _P, _P] + [C.c_int] * 9 + [_P]),
    "wun_op_conv1d_dgrad_bf16_scratch": (C.c_int64, [C.c_int] * 3),
    "wun_op_conv1d_dgrad_bf16": (C.c_int, [_P, _P, _P, _P] + [C.c_int] * 8 + [_P]),
    "wun_op_force_conv_variant": (C.c_int, [C.c_int, C.c_int]),
    "wun_op_num_conv_variants": (C.c_int, []),
    "wun_op_force_wgrad_variant": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "wun_op_set_wgrad_bf16": (C.c_int, [C.c_int]),
    "wun_op_set_wgrad_win": (C.c_int, [C.c_int]),
    "wun_op_set_wgrad_narrow": (C.c_int, [C.c_int]),
    "wun_op_conv1d_ex": (C.c_int, [_P, C.c_int, _P, C.c_int, _P, _P, _P, _P] + [C.c_int] * 12 + [_P]),
    "wun_op_set_conv_copies": (C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.c_int, _P, C.c_int, C.c_int, C.c_int]),
    "wun_profile_begin": (C.c_int, []),
    "wun_profile_end": (C.c_int, [C.c_char_p, C.c_int64]),
    "wun_abi_sizes": (C.c_int, [C.POINTER(C.c_int64), C.c_int]),
    "wun_last_error": (C.c_char_p, []),
    "wun_version": (C.c_char_p, []),
}

EXPORTED_SYMBOLS = tuple(sorted(_SIGS))
_lib = None

# Process-wide record of the scheduling hint wun_config.exclusive_streams (include/wun.h): plans created with it run
# their side streams on LOWEST-priority hardware queues, and a process that has ever created those queues is ~40 %
# slower once a communication stream (RCCL) shares the device.  parallel.init_distributed() refuses to start a
# process group after such a plan exists (the hazard cannot be undone inside the process).
LOW_PRIORITY_PLANS = {"created": 0}


def load():
    """Load libwun.so.  torch must already be imported on a GPU box so that the library
    binds to the same HIP runtime instance torch uses (shared SONAME libamdhip64.so.7)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "libwun.so not found at %s -- build it with `python __graft_entry__.py` or "
            "`make -C wave-u-net_amd/csrc`; there is no CPU fallback" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)      # AttributeError if the .so does not export it
        fn.restype = res
        fn.argtypes = args
    # the structs of this binding against the sizes the library was compiled with (wun_abi_sizes, include/wun.h)
    sizes = (C.c_int64 * 3)()
    lib.wun_abi_sizes(sizes, 3)
    mine = (C.sizeof(WunConfig), C.sizeof(WunPlanInfo), C.sizeof(WunTensorInfo))
    if tuple(sizes) != mine:
        raise RuntimeError("libwun.so struct sizes %s != this binding's %s (stale library or binding)" % (tuple(sizes), mine))
    _lib = lib
    return lib


class WunError(RuntimeError):
    pass


def check(rc):
    """Map wun_status to the exception the reference would raise in the same situation."""
    if rc == 0:
        return
    msg = load().wun_last_error().decode("utf-8", "replace")
    if rc == -2:
        raise NotImplementedError(msg)            # Training.py:33, UnetAudioSeparator.py:136,144
    if rc == -1:
        raise ValueError(msg)                     # the reference's shape asserts
    raise WunError("wun status %d: %s" % (rc, msg))
