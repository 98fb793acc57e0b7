"""Driven by tests/test_abi_asan.py inside a process that has the ASan runtime preloaded and
WUN_LIB=libwun_asan.so: exercises every host-side path of the C ABI that needs no GPU."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np                                   # noqa: E402
import wave_u_net_amd as wun                         # noqa: E402
from wave_u_net_amd import _lib                      # noqa: E402
from wave_u_net_amd.separator import UnetAudioSeparator, _wun_config   # noqa: E402
from oracle.golden_params import GOLDEN_CASES        # noqa: E402


def main():
    lib = _lib.load()
    assert os.path.basename(_lib.LIB_PATH) == "libwun_asan.so", _lib.LIB_PATH
    n = 0
    # shape solver over the reference-generated answers
    for c in json.load(open(os.path.join(ROOT, "tests", "golden", "get_padding.json"))):
        cfg = wun.get_config("baseline", num_layers=c["num_layers"], filter_size=c["filter_size"],
                             merge_filter_size=c["merge_filter_size"], input_filter_size=c["input_filter_size"],
                             output_filter_size=c["output_filter_size"], context=c["context"], mono_downmix=c["mono_downmix"])
        i, o = UnetAudioSeparator(cfg).get_padding(np.array([16, c["desired"], 0]))
        assert list(i) == c["input_shape"] and list(o) == c["output_shape"]
        n += 1
    # plan builder: every golden config, both compute modes, the full-size and the deep configs
    plans = []
    for dt in ("f32", "bf16"):
        for name, case in GOLDEN_CASES.items():
            sep = UnetAudioSeparator(wun.get_config("baseline", compute_dtype=dt, **case["cfg"]))
            i, _ = sep.get_padding(np.array([case["batch"], case["frames"], 0]))
            plans.append(sep._plan(case["batch"], int(i[1])))
            n += 1
    sep = UnetAudioSeparator(wun.get_config("m1_context"))
    big = sep._plan(16, 147443)
    deep = UnetAudioSeparator(wun.get_config("baseline", num_layers=16, num_initial_filters=48, mono_downmix=False,
                                             task="multi_instrument", output_type="difference"))._plan(16, 589824)
    for p in plans + [big, deep]:
        assert p.info.num_tensors == len(p.tensors) and p.info.workspace_floats > 0
    # rejected configurations / shapes
    for bad in (dict(filter_size=17), dict(num_layers=0), dict(output_filter_size=4000)):
        try:
            UnetAudioSeparator(wun.get_config("baseline", **bad))._plan(1, 16384)
            raise SystemExit("accepted %r" % bad)
        except (ValueError, NotImplementedError):
            n += 1
    try:
        sep._plan(1, 1000)
        raise SystemExit("accepted a too-short input")
    except ValueError:
        n += 1
    # tuning-table parser: garbage, truncation, foreign plans, out-of-range entries
    good_head = None
    for text in ["", "x", "wun-tune 1 B=16\n", "wun-tune 2 order=zz variants=1 B=1\ncf 0 0\n",
                 "wun-tune 2 " + "A" * 5000 + " cf=1 cb=1 wg=1\ncf 1 1\n", "\n\n\n", "cf 1 2\ncb 3 4\nwg 1 2 3 4\nend\n"]:
        rc = lib.wun_plan_tune_import(big.handle, text.encode())
        assert rc == -1, (text[:30], rc)
        n += 1
    buf = C.create_string_buffer(64)
    assert lib.wun_plan_tune_export(big.handle, buf, len(buf)) == -1            # untuned plan
    ti = _lib.WunTensorInfo()
    assert lib.wun_plan_tensor(big.handle, 10 ** 9, C.byref(ti)) == -1
    assert lib.wun_plan_tensor(big.handle, -1, C.byref(ti)) == -1
    assert lib.wun_op_num_conv_variants() > 30
    assert lib.wun_op_conv1d_wgrad_scratch(16, 24, 48, 15, 1000) > 0
    assert lib.wun_op_conv1d_bf16_scratch(24, 48, 15) > 0
    del plans, big, deep
    print("asan driver ok: %d checks" % n)


if __name__ == "__main__":
    main()
