"""Host side of libwun.so under AddressSanitizer + UBSan: the plan builder, the
shape solver, the tuning-table parser and the argument checks run in a child process with the
sanitizer runtime preloaded; any heap error / undefined behaviour aborts the child."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "wave-u-net_amd", "csrc")


@pytest.mark.skipif(shutil.which("make") is None or not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc + make")
def test_host_side_is_clean_under_asan_ubsan():
    subprocess.check_call(["make", "-C", CSRC, "-j4", "all", "asan"], stdout=subprocess.DEVNULL)
    rt = subprocess.check_output(["make", "-s", "-C", CSRC, "asan-rt"], text=True).strip()
    assert os.path.exists(rt), rt
    env = dict(os.environ, LD_PRELOAD=rt, WUN_LIB="libwun_asan.so",
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=1:halt_on_error=1",
               UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "asan_driver.py")], env=env, capture_output=True,
                       text=True, timeout=600)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-4000:])
    assert "asan driver ok" in r.stdout
