"""bench.py's own launcher: `python bench.py --gpus N` must start itself under
torch.distributed.run, and the exclusive_streams hazard must be refused.  CPU-only: --dry-run stops after
process-group initialisation (gloo) and the tuning-table broadcast."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _clean_env(**extra):
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "WUN_TUNE_CACHE")}
    env.update(extra)
    return env


def _json_lines(text):
    out = []
    for ln in text.splitlines():
        ln = ln.strip()
        if ln.startswith("{") and ln.endswith("}"):
            out.append(json.loads(ln))
    return out


@pytest.mark.parametrize("n", [1, 2])
def test_dry_run_launches_itself(n):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--dry-run"],
                       env=_clean_env(CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES=""), capture_output=True, text=True,
                       timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout                       # rank 0 prints ONE line
    d = lines[0]
    assert d["dry_run"] is True and d["n_gpus"] == n and d["scaling"] == "weak"
    assert d["config"]["parallelism"] == "dp%d" % n
    if n > 1:
        assert d["backend"] == "gloo"                      # no GPU here: the ranks rendezvous over gloo
    # the committed table of the headline config reached every rank (identical digest asserted inside bench.py)
    assert len(d["tune_table_sha16"]) == 16


def test_world_size_mismatch_is_an_error():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--dry-run"],
                       env=_clean_env(WORLD_SIZE="2", RANK="0", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1",
                                      MASTER_PORT="1", CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES=""),
                       capture_output=True, text=True, timeout=120, cwd=ROOT)
    assert r.returncode != 0                               # gpus=1 under a 2-rank environment: refused, not a hang


def test_process_group_after_low_priority_plan_is_refused(monkeypatch):
    """include/wun.h wun_config.exclusive_streams: lowest-priority side streams created before a process group
    cost the data-parallel run ~40 %; init_distributed refuses (no rendezvous is attempted)."""
    sys.path.insert(0, ROOT)
    from wave_u_net_amd import _lib, parallel
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "0")
    monkeypatch.setitem(_lib.LOW_PRIORITY_PLANS, "created", 1)
    with pytest.raises(RuntimeError, match="exclusive_streams"):
        parallel.init_distributed(backend="gloo")


def test_pinned_tables_are_never_the_writable_cache():
    """bench.py hands the committed table over as text; WUN_TUNE_CACHE is never pointed at it."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    assert 'os.environ["WUN_TUNE_CACHE"]' not in src
    assert "pinned_table=pinned_text" in src
