"""Observed-error log of the GPU parity tests: every comparison records the error it actually
measured (relative to the stated scale) next to the tolerance it asserted, so tolerances can be
kept within ~10x of what the kernels really deliver (gpurun_out/parity_observed.json on the GPU
box; summarised in DESIGN.md section 2)."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_PATH = os.path.join(ROOT, "gpurun_out", "parity_observed.json")
_LOG = {}


def record(test, what, observed, tolerance):
    ent = _LOG.setdefault(test, {})
    prev = ent.get(what)
    if prev is None or observed > prev["observed"]:
        ent[what] = {"observed": float(observed), "tolerance": float(tolerance)}
    try:
        os.makedirs(os.path.dirname(_PATH), exist_ok=True)
        old = {}
        if os.path.exists(_PATH):
            try:
                old = json.load(open(_PATH))
            except Exception:
                old = {}
        old.update(_LOG)
        with open(_PATH, "w") as f:
            json.dump(old, f, indent=1, sort_keys=True)
    except OSError:
        pass
    print("[parity] %-58s %-28s observed %.3e  (tolerance %.1e)" % (test, what, observed, tolerance))
