#!/usr/bin/env python3
"""Regenerate the per-config result tables of DESIGN.md (section 6) and BASELINE.md (section 4) from
profiles/<ROUND>_bench.json and profiles/<ROUND>_cfg_*.json (the latest round's files)."""
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ROUND = "round6"


def last_json(path):
    return json.loads(open(path).read().strip().splitlines()[-1])


def main():
    b = last_json(os.path.join(ROOT, "profiles", ROUND + "_bench.json"))
    cfg = {f.split(ROUND + "_cfg_")[1][:-5]: last_json(f) for f in glob.glob(os.path.join(ROOT, "profiles", ROUND + "_cfg_*.json"))}
    cb = b["cpu_baseline"]
    tf = lambda d: d["config"]["achieved_tflops_executed"]

    # ---- DESIGN.md
    p = os.path.join(ROOT, "DESIGN.md")
    s = open(p).read()
    rows = []
    for label, name in (("M1 same-pad T=16384 (`baseline`, configs[0])", "baseline"), ("M1 + context (configs[1], headline)", "m1_context"),
                        ("M4 `baseline_stereo` (configs[2])", "baseline_stereo"), ("M5 `full`", "full"),
                        ("M6 `full_multi_instrument` (configs[3])", "full_multi_instrument"),
                        ("deep L16/F48, 589 824 samples (configs[4]; fp32 heuristic, bf16 autotuned tilings)", "deep")):
        f32 = b if name == "m1_context" else cfg[name + "_f32"]
        bf = cfg[name + "_bf16"]
        rows.append("| %s | %.2f | %.3g | %.2f | %.3g | %.2f× |" % (label, f32["ms_per_step"], f32["value"], bf["ms_per_step"], bf["value"],
                                                                  f32["ms_per_step"] / bf["ms_per_step"]))
    head = "| config | fp32 ms/step | fp32 samples/s | bf16 ms/step | bf16 samples/s | speed-up |\n|---|---|---|---|---|---|\n"
    i = s.index(head)
    j = s.index("\n\n", i)
    s = s[:i] + head + "\n".join(rows) + s[j:]
    open(p, "w").write(s)

    # ---- BASELINE.md
    p = os.path.join(ROOT, "BASELINE.md")
    s = open(p).read()
    i = s.index("## 4. Results")
    r = []
    r.append("| **M1 + context (configs[1], the bench line)**, 147 443 → 16 389 | 1 | fp32 | **%.3g** | **%.2f** | executed FLOPs (dead odd outputs skipped, SURVEY §8d) %.1f TFLOP/s = %.0f %% of 157.3 (the reference graph's skipped FLOPs are NOT counted as achieved); kernel families `conv_mfma_kernel` %.3f, `wgrad_win_kernel` %.3f, `wgrad_mfma_kernel` (deep levels) %.3f by HIP events net of the event pair's own %.2f µs (conv %.3f uncorrected; DESIGN §6) | %.3g (%d cores, %s; whole batch) / %.3g (1 thread) | %.0f× |" % (
        b["value"], b["ms_per_step"], tf(b), 100 * tf(b) / 157.3, b["roofline"]["family_frac"]["conv_mfma_kernel"],
        b["roofline"]["family_frac"].get("wgrad_win_kernel", float("nan")), b["roofline"]["family_frac"]["wgrad_mfma_kernel"], b["roofline"].get("event_bracket_overhead_us", 0.0),
        b["roofline"].get("achieved_raw_events", b["roofline"]["achieved"]) / 157.3, cb["value"], cb["cores"], cb["cpu_model"], cb["value_1_thread"], b["value"] / cb["value"]))
    r.append("| same | 1 | bf16 mode | %.3g | %.2f | — | — | — |" % (cfg["m1_context_bf16"]["value"], cfg["m1_context_bf16"]["ms_per_step"]))
    c0 = cfg["baseline_f32"]
    c0cpu = c0.get("cpu_baseline")
    r.append("| M1 as shipped (same padding, configs[0]: SURVEY §8d's PR1 CPU-baseline config), T=16 384 | 1 | fp32 | %.3g | %.2f | %.0f %% of 1.76e8 (%.1f TFLOP/s) | %s | %s |" % (
        c0["value"], c0["ms_per_step"], 100 * c0["value"] / 1.76e8, tf(c0),
        ("%.3g (%d cores) / %.3g (1 thread)" % (c0cpu["value"], c0cpu["cores"], c0cpu["value_1_thread"])) if c0cpu else "—",
        ("%.0f×" % (c0["value"] / c0cpu["value"])) if c0cpu else "—"))
    r.append("| same | 1 | bf16 mode | %.3g | %.2f | — | — | — |" % (cfg["baseline_bf16"]["value"], cfg["baseline_bf16"]["ms_per_step"]))
    r.append("| M4 `baseline_stereo` (configs[2]) | 1 | fp32 | %.3g | %.2f | %.1f TFLOP/s executed | — | — |" % (
        cfg["baseline_stereo_f32"]["value"], cfg["baseline_stereo_f32"]["ms_per_step"], tf(cfg["baseline_stereo_f32"])))
    r.append("| same | 1 | bf16 mode | %.3g | %.2f | %.1f %% of the bf16 roofline §3 quotes (6.0e8; bf16 activations in HBM since round 5, the narrow layers stay bound by the instruction stream around their MFMAs, DESIGN §5.3) | — | — |" % (
        cfg["baseline_stereo_bf16"]["value"], cfg["baseline_stereo_bf16"]["ms_per_step"], 100 * cfg["baseline_stereo_bf16"]["value"] / 6.0e8))
    r.append("| M5 `full` (learned upsampling) | 1 | fp32 / bf16 mode | %.3g / %.3g | %.2f / %.2f | — | — | — |" % (
        cfg["full_f32"]["value"], cfg["full_bf16"]["value"], cfg["full_f32"]["ms_per_step"], cfg["full_bf16"]["ms_per_step"]))
    r.append("| M6 `full_multi_instrument` (configs[3] shape) | 1 | fp32 / bf16 mode | %.3g / %.3g | %.2f / %.2f | — | — | — |" % (
        cfg["full_multi_instrument_f32"]["value"], cfg["full_multi_instrument_bf16"]["value"], cfg["full_multi_instrument_f32"]["ms_per_step"],
        cfg["full_multi_instrument_bf16"]["ms_per_step"]))
    r.append("| Deep L16/F48, same-pad 589 824 (configs[4] shape; heuristic tilings) | 1 | fp32 | %.3g | %.1f | %.0f %% of the fp32 compute roofline (%.1f TFLOP/s) | — | — |" % (
        cfg["deep_f32"]["value"], cfg["deep_f32"]["ms_per_step"], 100 * tf(cfg["deep_f32"]) / 157.3, tf(cfg["deep_f32"])))
    r.append("| same | 1 | bf16 mode | %.3g | %.1f | %.1f %% of 6.9e8 | — | — |" % (cfg["deep_bf16"]["value"], cfg["deep_bf16"]["ms_per_step"],
                                                                             100 * cfg["deep_bf16"]["value"] / 6.9e8))
    new = ("## 4. Results (round 6; 1×MI355X, B=16; `profiles/round6_bench.json`, `profiles/round6_cfg_*.json`)\n\n"
           "fp32 = the reference's arithmetic (exact-fp32 MFMA) — the only numbers comparable with the metric; bf16 mode = the mode of\n"
           "BASELINE.json configs[2], [4] (activations and their gradients in HBM as bf16, bf16 MFMA, fp32 accumulate / weights / optimizer; DESIGN.md §5.3), reported beside.\n\n"
           "| Config | GPUs | dtype | samples/s (out) | ms/step | fraction of the binding roofline | CPU baseline samples/s | speed-up |\n"
           "|---|---|---|---|---|---|---|---|\n" + "\n".join(r) + "\n\n"
           "Multi-GPU rows are measured by the driver (`SCALE_rNN.json`); none was measured so far (no multi-GPU node was available in\n"
           "rounds 1-6; `python bench.py --gpus N` launches itself and its line carries `comm.exposed_ms` / `ms_per_step_no_comm`).  Earlier rounds, same rows: round 1 2.77e7 / 9.47 ms (M1 + context),\n"
           "6.82e7 / 3.85 ms (M1), 327 ms (deep); round 2 2.91e7 / 9.00 ms, 7.14e7 / 3.67 ms, 318 ms; round 3 3.10e7 / 8.46 ms (8.53 on the driver's box), 7.59e7 / 3.45 ms, 298 ms;\n"
           "round 4 3.12e7 / 8.41 ms, 7.84e7 / 3.35 ms, 283 ms (bf16 mode with fp32 tensors in HBM: M4 4.17 ms, deep 96 ms); round 5 3.11e7 / 8.43 ms, 7.87e7 / 3.33 ms,\n"
           "281 ms (bf16 mode, activations in HBM as bf16: M4 3.31 ms, deep 71.4 ms).  Round 6: every observed conv output is computed once (DESIGN.md §4).\n"
           "(`tools/update_result_tables.py` regenerates this section and DESIGN.md's table from `profiles/`.)\n")
    open(p, "w").write(s[:i] + new)
    print("\n".join(rows))


if __name__ == "__main__":
    main()
