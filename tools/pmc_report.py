#!/usr/bin/env python3
"""Turn the per-kernel PMC aggregates written by tools/pmc_summarize.py on the GPU box into the
committed profile artefacts.
usage: pmc_report.py <pmc_FETCH_SIZE.json> <pmc_WRITE_SIZE.json> <pmc_mfma.json> <out_traffic.json> <out_mfma.txt> [note] [tune_table.txt]

With a tuning table given, its sha256 is stored in the traffic file: bench.py reports `roofline.traffic`
only when it runs exactly that table (same launches as the PMC passes)."""
import json
import sys


def main():
    fetch, write, mfma = (json.load(open(a)) for a in sys.argv[1:4])
    note = sys.argv[6] if len(sys.argv) > 6 else ""
    table_sha = None
    if len(sys.argv) > 7:
        import hashlib
        table_sha = hashlib.sha256(open(sys.argv[7]).read().encode()).hexdigest()
    kernels = {}
    for name, f in fetch.items():
        w = write.get(name)
        if w is None or "FETCH_SIZE" not in f or "WRITE_SIZE" not in w:
            continue
        n = f["launches"]
        fb = 2.0 * f["FETCH_SIZE"] * 1024.0 / n         # gfx950: FETCH_SIZE reports half of a wide coalesced read
        wb = w["WRITE_SIZE"] * 1024.0 / max(w["launches"], 1)
        us = f["duration_ns"] / n / 1e3
        kernels[name] = {"launches_profiled": n, "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb,
                         "hbm_bytes_per_launch": fb + wb, "avg_launch_us": us, "hbm_GBps": (fb + wb) / (us * 1e-6) / 1e9}
    json.dump({"note": "rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE, separate passes, single-stream "
                       "(profiling) mode, tuned plan loaded from the tuning table, last step only "
                       "(tools/pmc_summarize.py); FETCH_SIZE doubled per MI355X_MICROARCH.md; counters are KiB; "
                       "WRITE_SIZE uncalibrated. " + note, "tune_table_sha256": table_sha,
               "step_hbm_bytes": sum(k["hbm_bytes_per_launch"] * k["launches_profiled"] for k in kernels.values()),
               "kernels": kernels}, open(sys.argv[4], "w"), indent=1)
    rows = []
    tot_busy = tot_act = 0.0
    for name, c in mfma.items():
        act = c.get("GRBM_GUI_ACTIVE", 0.0)
        if act <= 0:
            continue
        wc = max(c.get("SQ_WAVE_CYCLES", 0.0), 1.0)
        util = 8.0 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (act * 1024.0)
        tot_busy += 8.0 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0); tot_act += act * 1024.0
        rows.append((c["duration_ns"], name, c["launches"], util, c.get("SQ_WAIT_ANY", 0) / wc,
                     c.get("SQ_WAIT_INST_ANY", 0) / wc, c.get("SQ_ACTIVE_INST_ANY", 0) / wc,
                     c.get("SQ_LDS_BANK_CONFLICT", 0) / wc))
    rows.sort(reverse=True)
    with open(sys.argv[5], "w") as f:
        f.write("rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY "
                "SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE\n")
        f.write("single-stream (profiling) mode, tuned plan, last step only; GRBM_GUI_ACTIVE is summed over the 8 XCDs;\n"
                "mfma_util = 8*MFMA_BUSY/(GUI_ACTIVE*1024 SIMDs); the other columns are fractions of SQ_WAVE_CYCLES. %s\n\n" % note)
        f.write("%-40s %5s %9s %9s %9s %9s %9s %9s\n" % ("kernel", "disp", "ms", "mfma_util", "wait_any", "wait_inst", "active", "lds_conf"))
        for d, name, n, u, wa, wi, ac, lc in rows:
            f.write("%-40s %5d %9.3f %9.3f %9.3f %9.3f %9.3f %9.4f\n" % (name, n, d / 1e6, u, wa, wi, ac, lc))
        if tot_act > 0:
            f.write("\nall kernels, time-weighted: mfma_util %.3f\n" % (tot_busy / tot_act))


if __name__ == "__main__":
    main()
