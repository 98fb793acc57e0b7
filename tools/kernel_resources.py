#!/usr/bin/env python3
"""Per-kernel register / LDS / occupancy table of a .hip file (hipcc -Rpass-analysis=kernel-resource-usage).
usage: kernel_resources.py path/to/file.hip [name-filter]"""
import re, subprocess, sys
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
out = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "--cuda-device-only",
                      "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[3:],
                     capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        cur = re.sub(r"^void ", "", cur).split("(")[0]
        rows[cur] = {}
        continue
    m = re.search(r"remark: .*?\]?\s*(SGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).split()[0]] = int(m.group(2))
for k, v in rows.items():
    if flt in k:
        print("%-48s VGPR %3d AGPR %3d scratch %3d occ %d" % (k, v.get("VGPRs", -1), v.get("AGPRs", -1), v.get("ScratchSize", -1), v.get("Occupancy", -1)))
