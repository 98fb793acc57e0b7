#!/bin/bash
# Register / scratch usage of every kernel of one translation unit (device-only compile for gfx950; no GPU needed).
# usage: [SRC=wun_bf16.hip] tools/kernel_resources.sh [extra hipcc flags]   -> table on stdout   (default SRC: wun_kernels.hip;
# the bf16 mode's units and wun_elementwise.hip are built with: -Xclang -target-feature -Xclang -packed-fp32-ops, csrc/Makefile)
set -e
R=$(cd $(dirname $0)/.. && pwd)
T=$(mktemp -d)
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -Wno-unused-function "$@" \
    -c $R/wave-u-net_amd/csrc/${SRC:-wun_kernels.hip} -o $T/k.co
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$T/k.co \
    --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/k.elf
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $T/k.elf > $T/notes.txt
python3 - $T/notes.txt <<'PY'
import re, subprocess, sys
txt = open(sys.argv[1]).read()
kern = re.split(r"\n\s+- \.agpr_count:", txt)[1:]
rows = []
for k in kern:
    k = ".agpr_count:" + k
    d = dict(re.findall(r"\.(\w+):\s+(\S+)", k))
    nm = subprocess.run(["c++filt", d["name"]], capture_output=True, text=True).stdout.strip()
    nm = nm.split("(")[0].replace("void wun::", "").replace("wun::", "")
    rows.append((nm, d))
rows.sort()
print("%-56s %5s %5s %6s %8s %5s %7s" % ("kernel", "vgpr", "agpr", "spill", "scratch", "sgpr", "lds"))
for nm, d in rows:
    print("%-56s %5s %5s %6s %8s %5s %7s" % (nm, d.get("vgpr_count"), d.get("agpr_count"), d.get("vgpr_spill_count"),
                                            d.get("private_segment_fixed_size"), d.get("sgpr_count"), d.get("group_segment_fixed_size")))
bad = [nm for nm, d in rows if int(d.get("private_segment_fixed_size", "0")) > 0]
print("\nkernels with scratch:", bad if bad else "none")
PY
rm -rf $T
