#!/bin/bash
# the scalar-base LDS-DMA helper of wun_wgrad_win.hip writes M0 inside an asm statement (M0 cannot be named
# as a clobber: reserved register).  This check disassembles the gfx950 code object and verifies that EVERY
# global_load_lds instruction is fed by an M0 write that follows the previous global_load_lds / basic-block label, i.e.
# that the compiler never relies on an M0 value surviving one of the asm statements.  No GPU needed.
# usage: tools/m0_check.sh [file.hip ...]   (default: every .hip of the library that uses LDS-DMA)
set -e
R=$(cd $(dirname $0)/.. && pwd)
T=$(mktemp -d)
FILES=${@:-wun_wgrad_win.hip wun_kernels.hip wun_bf16.hip}
rc=0
for f in $FILES; do
    /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 --cuda-device-only -Wno-unused-function -w \
        -c $R/wave-u-net_amd/csrc/$f -o $T/k.co
    /opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$T/k.co \
        --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/k.elf
    /opt/rocm/lib/llvm/bin/llvm-objdump -d $T/k.elf > $T/k.s
    python3 - $T/k.s $f <<'PY' || rc=1
import re, sys
fresh = False          # has M0 been written since the last consumer / label?
n = bad = 0
for ln in open(sys.argv[1]):
    if re.match(r"^[0-9a-f]+ <", ln):      # function / basic-block label: assume nothing
        fresh = False
        continue
    m = re.search(r"\t(s_mov_b32 m0|s_add_[iu]32 m0|s_lshl_b32 m0|s_or_b32 m0|s_readfirstlane_b32 m0|v_readfirstlane_b32 m0)", ln)
    if m:
        fresh = True
        continue
    if "global_load_lds" in ln or "buffer_load" in ln and " lds" in ln:
        n += 1
        if not fresh:
            bad += 1
        fresh = False
print("%-22s %5d LDS-DMA instructions, %d without a fresh M0 write" % (sys.argv[2], n, bad))
sys.exit(1 if bad else 0)
PY
done
rm -rf $T
exit $rc
