#!/bin/bash
# ADVICE round 5 / DESIGN.md section 5.3: the translation units whose kernels can run beside a bf16 MFMA kernel must contain NO
# packed fp32 VALU instruction (v_pk_fma_f32, v_pk_mul_f32, v_pk_add_f32, v_pk_mov_b32): tools/probes/pk_fma_probe.hip shows the
# compiler's packed instruction mix returning different results from run to run while v_mfma_f32_16x16x32_bf16 waves share the
# CU (profiles/round6_pk_fma_probe.txt).  This check disassembles the OBJECTS the library is linked from (so it sees the flags
# the Makefile really used) and fails if one of those units carries such an instruction.  No GPU needed.
# usage: tools/pk_check.sh      (after `make -C wave-u-net_amd/csrc`)
R=$(cd $(dirname $0)/.. && pwd)
T=$(mktemp -d)
rc=0
for u in wun_narrow wun_bf16 wun_wgrad_bf16 wun_elementwise; do
    o=$R/wave-u-net_amd/csrc/$u.o
    [ -f $o ] || { echo "$u.o: not built"; rc=1; continue; }
    # the device code object is a fat binary embedded in the host object's .hip_fatbin section
    /opt/rocm/lib/llvm/bin/llvm-objcopy -O binary --only-section=.hip_fatbin $o $T/$u.fat 2>/dev/null
    /opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$T/$u.fat \
        --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/$u.elf 2>/dev/null || { echo "$u.o: no gfx950 code object"; rc=1; continue; }
    n=$(/opt/rocm/lib/llvm/bin/llvm-objdump -d $T/$u.elf | grep -c -E "v_pk_(fma|mul|add)_f32|v_pk_mov_b32")
    k=$(/opt/rocm/lib/llvm/bin/llvm-objdump -d $T/$u.elf | grep -c -E "^[0-9a-f]+ <.*>:")
    printf "%-18s %4d functions, %d packed fp32 VALU instructions\n" $u.o $k $n
    [ "$n" = 0 ] || rc=1
done
# control: the exact-fp32 units keep them (they never run beside a bf16 MFMA kernel)
o=$R/wave-u-net_amd/csrc/wun_kernels.o
if [ -f $o ]; then
    /opt/rocm/lib/llvm/bin/llvm-objcopy -O binary --only-section=.hip_fatbin $o $T/k.fat 2>/dev/null
    /opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$T/k.fat --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$T/k.elf 2>/dev/null
    n=$(/opt/rocm/lib/llvm/bin/llvm-objdump -d $T/k.elf | grep -c -E "v_pk_(fma|mul|add)_f32|v_pk_mov_b32")
    printf "%-18s (control: exact-fp32 unit, flag not applied) %d packed fp32 VALU instructions\n" wun_kernels.o $n
fi
rm -rf $T
exit $rc
