#!/bin/bash
# kasm.sh <mangled-name-fragment> : disassemble csrc/synth_kernels.hip for gfx950 and cut one kernel out into /tmp/kasm.s;
# prints its register / spill counts and where scratch traffic sits relative to the main loop
cd "$(dirname "$0")/../galileo-sdr-sim_amd" || exit 1
frag=${1:-k_synthILi12ELb0ELi0ELi1ELi0}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-value $KASM_FLAGS -S --cuda-device-only -o /tmp/sk.s csrc/synth_kernels.hip || exit 1
a=$(grep -n "^_Z7${frag}" /tmp/sk.s | head -1 | cut -d: -f1)
b=$(grep -n "amdhsa_kernel _Z7${frag}" /tmp/sk.s | cut -d: -f1)
sed -n "${a},${b}p" /tmp/sk.s > /tmp/kasm.s
wc -l /tmp/kasm.s
grep -E "^; (NumVgprs|ScratchSize|Occupancy|NumSgprs|LDSByteSize)" /tmp/kasm.s
echo "scratch ops at lines: $(grep -n 'scratch_' /tmp/kasm.s | cut -d: -f1 | tr '\n' ' ')"
echo "loops: $(grep -n 'Loop: Header=.*Depth=1' /tmp/kasm.s | sed -n '1p;$p' | cut -d: -f1 | tr '\n' ' ')"
echo "v_mul_lo_u32: $(grep -c v_mul_lo_u32 /tmp/kasm.s)"
