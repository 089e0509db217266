#!/bin/bash
# kasm_g.sh [mangled-name-fragment] : disassemble csrc/synth_group.hip for gfx950 and cut one kernel out into /tmp/kasm_g.s;
# prints its register / spill counts and instruction mix
cd "$(dirname "$0")/../galileo-sdr-sim_amd" || exit 1
frag=${1:-k_synth_gILi12ELb0ELi1E}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-value $KASM_FLAGS -S --cuda-device-only -o /tmp/sg.s csrc/synth_group.hip || exit 1
a=$(grep -n "^_Z9${frag}" /tmp/sg.s | head -1 | cut -d: -f1)
b=$(grep -n "amdhsa_kernel _Z9${frag}" /tmp/sg.s | cut -d: -f1)
sed -n "${a},${b}p" /tmp/sg.s > /tmp/kasm_g.s
wc -l /tmp/kasm_g.s
grep -E "^; (NumVgprs|NumAgprs|ScratchSize|Occupancy|NumSgprs|LDSByteSize|SGPRSpill|VGPRSpill)" /tmp/kasm_g.s
echo "scratch ops: $(grep -c 'scratch_' /tmp/kasm_g.s)  v_writelane: $(grep -c v_writelane /tmp/kasm_g.s) v_readlane: $(grep -c v_readlane /tmp/kasm_g.s)"
for i in v_pk_mad_u16 ds_read_b32 ds_read2_b32 ds_read_b64 ds_read_b128 v_add_f64 v_fma_f64 v_lshl_add_u32 v_bfe_i32 v_min3_u32 v_bfi_b32 s_load s_waitcnt global_load v_readfirstlane v_mov_b32 v_cndmask; do echo -n "$i: $(grep -c "$i" /tmp/kasm_g.s)  "; done; echo
