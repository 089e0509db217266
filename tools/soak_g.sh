#!/bin/bash
# Randomised parity soak centred on k_synth_g / k_repair_g (run on the GPU box).  tools/soak_g.sh <tag> [scale] -> gpurun_out/<tag>_fuzz_soak_g.log
set -u
tag=${1:-rXX}; k=${2:-1}
out=gpurun_out/${tag}_fuzz_soak_g.log
mkdir -p gpurun_out
{
echo "### batches the group kernel can take: fuzz_parity.py $((12000*k)) 401 / $((400*k)) 402 big"
GAL_FUZZ_GROUP=1 timeout 1500 python tools/fuzz_parity.py $((12000*k)) 401 2>&1 | tail -2
GAL_FUZZ_GROUP=1 timeout 1500 python tools/fuzz_parity.py $((400*k)) 402 big 2>&1 | tail -2
echo "### the same with 8 legs per block of the stitch (hooks build: its look-back under every batch)"
GAL_FUZZ_GROUP=1 GAL_FUZZ_HOOKS=1 GAL_SCAN_BLOCK_LEGS=8 timeout 900 python tools/fuzz_parity.py $((3000*k)) 403 2>&1 | tail -2
echo "### general mix: fuzz_parity.py $((8000*k)) 404 / $((200*k)) 405 big"
timeout 1500 python tools/fuzz_parity.py $((8000*k)) 404 2>&1 | tail -2
timeout 1500 python tools/fuzz_parity.py $((200*k)) 405 big 2>&1 | tail -2
echo "### end to end (streamed in 1-3 calls per scenario): fuzz_scenarios.py $((40*k)) cases seed 41"
timeout 900 python tools/fuzz_scenarios.py $((40*k)) 41 2>&1 | tail -1
} > $out 2>&1
cat $out
