#!/bin/bash
# Randomised parity soak of the current tree (run on the GPU box): product build, the stitch with 8 / 3 legs per block (hooks build: its look-back under every small batch),
# CBOC mode, end-to-end scenarios.   tools/soak_round.sh <tag> [scale]   -> gpurun_out/<tag>_fuzz_soak.log
set -u
tag=${1:-rXX}
k=${2:-1}
out=gpurun_out/${tag}_fuzz_soak.log
mkdir -p gpurun_out
{
echo "### product build: fuzz_parity.py $((20000*k)) 301 / $((600*k)) 302 big"
timeout 1500 python tools/fuzz_parity.py $((20000*k)) 301 2>&1 | tail -1
timeout 1500 python tools/fuzz_parity.py $((600*k)) 302 big 2>&1 | tail -1
echo "### hooks build, 8 / 3 legs per block of the stitch (look-back under every batch): fuzz_parity.py $((3000*k)) 303 / $((150*k)) 304 big / $((2000*k)) 307"
GAL_FUZZ_HOOKS=1 GAL_SCAN_BLOCK_LEGS=8 timeout 900 python tools/fuzz_parity.py $((3000*k)) 303 2>&1 | tail -1
GAL_FUZZ_HOOKS=1 GAL_SCAN_BLOCK_LEGS=8 timeout 900 python tools/fuzz_parity.py $((150*k)) 304 big 2>&1 | tail -1
GAL_FUZZ_HOOKS=1 GAL_SCAN_BLOCK_LEGS=3 timeout 900 python tools/fuzz_parity.py $((2000*k)) 307 2>&1 | tail -1
echo "### CBOC: fuzz_parity.py $((4000*k)) 305 / $((150*k)) 306 big"
GAL_FUZZ_CBOC=1 timeout 900 python tools/fuzz_parity.py $((4000*k)) 305 2>&1 | tail -1
GAL_FUZZ_CBOC=1 timeout 900 python tools/fuzz_parity.py $((150*k)) 306 big 2>&1 | tail -1
echo "### round 6: plan_async, a plan staged while the batch before is in flight: fuzz_parity.py $((6000*k)) 308 / $((200*k)) 309 big"
GAL_FUZZ_ASYNC=1 timeout 1500 python tools/fuzz_parity.py $((6000*k)) 308 2>&1 | tail -1
GAL_FUZZ_ASYNC=1 timeout 1500 python tools/fuzz_parity.py $((200*k)) 309 big 2>&1 | tail -1
echo "### round 6: k_synth_g's wide instances for every batch of more than 12 channels (hooks build, GAL_G_WIDE): fuzz_parity.py $((5000*k)) 310 group"
GAL_FUZZ_HOOKS=1 GAL_G_WIDE=1 GAL_FUZZ_GROUP=1 timeout 1500 python tools/fuzz_parity.py $((5000*k)) 310 2>&1 | tail -1
echo "### round 6: k_synth_g's bisection instances at every rate (hooks build, GAL_G_SEARCH): fuzz_parity.py $((5000*k)) 311 group"
GAL_FUZZ_HOOKS=1 GAL_G_SEARCH=1 GAL_FUZZ_GROUP=1 timeout 1500 python tools/fuzz_parity.py $((5000*k)) 311 2>&1 | tail -1
echo "### end to end (streamed in 1-3 calls per scenario): fuzz_scenarios.py $((40*k)) cases seed 31"
timeout 900 python tools/fuzz_scenarios.py $((40*k)) 31 2>&1 | tail -1
} > $out 2>&1
cat $out
