"""The N > 1 launch path of bench.py on ONE GPU: two ranks under torch.distributed.run, both on device 0, gloo for the barrier and
the report (GAL_BENCH_DEVICE / GAL_BENCH_BACKEND) -- rank_workload / epoch_range / gal_synth_execute_range / reduce_report together
on a real device, from two processes.  A launch-path check: it says nothing about scaling (the ranks share the GPU)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(args, nproc=0, env_extra=None):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **(env_extra or {}))
    if nproc:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr",
               "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc)] + args
    else:
        cmd = [sys.executable, os.path.join(ROOT, "bench.py")] + args
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_two_ranks_on_one_gpu_strong_split_sums_to_the_single_process_checksum():
    common = ["--steps", "3", "--warmup", "1", "--epochs", "96", "--no-extras", "--no-cpu-baseline", "--preroll-ms", "0"]
    one = _bench(common)
    two = _bench(common + ["--shard", "scenario"], nproc=2, env_extra={"GAL_BENCH_DEVICE": "0", "GAL_BENCH_BACKEND": "gloo"})
    assert two["n_gpus"] == 2 and two["scaling"] == "strong" and "rehearsal" in two
    # the ranges are contiguous, cover the scenario, and the later rank's is the shorter one (it walks the longer prefix)
    ranks = sorted(two["ranks"], key=lambda r: r["rank"])
    assert ranks[0]["epochs"][0] == 0 and ranks[0]["epochs"][1] == ranks[1]["epochs"][0] and ranks[1]["epochs"][1] == 96
    assert ranks[0]["epochs"][1] - ranks[0]["epochs"][0] >= ranks[1]["epochs"][1] - ranks[1]["epochs"][0]
    assert "rank_imbalance" in two and two["config"]["chain_mismatch"] == 0
    # SUM over the ranks of the 32-bit checksums of their ranges == the checksum of the whole scenario from one process
    assert two["config"]["output_checksum"] == one["config"]["output_checksum"]


def test_two_ranks_on_one_gpu_weak_split():
    common = ["--steps", "3", "--warmup", "1", "--epochs", "48", "--no-extras", "--no-cpu-baseline", "--preroll-ms", "0"]
    two = _bench(common, nproc=2, env_extra={"GAL_BENCH_DEVICE": "0", "GAL_BENCH_BACKEND": "gloo"})
    assert two["n_gpus"] == 2 and two["scaling"] == "weak" and len(two["ranks"]) == 2
    assert [r["epochs"] for r in sorted(two["ranks"], key=lambda r: r["rank"])] == [[0, 48], [0, 48]]
