"""bench.py prints ONE JSON line with the fields the driver reads (small run)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_json_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--epochs", "24", "--steps", "3", "--warmup", "1"],
                       capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["preroll_steps"] > 0  # the device wake-up in front of the warm-up steps is reported, never timed
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"] and d["config"]["chain_mismatch"] == 0
    # the oracle's verdict on the timed output is part of the record (every epoch of this small batch)
    assert d["config"]["output_equals_oracle"] is True
    v = d["config"]["output_vs_oracle"]
    assert v["epochs_compared"] == 24 and v["int16_different"] == 0 and v["oracle_checksum"] == d["config"]["output_checksum"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-4
    cb = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in cb, k
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0
    assert d["value"] > 1000.0  # far above the 260 Msamples/s target even on a 24-epoch batch
    # the same step on fresh parameters: a new scenario planned every step, the last two outputs equal to the oracle's
    fp = d["configs"]["fresh_plan"]
    assert fp["output_equals_oracle"] is True and len(fp["oracle_checks"]) == 2 and fp["distinct_parameter_sets"] >= 3
    assert fp["oracle_checks"][0]["seed"] != fp["oracle_checks"][1]["seed"]
    assert fp["plan_ms"] > 0 and fp["h2d_ms"] > 0 and fp["ms_per_step"] > 0 and d["config"]["plan_ms"] == fp["plan_ms"]


def test_report_exchange_falls_back_to_the_control_group():
    """bench.exchange_report: over RCCL when the backend is nccl; if that raises (communicator did not come up), the same
    exchange through the gloo group that carried the barriers, and the line names the backend it used."""
    sys.path.insert(0, ROOT)
    import bench

    calls = []

    def report_ok(group, device):
        calls.append((group, device))
        return 1.5, 100, 7, [{"rank": 0}]

    def report_rccl_down(group, device):
        calls.append((group, device))
        if device == "cuda":
            raise RuntimeError("NCCL error")
        return 1.5, 100, 7, [{"rank": 0}]

    class FakeDist:  # the agreement on the fallback: MIN of the ranks' ok flags over the control group (here: one rank)
        class ReduceOp:
            MIN = "min"

        def __init__(self, others_ok=1):
            self.others_ok, self.calls = others_ok, []

        def all_reduce(self, t, op=None, group=None):
            self.calls.append((op, group))
            t[0] = min(int(t[0]), self.others_ok)

    dist = FakeDist()
    assert bench.exchange_report(None, "nccl", {"group": None}, report_ok) == (1.5, 100, 7, [{"rank": 0}], None)
    assert bench.exchange_report(dist, "gloo", {"group": None}, report_ok)[4] == "gloo" and calls[-1] == (None, "cpu")
    assert bench.exchange_report(dist, "nccl", {"group": "ctl"}, report_ok)[4] == "rccl" and calls[-1] == (None, "cuda")
    assert dist.calls[-1] == ("min", "ctl")
    out = bench.exchange_report(dist, "nccl", {"group": "ctl"}, report_rccl_down)
    assert out[:4] == (1.5, 100, 7, [{"rank": 0}]) and out[4].startswith("gloo (RCCL failed: RuntimeError")
    assert calls[-2:] == [(None, "cuda"), ("ctl", "cpu")]
    # RCCL worked HERE but failed on another rank: this rank falls back with the others instead of keeping its result
    n = len(calls)
    out = bench.exchange_report(FakeDist(others_ok=0), "nccl", {"group": "ctl"}, report_ok)
    assert out[4] == "gloo (RCCL failed: on another rank)" and calls[n:] == [(None, "cuda"), ("ctl", "cpu")]
    with pytest.raises(RuntimeError):  # no control group to fall back to: the error is the caller's
        bench.exchange_report(dist, "nccl", {"group": None}, report_rccl_down)


def test_committed_pmc_summaries_belong_to_this_build_of_the_kernel():
    """bench.py quotes HBM traffic and instruction counts of k_synth_g from the newest committed PMC summaries (it cannot run
    rocprofv3 on itself): they must have been measured on the kernel sources in this tree (tools/pmc_synth.sh records their hash) --
    VERDICT r5 item 8: a stale file would otherwise go unnoticed when the kernel changes."""
    import glob

    sys.path.insert(0, ROOT)
    import bench

    for pat in ("*_pmc_k_synth.json", "*_pmc_k_synth_all.json"):
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", pat)))
        assert files, pat
        assert bench.profile_is_current(files[-1]), "%s was measured on other kernel sources: run tools/profile_round.sh and commit its summaries" % os.path.basename(files[-1])
