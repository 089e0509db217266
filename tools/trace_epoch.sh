#!/bin/bash
# Kernel timeline of ONE-EPOCH calls (INTEGRATION.md option B: gal_synth_run_host per 0.1 s epoch): rocprofv3 --kernel-trace over
# tools/per_epoch_latency.py, the last complete call printed: start offset, duration, gap to the previous kernel end.  tools/trace_epoch.sh <tag>
export TMPDIR=/tmp
out=gpurun_out/trace_epoch_$1
mkdir -p $out
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $out -- python tools/per_epoch_latency.py > $out/log.txt 2>&1
python3 - $out <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_publish" in r["Kernel_Name"]]  # (the last kernel of a call)
i0, i1 = idx[-3] + 1, idx[-2] + 1
t0 = int(rows[i0]["Start_Timestamp"])
prev_end = t0
for r in rows[i0:i1]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][:44]
    print("%9.1f us  dur %8.1f  gap %7.1f  q%s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r.get("Queue_Id", "?"), name))
    prev_end = max(prev_end, e)
print("call period: %.1f us" % ((int(rows[i1]["Start_Timestamp"]) - t0) / 1e3))
PY
