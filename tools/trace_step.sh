#!/bin/bash
# Kernel timeline of one single-handle step (rocprofv3 --kernel-trace): start offset, duration, gap to the previous kernel end.
export TMPDIR=/tmp
out=gpurun_out/trace_$1
mkdir -p $out
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $out -- python bench.py --pipeline ${2:-1} --steps 6 --warmup 2 --no-cpu-baseline --no-extras > $out/log.txt 2>&1
python3 - $out <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last complete step: from the last-but-one k_walk_carr (the head of the walker chain) to the end of the following k_synth
idx = [i for i, r in enumerate(rows) if "k_walk_carr" in r["Kernel_Name"]]
i0 = idx[-2]
t0 = int(rows[i0]["Start_Timestamp"])
prev_end = t0
for r in rows[i0:]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].split("(")[0][:40]
    print("%9.1f us  dur %8.1f  gap %7.1f  q%s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, r.get("Queue_Id", "?"), name))
    prev_end = max(prev_end, e)
    if "k_synth" in r["Kernel_Name"]:
        break
PY
