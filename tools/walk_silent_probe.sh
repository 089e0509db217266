#!/bin/bash
export TMPDIR=/tmp
out=gpurun_out/silent_probe
mkdir -p $out
timeout 200 rocprofv3 --kernel-trace --output-format csv -d $out -- python tools/walk_silent_probe.py > $out/log.txt 2>&1
python3 - $out <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*kernel_trace.csv")[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
per = collections.defaultdict(list)
for r in rows:
    per[r["Kernel_Name"].split("(")[0][:32]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in per.items():
    print("%-34s n=%d  %s" % (k, len(v), " ".join("%.1f" % x for x in v[-6:])))
PY
