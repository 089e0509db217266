#!/usr/bin/env python3
"""Per-handle cycle of the pipelined bench out of a rocprofv3 --kernel-trace CSV: for every k_synth launch the time since the
previous k_synth END on the same queue splits into host gap (k_synth end -> first k_walk_carr of that handle's next batch),
walker chain (that k_walk_carr's start -> last walker kernel end) and wait (walker end -> k_synth start); plus the kernel's own
duration.  Medians over the steady-state launches.   tools/step_timeline.py <kernel_trace.csv> [label]"""
import csv
import statistics
import sys


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    label = sys.argv[2] if len(sys.argv) > 2 else ""
    qcol = [c for c in rows[0] if c.lower().replace("_", "") == "queueid"][0]
    ev = []
    for r in rows:
        n = r["Kernel_Name"]
        ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r[qcol], n.split("(")[0].replace("void ", "")))
    ev.sort()
    synth = [e for e in ev if e[3].startswith("k_synth")]
    walkers = [e for e in ev if e[3].startswith(("k_walk", "k_scanm", "k_pages"))]
    qs = sorted(set(e[2] for e in synth))
    out = {"gap": [], "walk": [], "wait": [], "synth": [], "cycle": []}
    for q in qs:
        s = [e for e in synth if e[2] == q]
        for a, b in zip(s[:-1], s[1:]):
            w = [e for e in walkers if a[1] <= e[0] <= b[0]]
            # the walkers of THIS handle's next batch: those from the first k_walk_carr after a's end on; with two
            # handles the other handle's walkers run while a runs, not after it
            g = [e for e in w if e[3].startswith("k_walk_carr")]
            if not g:
                continue
            g0 = g[0][0]
            mine = [e for e in w if e[0] >= g0]
            wend = max(e[1] for e in mine)
            wend = min(wend, b[0]) if wend > b[0] else wend
            out["gap"].append((g0 - a[1]) / 1e3)
            out["walk"].append((wend - g0) / 1e3)
            out["wait"].append((b[0] - wend) / 1e3)
            out["synth"].append((b[1] - b[0]) / 1e3)
            out["cycle"].append((b[1] - a[1]) / 1e3)
    def med(v):
        v = v[len(v) // 4:]  # steady state
        return statistics.median(v) if v else float("nan")
    print("%-28s n=%3d  per-handle cycle %7.1f us = host gap %6.1f + walker chain %6.1f + wait %6.1f + k_synth %7.1f   (queues %s)" % (
        label, len(out["cycle"]), med(out["cycle"]), med(out["gap"]), med(out["walk"]), med(out["wait"]), med(out["synth"]), ",".join(qs)))


if __name__ == "__main__":
    main()
