#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per (kernel, grid) time per step.
usage: trace_summary.py <dir> <steps | 0 = count bench.py training steps> [top]"""
import collections
import csv
import glob
import sys

d, steps = sys.argv[1], int(sys.argv[2])
f = (glob.glob(f"{d}/*/*kernel_trace.csv") + glob.glob(f"{d}/*kernel_trace.csv"))[0]
rows = list(csv.DictReader(open(f)))
marks = sorted(int(r["Start_Timestamp"]) for r in rows if "wgrad_stream" in r["Kernel_Name"])
if marks and steps == 0:
    # bench.py trace: the steady-state window scripts/stream_busy.py uses -- the last (up to) 8 training steps, delimited by the
    # stem weight-gradient kernel (the last kernel of every backward pass); warm-up steps (event timing on, map plans still being
    # recorded: ~55 more launches per step) and the forward-only phase after the timed region are left out
    steps = min(8, len(marks) - 1)
    t0, t1 = marks[-1 - steps], marks[-1]
    rows = [r for r in rows if t0 <= int(r["Start_Timestamp"]) < t1]
agg = collections.defaultdict(list)
for r in rows:
    name = r["Kernel_Name"].split("(")[0][:46]
    g = (int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
    agg[(name, g)].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
tot = sum(sum(v) for v in agg.values())
print(f"total GPU kernel time per step: {tot / steps / 1e6:.3f} ms over {len(rows) / steps:.0f} launches/step")
top = int(sys.argv[3]) if len(sys.argv) > 3 else 30
for (name, g), v in sorted(agg.items(), key=lambda kv: -sum(kv[1]))[:top]:
    print(f"{name:48s} grid={str(g):18s} calls={len(v):4d} avg={sum(v) / len(v) / 1e3:8.1f}us per-step={sum(v) / steps / 1e3:7.1f}us")
