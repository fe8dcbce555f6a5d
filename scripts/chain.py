#!/usr/bin/env python3
"""The compute queue's chain of one steady-state step from a rocprofv3 kernel trace of bench.py: every launch with its
duration and the gap behind its predecessor on the same queue.  usage: chain.py <trace dir> [queue = the busiest]"""
import collections, csv, glob, sys

d = sys.argv[1]
f = (glob.glob(f"{d}/*/*kernel_trace.csv") + glob.glob(f"{d}/*kernel_trace.csv"))[0]
rows = list(csv.DictReader(open(f)))
marks = sorted(int(r["Start_Timestamp"]) for r in rows if "wgrad_stream" in r["Kernel_Name"])
t0, t1 = marks[-3], marks[-2]
sel = sorted((r for r in rows if t0 <= int(r["Start_Timestamp"]) < t1), key=lambda r: int(r["Start_Timestamp"]))
busy = collections.Counter()
for r in sel:
    busy[r["Queue_Id"]] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
q = sys.argv[2] if len(sys.argv) > 2 else busy.most_common(1)[0][0]
prev, tot_k, tot_g = None, 0, 0
print(f"step {(t1 - t0) / 1e3:.1f} us; queue {q}")
for r in sel:
    if r["Queue_Id"] != q:
        continue
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev) / 1e3 if prev is not None else 0.0
    prev = e
    tot_k += e - s
    tot_g += max(gap, 0) * 1e3
    g = (int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
    print(f"{(s - t0) / 1e3:8.1f} gap {gap:6.1f}  {(e - s) / 1e3:7.1f} us  {r['Kernel_Name'].split('(')[0][:52]:52s} {g}")
print(f"kernels {tot_k / 1e3:.1f} us, gaps {tot_g / 1e3:.1f} us")
