#!/usr/bin/env python3
"""Split a rocprofv3 --kernel-trace of bench.py into its slow steps (the first ones after a fence) and its fast ones, and set the
two side by side: step time, kernel time per queue, and the kernels whose duration differs most.
usage: phase_summary.py <dir> [threshold_ms]"""
import collections
import csv
import glob
import statistics
import sys

d = sys.argv[1]
f = (glob.glob(f"{d}/*/*kernel_trace.csv") + glob.glob(f"{d}/*kernel_trace.csv"))[0]
rows = list(csv.DictReader(open(f)))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
marks = sorted(r["e"] for r in rows if "wgrad_stream" in r["Kernel_Name"])
steps = [(marks[i], marks[i + 1]) for i in range(len(marks) - 1)]
dur = [(b - a) / 1e6 for a, b in steps]
print("step ms:", " ".join(f"{x:.2f}" for x in dur))
thr = float(sys.argv[2]) if len(sys.argv) > 2 else (statistics.median(dur[-10:]) * 1.08)
kind = ["slow" if thr < x < 6 else "fast" if x <= thr else "skip" for x in dur]
print("threshold %.2f ms: %d slow, %d fast, %d skipped" % (thr, kind.count("slow"), kind.count("fast"), kind.count("skip")))
agg = {"slow": collections.defaultdict(list), "fast": collections.defaultdict(list)}
queue = {"slow": collections.Counter(), "fast": collections.Counter()}
rows.sort(key=lambda r: r["s"])
import bisect
starts = [a for a, _ in steps]
for r in rows:
    i = bisect.bisect_right(starts, r["s"]) - 1
    if i < 0 or i >= len(steps) or r["s"] >= steps[i][1] or kind[i] == "skip":
        continue
    name = r["Kernel_Name"].split("(")[0][:44]
    g = (int(r["Grid_Size_X"]) // int(r["Workgroup_Size_X"]), int(r["Grid_Size_Y"]), int(r["Grid_Size_Z"]))
    agg[kind[i]][(name, g)].append(r["e"] - r["s"])
    queue[kind[i]][r["Queue_Id"]] += r["e"] - r["s"]
for k in ("slow", "fast"):
    n = kind.count(k)
    if n:
        print(k, "steps: kernel ms per step by queue:", {q: round(v / n / 1e6, 3) for q, v in sorted(queue[k].items())},
              "sum %.3f" % (sum(queue[k].values()) / n / 1e6), "step %.3f" % statistics.mean(x for x, kk in zip(dur, kind) if kk == k))
ns, nf = kind.count("slow"), kind.count("fast")
if ns and nf:
    diff = []
    for key in set(agg["slow"]) | set(agg["fast"]):
        a, b = agg["slow"].get(key, []), agg["fast"].get(key, [])
        diff.append((sum(a) / ns - sum(b) / nf, key, len(a) / ns, len(b) / nf, sum(a) / max(len(a), 1), sum(b) / max(len(b), 1)))
    diff.sort(reverse=True)
    print("kernels by (slow - fast) time per step:")
    for dlt, (name, g), ca, cb, ma, mb in diff[:25] + diff[-6:]:
        print(f"{name:46s} grid={str(g):16s} calls/step {ca:5.1f} {cb:5.1f}  avg us {ma / 1e3:8.1f} {mb / 1e3:8.1f}  delta/step {dlt / 1e3:8.1f} us")
