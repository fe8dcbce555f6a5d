#!/usr/bin/env python3
"""One training step of a rocprofv3 kernel trace of bench.py as a launch-by-launch listing per hardware queue:
start (us after the step began), duration, gap to the previous launch of the same queue, short kernel name + grid.
Steps are delimited by the stem weight-gradient kernel.  usage: step_chain.py <trace dir> [step index from the end] [queue]"""
import csv
import glob
import re
import sys

d = sys.argv[1]
which = int(sys.argv[2]) if len(sys.argv) > 2 else 2
only_q = sys.argv[3] if len(sys.argv) > 3 else None
f = (glob.glob(f"{d}/*/*kernel_trace.csv") + glob.glob(f"{d}/*kernel_trace.csv"))[0]
rows = list(csv.DictReader(open(f)))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
marks = [r["e"] for r in rows if "wgrad_stream" in r["Kernel_Name"]]
t0, t1 = marks[-2 - which], marks[-1 - which]
seg = [r for r in rows if t0 <= r["s"] < t1]
print(f"step wall {(t1 - t0) / 1e3:.1f} us, {len(seg)} launches")


def short(nm):
    nm = re.sub(r"^void ", "", nm)
    nm = nm.replace("mink::", "").replace("at::native::", "at:")
    m = re.match(r"([A-Za-z0-9_:]+)(<[^(]*>)?", nm)
    base = m.group(1) if m else nm
    targs = (m.group(2) or "") if m else ""
    return (base + targs)[:44]


queues = sorted({r["Queue_Id"] for r in seg})
for q in queues:
    if only_q and q != only_q:
        continue
    rs = [r for r in seg if r["Queue_Id"] == q]
    busy = sum(r["e"] - r["s"] for r in rs)
    print(f"--- queue {q}: {len(rs)} launches, busy {busy / 1e3:.1f} us")
    prev = None
    for r in rs:
        gap = (r["s"] - prev) / 1e3 if prev is not None else 0.0
        wg = [int(r[f"Grid_Size_{a}"]) // max(1, int(r[f"Workgroup_Size_{a}"])) for a in "XYZ"]
        print(f"{(r['s'] - t0) / 1e3:8.1f} {(r['e'] - r['s']) / 1e3:7.1f} gap {gap:6.1f}  {short(r['Kernel_Name']):44s} {wg}")
        prev = r["e"]
