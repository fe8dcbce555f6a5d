#!/usr/bin/env python3
"""Per-queue busy time, union busy time and per-category kernel time per step from a rocprofv3
kernel trace of bench.py (training steps are delimited by the stem weight-gradient kernel, the last
kernel of every backward pass; the forward-only phase that follows the timed region is left out)."""
import collections, csv, glob, sys

d = sys.argv[1]
f = (glob.glob(f"{d}/*/*kernel_trace.csv") + glob.glob(f"{d}/*kernel_trace.csv"))[0]
rows = list(csv.DictReader(open(f)))
for r in rows:
    r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
marks = [r["s"] for r in rows if "wgrad_stream" in r["Kernel_Name"]]
n = min(8, len(marks) - 1)
t0, t1 = marks[-1 - n], marks[-1]
seg = [r for r in rows if t0 <= r["s"] < t1]
print(f"steps {n}: wall/step {(t1 - t0) / n / 1e6:.3f} ms (under the profiler)")
byq = collections.defaultdict(list)
for r in seg:
    byq[r["Queue_Id"]].append(r)
for q, rs in sorted(byq.items()):
    print(f"  queue {q}: busy/step {sum(r['e'] - r['s'] for r in rs) / n / 1e6:.3f} ms, launches/step {len(rs) / n:.0f}")
# union of busy intervals
iv = sorted((r["s"], r["e"]) for r in seg)
busy, cs, ce = 0, None, None
for s, e in iv:
    if cs is None:
        cs, ce = s, e
    elif s <= ce:
        ce = max(ce, e)
    else:
        busy += ce - cs
        cs, ce = s, e
busy += ce - cs
print(f"  any-queue busy/step {busy / n / 1e6:.3f} ms, idle/step {((t1 - t0) - busy) / n / 1e6:.3f} ms")

def cat(r):
    nm = r["Kernel_Name"]
    if "wgrad_stream" in nm: return "stem wgrad"
    if "wgrad_kernel" in nm: return "wgrad (mid layers)"
    if "gather_gemm2" in nm:
        if int(r["Grid_Size_X"]) // 256 > 6000: return "stem fwd"
        t = nm.split("<")[1].split(",")
        return "dgrad" if t[0].strip() == "true" else "fwd (mid layers)"
    for k in ("colreduce", "bn_", "slab_reduce", "splitk_reduce", "pool", "kernel_map", "scan", "insert", "assign", "flag_", "inverse",
              "make_keys", "class_partition", "fillBuffer", "copyBuffer", "at::native", "Cijk"):
        if k in nm: return k
    return nm[:32]
agg = collections.defaultdict(lambda: [0, 0])
for r in seg:
    c = cat(r)
    agg[c][0] += r["e"] - r["s"]
    agg[c][1] += 1
for c, (t, k) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print(f"  {c:28s} {t / n / 1e3:8.1f} us/step {k / n:6.1f} launches/step")
