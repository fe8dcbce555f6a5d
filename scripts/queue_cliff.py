#!/usr/bin/env python3
"""What happens at GPU_MAX_HW_QUEUES = 8: from a rocprofv3 kernel trace of the one-rank data-parallel bench
(BENCH_FORCE_REDUCER=1), per HIP stream: the hardware queue(s) its dispatches went to, launches and busy time per step;
per queue: which streams share it; and, for the step's critical stream, how its kernel-to-kernel gaps are distributed and
which OTHER queue was running during the long ones.  usage: queue_cliff.py <trace dir>"""
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
by_stream = collections.defaultdict(list)
for r in seg:
    by_stream[r["Stream_Id"]].append(r)
q_streams = collections.defaultdict(set)
for sid, rs in sorted(by_stream.items(), key=lambda kv: -len(kv[1])):
    qs = collections.Counter(r["Queue_Id"] for r in rs)
    for q in qs:
        q_streams[q].add(sid)
    top = collections.Counter(r["Kernel_Name"].split("(")[0][-40:] for r in rs).most_common(2)
    print(f"  stream {sid:>3}: queues {dict(qs)}  launches/step {len(rs) / n:6.1f}  busy/step {sum(r['e'] - r['s'] for r in rs) / n / 1e6:6.3f} ms   e.g. {top}")
for q, ss in sorted(q_streams.items()):
    print(f"  queue {q}: streams {sorted(ss)}")
# the critical stream = the one holding the stem forward (gather_gemm2 ... 28)
crit = max(by_stream, key=lambda s: sum(1 for r in by_stream[s] if "gather_gemm2" in r["Kernel_Name"]))
rs = sorted(by_stream[crit], key=lambda r: r["s"])
gaps = [(b["s"] - a["e"], a, b) for a, b in zip(rs, rs[1:])]
hist = collections.Counter()
for g, _, _ in gaps:
    hist["<2us" if g < 2000 else "<5us" if g < 5000 else "<10us" if g < 10000 else "<20us" if g < 20000 else "<50us" if g < 50000 else ">=50us"] += 1
print(f"  critical stream {crit}: {len(rs) / n:.0f} launches/step, gaps between consecutive kernels per step: " +
      ", ".join(f"{k}: {v / n:.1f}" for k, v in sorted(hist.items())) + f"; sum of gaps/step {sum(g for g, _, _ in gaps if g > 0) / n / 1e6:.3f} ms")
big = sorted(gaps, key=lambda t: -t[0])[:12]
for g, a, b in big:
    during = collections.Counter()
    for r in seg:
        if r["Stream_Id"] != crit and r["s"] < b["s"] and r["e"] > a["e"]:
            during[(r["Queue_Id"], r["Stream_Id"], r["Kernel_Name"].split("(")[0][-32:])] += 1
    print(f"    gap {g / 1e3:7.1f} us  after {a['Kernel_Name'].split('(')[0][-36:]:36s} before {b['Kernel_Name'].split('(')[0][-36:]:36s} running meanwhile: {dict(during.most_common(3))}")
