#!/bin/bash
# per-kernel durations of the offset-major path inside a bench run (GPU box); usage: scripts/om_prof.sh <outdir-tag>
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/${1:-om_prof}
rm -rf $out
rocprofv3 --kernel-trace --stats -d $out -o k --output-format csv -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing > $out.log 2>&1
python - <<PY
import csv,glob,collections
f=glob.glob("$out/**/k_kernel_trace.csv",recursive=True)[0]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r["Kernel_Name"]
    if "offset_" in n:
        agg[(n[:48], r["Grid_Size_X"], r["Grid_Size_Y"], r["Grid_Size_Z"])].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
for k,v in sorted(agg.items()):
    print(k, len(v), "avg %.1f us"%(sum(v)/len(v)))
PY
