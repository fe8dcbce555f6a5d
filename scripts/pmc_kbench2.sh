#!/bin/bash
# Instruction-fetch / branch attribution for one kbench mode (GPU box).  usage: pmc_kbench2.sh [mode=l1] [kernel-name regex]
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mode=${1:-l1}
export KB_FILTER=${2:-compact_gemm}
out=gpurun_out/pmc_kb2
rm -rf $out; mkdir -p $out
rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE -d $out/p1 -o pmc --output-format csv -- python3 scripts/kbench.py $mode 3 > $out/p1.log 2>&1 || { tail -5 $out/p1.log; exit 1; }
python3 - <<'PY'
import csv, glob, collections, os, re
flt = re.compile(os.environ["KB_FILTER"])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_kb2/p*/pmc_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-60:] + " grid " + r["Grid_Size"]
        if flt.search(k):
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    print(k)
    for c in sorted(m):
        print(f"    {c:28s} {m[c]:16.0f}")
    if m.get("SQ_IFETCH"):
        print(f"    -> average fetch latency {m['SQ_IFETCH_LEVEL'] / m['SQ_IFETCH']:.1f} (level units per fetch); fetches per MFMA {m['SQ_IFETCH'] / max(m.get('SQ_INSTS_MFMA', 1), 1):.2f}; "
              f"branches per MFMA {m['SQ_INSTS_BRANCH'] / max(m.get('SQ_INSTS_MFMA', 1), 1):.2f}; IFETCH_LEVEL / WAVE_CYCLES {m['SQ_IFETCH_LEVEL'] / m['SQ_WAVE_CYCLES']:.3f}")
PY
