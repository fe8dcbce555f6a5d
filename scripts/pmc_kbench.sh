#!/bin/bash
# Stall attribution: three PMC passes over scripts/kbench.py <mode> (GPU box).  usage: pmc_kbench.sh [mode=stem] [kernel-name regex]
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
mode=${1:-stem}
export KB_FILTER=${2:-gather_gemm2|wgrad_stream}
out=gpurun_out/pmc_kb
rm -rf $out; mkdir -p $out
i=0
for pass in "SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE" \
            "SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_LDS_DATA_FIFO_FULL GRBM_GUI_ACTIVE" \
            "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_CMD_FIFO_FULL SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VALU_MFMA_COEXEC_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $pass -d $out/p$i -o pmc --output-format csv -- python3 ${KB_SCRIPT:-scripts/kbench.py} $mode 3 > $out/p$i.log 2>&1 || { tail -5 $out/p$i.log; exit 1; }
done
python3 - <<'PY'
import csv, glob, collections, os, re
flt = re.compile(os.environ["KB_FILTER"])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_kb/p*/pmc_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-60:] + " grid " + r["Grid_Size"]
        if flt.search(k):
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    g = m.get("GRBM_GUI_ACTIVE", 1.0) / 8.0   # cycles per XCD
    print(k)
    for c in sorted(m):
        print(f"    {c:32s} {m[c]:16.0f}   per-SIMD-cycle {m[c] / 1024.0 / g:8.3f}   per-CU-cycle {m[c] / 256.0 / g:8.3f}")
PY
