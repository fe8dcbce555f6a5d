#!/bin/bash
# A/B of CU-subset streams (MINK_CUS_PREPARE / MINK_CUS_WGRAD / MINK_CUS_BRANCH) on the default bench; one process per setting.
# usage: scripts/cu_sweep.sh "<env assignments>" ...   (each argument is one setting; "" = baseline)
out=gpurun_out/cu_sweep.txt
: > $out
for cfg in "$@"; do
  line=$(env $cfg python bench.py --steps 40 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('%.3f ms/step  dominant %s %.3f ms  fwd-only %.3f ms' % (d['ms_per_step'], r['kernel'], r['avg_ms'], r['forward_only']['ms']))")
  echo "[$cfg] $line" | tee -a $out
done
