#!/bin/bash
# The stem weight gradient's HIP-event time inside a step, with the work that can overlap it taken away piece by piece.  (GPU box)
cd "$GRAFT_REPO_ROOT" || exit 1
for cfg in "X=1" "BENCH_ABLATE_MAPS=1" "BENCH_ABLATE_MAPS=1 BENCH_WGRAD_OVERLAP=0" "BENCH_WGRAD_OVERLAP=0"; do
  env $cfg python bench.py --no-cpu-baseline --steps 30 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('%-46s step %.3f ms  %s %.3f ms  frac %.3f' % ('$cfg', d['ms_per_step'], r['kernel'], r['avg_ms'], r['frac']))"
done
python scripts/kbench.py wxcd 2>&1 | tail -4
