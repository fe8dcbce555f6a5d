#!/bin/bash
# bench.py's own line (K=20 / W=5) with the map-build gate off / on: step, dominant kernel's HIP-event time and fraction.  (GPU box)
cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2 3; do
  for g in "$@"; do
    MINK_PREPARE_GATE=$g python bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); r=d['roofline']
print('gate %-4s step %.3f ms  %s %.3f ms  frac %.3f' % ('$g', d['ms_per_step'], r['kernel'], r['avg_ms'], r['frac']))"
  done
done
