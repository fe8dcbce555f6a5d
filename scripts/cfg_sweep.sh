#!/bin/bash
# step time of the other configurations BASELINE.json names (and the shapes a data-parallel run puts on one GPU)
for args in "--model ResNet34 --batch 4" "--batch 8" "--math bf16" "--math bf16 --storage fp32" "--math bf16x3" "--model ResNet34" "--batch 32" "--math bf16 --model ResNet34 --batch 4"; do
  python bench.py --steps 30 --warmup 8 --no-cpu-baseline $args 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); r = d.get('roofline', {})
print('[$args] %.3f ms/step  %.1f M voxels/s  forward only %.3f ms' % (d['ms_per_step'], d['value'] / 1e6, r.get('forward_only', {}).get('ms', float('nan'))))"
done
