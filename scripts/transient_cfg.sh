#!/bin/bash
# scripts/transient_sweep.sh for other bench configurations: "args" per line.  (GPU box)
cd "$GRAFT_REPO_ROOT" || exit 1
for args in "$@"; do
  BENCH_STEP_TIMES=1 python bench.py --gpus 1 --steps ${STEPS:-60} --warmup 5 --no-cpu-baseline --no-kernel-timing $args 2>&1 | grep -a "\[bench\] GPU ms\|\[bench\] host ms" | cut -d: -f2 | python3 -c "
import sys
g, h = [[float(x) for x in l.split()] for l in sys.stdin.read().strip().splitlines()[:2]]
a, b = g[2:int(0.35 * len(g))], g[-8:]
print('%-44s with maps %.3f ms   drain %.3f ms   host %.3f ms/step | %s' % ('$args', sum(a) / len(a), sum(b) / len(b), sum(h) / len(h), ' '.join('%.2f' % x for x in g[:40])))
"
done
