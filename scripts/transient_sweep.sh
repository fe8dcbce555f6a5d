#!/bin/bash
# Per-step GPU times of bench.py's timed region under settings: while the host is still queuing, every step
# overlaps the NEXT batch's map build ("with maps"); once the host has queued its last step the prepare stream is done and the
# rest of the steps run alone ("drain").  (GPU box)
cd "$GRAFT_REPO_ROOT" || exit 1
for cfg in "X=1" "$@"; do
  env $cfg BENCH_STEP_TIMES=1 python bench.py --gpus 1 --steps ${STEPS:-50} --warmup 5 --no-cpu-baseline --no-kernel-timing 2>&1 | grep -a "\[bench\] GPU ms" | cut -d: -f2 | python3 -c "
import sys
v = [float(x) for x in sys.stdin.read().split()]
a, b = v[2:int(0.4 * len(v))], v[-10:]
print('%-28s with maps (steps 2 .. 0.4 K) %.3f ms   drain (last 10) %.3f ms   | %s' % ('$cfg', sum(a) / len(a), sum(b) / len(b), ' '.join('%.2f' % x for x in v[:30])))
"
done
