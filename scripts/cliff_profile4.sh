#!/bin/bash
# One-rank RCCL run of the data-parallel machinery: collectives issued from inside the backward call (default, round 5) against
# the round-4 launch stream (MINK_DP_LAUNCH=stream), at GPU_MAX_HW_QUEUES 7 / 8 / 16, and without the machinery.  (GPU box)
cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/cliff4_summary.txt
: > $out
for args in "--model ResNet14 --batch 16" "--model ResNet34 --batch 4"; do
  ms=$(python3 bench.py $args --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c 'import json,sys; print("%.3f" % json.loads(sys.stdin.read())["ms_per_step"])')
  echo "$args without the data-parallel machinery -> $ms ms/step" >> $out
  for q in 7 8 16; do
    for mode in call stream; do
      ms=$(GPU_MAX_HW_QUEUES=$q BENCH_FORCE_REDUCER=1 MINK_DP_LAUNCH=$mode MINK_HWQUEUES_KEEP=1 python3 bench.py $args --steps 30 --warmup 5 --no-cpu-baseline 2>>gpurun_out/cliff4.err | tail -1 | python3 -c 'import json,sys; print("%.3f" % json.loads(sys.stdin.read())["ms_per_step"])')
      echo "$args one-rank RCCL group, GPU_MAX_HW_QUEUES=$q requested, collectives issued from: $mode -> $ms ms/step" >> $out
    done
  done
done
cat $out; grep hwqueues gpurun_out/cliff4.err | sort | uniq -c
