#!/bin/bash
# Five busy hardware queues or four: the bucket-launch stream on a queue of its own (default) against on the weight-gradient
# stream (MINK_DP_LAUNCH_STREAM=side), at GPU_MAX_HW_QUEUES 7 / 8 / 16.  (GPU box) -> gpurun_out/cliff3_summary.txt
cd "$GRAFT_REPO_ROOT" || exit 1
: > gpurun_out/cliff3_summary.txt
for args in "--model ResNet14 --batch 16" "--model ResNet34 --batch 4"; do
for q in 7 8 16; do
  for ls in own side; do
    export GPU_MAX_HW_QUEUES=$q BENCH_FORCE_REDUCER=1 MINK_DP_LAUNCH_STREAM=$ls
    ms=$(python3 bench.py $args --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c 'import json,sys; print("%.3f" % json.loads(sys.stdin.read())["ms_per_step"])')
    echo "$args GPU_MAX_HW_QUEUES=$q bucket-launch stream: $ls -> $ms ms/step" >> gpurun_out/cliff3_summary.txt
  done
done
done
cat gpurun_out/cliff3_summary.txt
