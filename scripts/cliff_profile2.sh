#!/bin/bash
# Which hardware queue RCCL's stream sits on decides the cliff: GPU_MAX_HW_QUEUES=16 and N dummy streams that take the
# queues in front of it (bench.py BENCH_DUMMY_STREAMS).  usage (GPU box): cliff_profile2.sh "0 1 2 3 4 5" -> gpurun_out/cliff2_summary.txt
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
vals=${1:-"0 1 2 3 4"}
: > gpurun_out/cliff2_summary.txt
for n in $vals; do
  export GPU_MAX_HW_QUEUES=${CLIFF_Q:-16} BENCH_FORCE_REDUCER=1 BENCH_DUMMY_STREAMS=$n
  ms=$(python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c 'import json,sys; print("%.3f" % json.loads(sys.stdin.read())["ms_per_step"])')
  rm -rf gpurun_out/cliff2_$n
  rocprofv3 --kernel-trace -d gpurun_out/cliff2_$n -o bench --output-format csv -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing > gpurun_out/cliff2_$n.log 2>&1 || { tail -5 gpurun_out/cliff2_$n.log; exit 1; }
  python3 scripts/queue_cliff.py gpurun_out/cliff2_$n > gpurun_out/cliff2_$n.txt
  rm -rf gpurun_out/cliff2_$n
  echo "GPU_MAX_HW_QUEUES=$GPU_MAX_HW_QUEUES, $n dummy streams: un-profiled $ms ms/step; $(head -1 gpurun_out/cliff2_$n.txt)" >> gpurun_out/cliff2_summary.txt
  grep "^  stream" gpurun_out/cliff2_$n.txt | cut -c1-60 >> gpurun_out/cliff2_summary.txt
done
cat gpurun_out/cliff2_summary.txt
