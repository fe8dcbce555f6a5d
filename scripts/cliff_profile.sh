#!/bin/bash
# The 8-hardware-queue cliff of the one-rank data-parallel bench: for each GPU_MAX_HW_QUEUES value the
# un-profiled step time, then a kernel trace of the same command for the stream -> hardware-queue assignment
# (scripts/queue_cliff.py).  usage (GPU box): cliff_profile.sh "7 8 9 10 12 16"  -> gpurun_out/cliff_summary.txt
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
vals=${1:-"7 8"}
: > gpurun_out/cliff_summary.txt
for q in $vals; do
  export GPU_MAX_HW_QUEUES=$q BENCH_FORCE_REDUCER=1
  ms=$(python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c 'import json,sys; print("%.3f" % json.loads(sys.stdin.read())["ms_per_step"])')
  rm -rf gpurun_out/cliff_$q
  rocprofv3 --kernel-trace -d gpurun_out/cliff_$q -o bench --output-format csv -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-kernel-timing > gpurun_out/cliff_$q.log 2>&1 || { tail -5 gpurun_out/cliff_$q.log; exit 1; }
  python3 scripts/queue_cliff.py gpurun_out/cliff_$q > gpurun_out/cliff_$q.txt
  rm -rf gpurun_out/cliff_$q
  echo "GPU_MAX_HW_QUEUES=$q: un-profiled $ms ms/step; $(head -1 gpurun_out/cliff_$q.txt)" >> gpurun_out/cliff_summary.txt
  grep "^  stream" gpurun_out/cliff_$q.txt | cut -c1-90 >> gpurun_out/cliff_summary.txt
done
cat gpurun_out/cliff_summary.txt
