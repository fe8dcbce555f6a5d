#!/bin/bash
# Round profile on the GPU box: kernel trace + stats of the default bench, then separate PMC passes
# (TCC read / TCC write / SQ+GRBM) as MI355X_MICROARCH.md prescribes.  usage: GIT_HEAD=$(git rev-parse --short HEAD) scripts/profile_round.sh <tag>
set -o pipefail
tag=${1:-r01_v6}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
out=gpurun_out/$tag
# (GIT_HEAD=<commit> in the environment is recorded next to the PMC summary: the box has no .git;
#  BENCH_ARGS="--math bf16 --storage bf16" profiles another configuration of bench.py)
mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/trace -o bench --output-format csv -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-other-configs $BENCH_ARGS > $out/trace.log 2>&1 || exit 1
for pass in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE"; do
  name=$(echo $pass | cut -d' ' -f1)
  rocprofv3 --pmc $pass -d $out/pmc_$name -o pmc --output-format csv -- python3 bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-other-configs --no-kernel-timing $BENCH_ARGS > $out/pmc_$name.log 2>&1 || exit 1
  echo "pass $name done"
done
python scripts/trace_summary.py $out/trace 0 60 > $out/trace_summary.txt
python scripts/stream_busy.py $out/trace > $out/stream_busy.txt
python scripts/pmc_summary.py $out/pmc.json $out/pmc_FETCH_SIZE $out/pmc_WRITE_SIZE $out/pmc_SQ_VALU_MFMA_BUSY_CYCLES > $out/pmc_summary.txt
head -30 $out/stream_busy.txt; head -12 $out/pmc_summary.txt
