#!/bin/bash
# The map-build gate (MINK_PREPARE_GATE) in the other configurations: ms/step of bench.py, gate off / on, alternated.  (GPU box)
cd "$GRAFT_REPO_ROOT" || exit 1
for args in "--math bf16" "--batch 8" "--model ResNet34 --batch 4" ""; do
  for rep in 1 2; do
    for g in ${GATES:-0 f0}; do
      ms=$(MINK_PREPARE_GATE=$g python bench.py --steps ${STEPS:-40} --warmup 5 --no-cpu-baseline $args 2>/dev/null | tail -1 | python3 -c 'import json,sys; print("%.3f" % json.loads(sys.stdin.read())["ms_per_step"])')
      echo "[$args] gate $g -> $ms ms/step"
    done
  done
done
