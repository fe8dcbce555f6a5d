#!/bin/bash
# kbench modes under every A/B build of the library (scripts/build_variant.sh): usage ab_variants.sh "<modes>" <reps> <variants...>
# ("default" = the in-tree library)
modes=$1; reps=$2; shift 2
for v in "$@"; do
  if [ "$v" = default ]; then unset MINK_HIP_LIB; else export MINK_HIP_LIB=$PWD/nerf_downstream_amd/variants/libmink_hip_$v.so; fi
  for m in $modes; do
    echo "=== variant $v mode $m"
    timeout -k 10 300 python scripts/kbench.py $m $reps 2>&1 | grep -v "amdgpu.ids" || exit 1
  done
done
