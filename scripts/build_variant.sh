#!/bin/bash
# A/B builds of the same ABI: scripts/build_variant.sh <name> "<-D flags>" [file=conv.hip] -> nerf_downstream_amd/variants/libmink_hip_<name>.so
# (select at run time with MINK_HIP_LIB=<path>; the other objects are the in-tree ones, so run `make` in csrc first)
set -e
name=$1; flags=$2; file=${3:-conv.hip}
root=$(cd "$(dirname "$0")/.." && pwd)
csrc=$root/nerf_downstream_amd/csrc
out=$root/nerf_downstream_amd/variants
mkdir -p $out
obj=$out/${file%.hip}_$name.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I$root/include -Wall -Wno-unused-function $flags -c $csrc/$file -o $obj
others=$(for f in coords conv elementwise dense stem16 augment trunk; do [ "$f.hip" = "$file" ] || echo $csrc/$f.o; done)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $out/libmink_hip_$name.so $obj $others
rm -f $obj
echo "built $out/libmink_hip_$name.so"
