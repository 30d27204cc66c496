#!/bin/bash
# Build the working tree's kernels (or those of a git revision) into ab/<name>.so for A/B timing on the GPU box:
#   tools/build_variant.sh <name> [git-rev] [extra hipcc flags...]
set -e
name=$1; rev=$2; shift; shift || true
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
mkdir -p $tmp/raytracer_amd/csrc $tmp/include $root/ab
if [ -n "$rev" ] && [ "$rev" != "-" ]; then
  for f in raytracer_amd/csrc/rt_hip.hip raytracer_amd/csrc/rt_bvh.cpp raytracer_amd/csrc/rt_bvh_internal.h raytracer_amd/csrc/rt_dev_math.h raytracer_amd/csrc/rt_dev_trace.h include/rt_hip.h; do
    git -C $root show $rev:$f > $tmp/$f
  done
else
  cp $root/raytracer_amd/csrc/* $tmp/raytracer_amd/csrc/; cp $root/include/rt_hip.h $tmp/include/
fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-slp-vectorize -fPIC -shared -DRT_WAVES=6 "$@" \
  $tmp/raytracer_amd/csrc/rt_hip.hip $tmp/raytracer_amd/csrc/rt_bvh.cpp -o $root/ab/$name.so
rm -rf $tmp
echo "built ab/$name.so"
