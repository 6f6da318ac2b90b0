#!/bin/bash
# Build the library of another git revision into ab/libogg_hip_<name>.so (in-tree, so that it travels to the GPU box; *.so is
# git-ignored), for A/B timing of two builds on ONE box:   scripts/ab_build.sh <rev> <name>
#   then on the GPU box:   python scripts/ab_time.py --libs ab/libogg_hip_<name>.so ocean_model_grid_generator_amd/csrc/libogg_hip.so
set -e
rev=$1; name=$2; extra=$3   # rev WORK: the working tree; extra: more compiler flags (e.g. -DOGG_LL_NT=1)
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d)
mkdir -p "$tmp/ocean_model_grid_generator_amd" "$root/ab"
if [ "$rev" = WORK ]; then
  mkdir -p "$tmp/ocean_model_grid_generator_amd/csrc" "$tmp/include"
  cp "$root"/ocean_model_grid_generator_amd/csrc/*.h "$root"/ocean_model_grid_generator_amd/csrc/*.hip "$tmp/ocean_model_grid_generator_amd/csrc/"
  cp "$root"/include/*.h "$tmp/include/"
else
  git -C "$root" archive "$rev" ocean_model_grid_generator_amd/csrc include | tar -x -C "$tmp"
fi
cd "$tmp/ocean_model_grid_generator_amd/csrc"
for f in *.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -w $extra -c $f -o ${f%.hip}.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC *.o -o "$root/ab/libogg_hip_$name.so"
rm -rf "$tmp"
ls -la "$root/ab/libogg_hip_$name.so"
