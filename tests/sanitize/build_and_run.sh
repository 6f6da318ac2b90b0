#!/bin/bash
# Build the library's host code with AddressSanitizer + UndefinedBehaviorSanitizer against the fake HIP runtime and run the driver.
#   tests/sanitize/build_and_run.sh <build dir> [tsan]
# Host compilation only (--cuda-host-only: no device code is generated, kernel launches become calls into the fake runtime).
set -e
here=$(cd "$(dirname "$0")" && pwd); root=$(cd "$here/../.." && pwd); out=${1:-/tmp/ogg_sanitize}; mode=${2:-asan}
mkdir -p "$out"; cd "$out"
hipcc=${HIPCC:-/opt/rocm/bin/hipcc}
san="-fsanitize=address,undefined -fno-sanitize-recover=undefined"; [ "$mode" = tsan ] && san="-fsanitize=thread"
flags="--cuda-host-only -O1 -g -std=c++17 -fPIC -ffp-contract=off -fno-omit-frame-pointer -w $san"
csrc="$root/ocean_model_grid_generator_amd/csrc"
pids=()
for f in ogg_api ogg_axes ogg_midas ogg_bipolar ogg_dpole ogg_elementwise ogg_latlon_fused ogg_reduce; do
  $hipcc $flags -c "$csrc/$f.hip" -o $f.o & pids+=($!)
done
$hipcc $flags -c "$here/driver.hip" -o driver.o & pids+=($!)
$hipcc $flags -I/opt/rocm/include -D__HIP_PLATFORM_AMD__ -x c++ -c "$here/fake_hip_runtime.cpp" -o fake.o & pids+=($!)
for p in "${pids[@]}"; do wait $p; done
# every translation unit references the device binary that a full build embeds: __hip_fatbin_<id>; give each a byte
nm *.o | awk '/ U __hip_fatbin/ {print $2}' | sort -u | awk '{print "extern \"C\" const char " $1 "[8] = {0};"}' > fatbins.cpp
$hipcc $flags -x c++ -c fatbins.cpp -o fatbins.o
$hipcc $san -o driver_$mode *.o -lpthread
ASAN_OPTIONS=detect_leaks=1:abort_on_error=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 ./driver_$mode
