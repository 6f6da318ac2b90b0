#!/bin/bash
# compile ONE translation unit of libogg_hip with resource-usage remarks: scripts/cc1.sh ogg_dpole [extra flags]
cd /root/repo/ocean_model_grid_generator_amd/csrc
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function "$@" -c $f.hip -o $f.o -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|warning:|Function Name|  VGPRs:|AGPRs|Occupancy|ScratchSize" | sed -e 's/.*remark: *//' -e 's/ \[-Rpass.*//'
