#!/bin/bash
# usage: scripts/kernel_resources.sh [file.hip ...]   -- VGPRs / spills / occupancy / LDS of the kernels of the library's translation units
# (hipcc -Rpass-analysis=kernel-resource-usage; cross-compiles without a GPU)
cd "$(dirname "$0")/../ocean_model_grid_generator_amd/csrc"
for f in ${*:-ogg_pass.hip ogg_bipolar.hip ogg_dpole.hip ogg_midas.hip ogg_latlon_fused.hip}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -w -Rpass-analysis=kernel-resource-usage -c $f -o /tmp/_kr.o 2>&1 |
    sed -n 's/.*remark: *//p' | sed 's/ \[-Rpass.*//' |
    awk -v file=$f '/^Function Name:/ {name=$3} /^VGPRs:/ {v=$2} /^ScratchSize/ {sc=$NF} /^Occupancy/ {oc=$NF} /^SGPRs Spill:/ {ss=$NF} /^VGPRs Spill:/ {vs=$NF}
                    /^LDS Size/ {printf "%s %s VGPRs %s vgpr-spill %s sgpr-spill %s scratch %s waves/SIMD %s LDS %s\n", file, name, v, vs, ss, sc, oc, $NF}' | c++filt | cut -c1-260
done
