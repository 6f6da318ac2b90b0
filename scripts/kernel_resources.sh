#!/bin/bash
# usage: scripts/kernel_resources.sh [file.hip ...]   -- VGPRs / spills / occupancy / LDS of the kernels of the library's translation units
cd "$(dirname "$0")/../ocean_model_grid_generator_amd/csrc"
for f in ${*:-ogg_pass.hip ogg_bipolar.hip ogg_dpole.hip ogg_midas.hip ogg_latlon_fused.hip}; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -w -Rpass-analysis=kernel-resource-usage -c $f -o /tmp/_kr.o 2>&1 |
    awk '/Function Name/ {name=$NF} /VGPRs:/ {v=$NF} /VGPRs Spill/ {vs=$NF} /SGPRs Spill/ {ss=$NF} /ScratchSize/ {sc=$NF} /Occupancy/ {oc=$NF} /LDS Size/ {print name, "VGPRs", v, "vgpr-spill", vs, "sgpr-spill", ss, "scratch", sc, "waves/SIMD", oc, "LDS", $NF}' | sed 's/\[-Rpass.*\]//' | c++filt | cut -c1-220
done
