"""Time of the generic stencil kernel's three launches over the lat-lon sub-grids of the 1/8 degree grid (events around every kernel) for
the library in OGG_LIB_PATH:   OGG_LIB_PATH=$PWD/ab/libogg_hip_x.so python3 scripts/midas_time.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from ocean_model_grid_generator_amd import supergrid  # noqa: E402

plan = supergrid.SupergridPlan(**bench.WORKLOADS["r8"])
sg = supergrid.Supergrid(plan, device="cuda:0", latlon="stencil", halo="recompute")
sg.launch, sg.overlap = "kernels", False
for _ in range(20):
    sg.step()
torch.cuda.synchronize()
sg.step(time_kernels=True)
for _ in range(20):
    sg.run_pass()
k = sg.kernel_times_ms()
print(os.path.basename(os.environ.get("OGG_LIB_PATH", "lib")), {n: (v["launches"], round(v["total_ms"] / 21, 4)) for n, v in k.items() if "midas" in n or "tile" in n}, flush=True)
