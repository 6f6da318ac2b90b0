"""Does the step time drift over the life of a process?  20 batches of 100 passes, time of each batch."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from ocean_model_grid_generator_amd import supergrid  # noqa: E402

plan = supergrid.SupergridPlan(**bench.WORKLOADS["r8"])
sg = supergrid.Supergrid(plan, device="cuda:0", halo="recompute")
sg.launch = "pass"
out = []
for b in range(20):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100):
        sg.run_pass()
    torch.cuda.synchronize()
    out.append((time.perf_counter() - t0) / 100 * 1e3)
    if b == 9:
        time.sleep(2.0)
print(" ".join("%.4f" % t for t in out))
