"""Host cost of a pass: time to ENQUEUE one pass from Python (ctypes call + launches) against the time the GPU needs for it, for a whole
1/8 degree grid and for one rank's share of eight (1/8 and 1/2 degree).  usage: python scripts/host_cost.py
Measured: 15 us of host time per pass; an eighth of the 1/8 degree grid needs 41 us on the GPU (GPU-bound), of the 1/2 degree grid 18 us."""
import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
from ocean_model_grid_generator_amd import supergrid
for wl, world in (("r8", 8), ("r2", 8), ("r8", 1)):
    plan = supergrid.SupergridPlan(**bench.WORKLOADS[wl])
    sg = supergrid.Supergrid(plan, rank=3 if world > 1 else 0, world=world, device="cuda:0", halo="recompute")
    sg.launch = "pass"
    for _ in range(50): sg.run_pass()
    torch.cuda.synchronize()
    n = 2000
    t0 = time.perf_counter()
    for _ in range(n): sg.run_pass()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(wl, world, "enqueue per pass %.1f us, total per pass %.1f us" % ((t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6), flush=True)
