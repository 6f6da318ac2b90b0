"""Host cost of a pass: time to ENQUEUE one pass from Python (ctypes call + launches) against the time the GPU needs for it, for a whole
1/8 degree grid and for one rank's share of eight (1/8 and 1/2 degree).  usage: python scripts/host_cost.py
Round 2 (every pass re-planned): 15 us of host time per pass; an eighth of the 1/8 degree grid needs 41 us on the GPU (GPU-bound), of the
1/2 degree grid 18 us.  With the plan handle (ogg_supergrid_pass_plan_dev / _run_dev) a pass is one ctypes call + its launches."""
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
    # host-only cost: bursts of 16 passes into an EMPTY queue (a long loop is throttled by the queue once the GPU is the slower side)
    burst, reps, host = 16, 50, 0.0
    for _ in range(reps):
        torch.cuda.synchronize()
        ta = time.perf_counter()
        for _ in range(burst): sg.run_pass()
        host += time.perf_counter() - ta
    torch.cuda.synchronize()
    print(wl, world, "host time per pass %.1f us (bursts of %d into an empty queue); steady loop: enqueue %.1f us, total %.1f us per pass" % (
        host / (burst * reps) * 1e6, burst, (t1 - t0) / n * 1e6, (t2 - t0) / n * 1e6), flush=True)
