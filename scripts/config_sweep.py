"""Round-robin timing of the fused pass of one workload under several CONFIGURATIONS of environment knobs, in one process on one box
(two processes on the same box differ by up to 5 %: only numbers from one process compare).

usage: python scripts/config_sweep.py --workload r8 --config OGG_CAP_SYMMETRY=0 --config OGG_CAP_SYMMETRY=1,OGG_PASS_LL_HELPERS=0 [--rounds 4]
Every knob named in any configuration is unset in the configurations that do not name it.  Prints min / median ms per configuration."""
import argparse
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from ocean_model_grid_generator_amd import supergrid  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="r8")
ap.add_argument("--config", action="append", required=True)
ap.add_argument("--rounds", type=int, default=4)
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--as-rank", type=int, default=0)
ap.add_argument("--as-world", type=int, default=1)
args = ap.parse_args()
configs = [dict(kv.split("=", 1) for kv in c.split(",") if kv) for c in args.config]
knobs = sorted({k for c in configs for k in c})
plan = supergrid.SupergridPlan(**bench.WORKLOADS[args.workload])
sg = supergrid.Supergrid(plan, rank=args.as_rank, world=args.as_world, device="cuda:0", halo="recompute")
sg.launch = "pass"
for _ in range(100):
    sg.run_pass()
torch.cuda.synchronize()
res = [[] for _ in configs]
for rnd in range(args.rounds):
    for k, c in enumerate(configs):
        for name in knobs:
            if name in c:
                os.environ[name] = c[name]
            else:
                os.environ.pop(name, None)
        sg.replan()
        for _ in range(20):
            sg.run_pass()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            sg.run_pass()
        torch.cuda.synchronize()
        res[k].append((time.perf_counter() - t0) / args.steps * 1e3)
for c, r in zip(args.config, res):
    print("%s %-70s min %.4f median %.4f  %s" % (args.workload, c, min(r), statistics.median(r), " ".join("%.4f" % v for v in r)), flush=True)
