"""Sweep the tuning knobs of ogg_tripolar_pass_dev (read from the environment at every call) on one GPU.

usage: python scripts/pass_sweep.py [--as-rank R --as-world W] [--workload r8]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from ocean_model_grid_generator_amd import supergrid  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--as-rank", type=int, default=0)
ap.add_argument("--as-world", type=int, default=1)
ap.add_argument("--workload", default="r8")
ap.add_argument("--wg", default="36,48,60,84,120")
ap.add_argument("--wg-small", default="")
ap.add_argument("--steps", type=int, default=50)
args = ap.parse_args()
plan = supergrid.SupergridPlan(**bench.WORKLOADS[args.workload])
sg = supergrid.Supergrid(plan, rank=args.as_rank, world=args.as_world, device="cuda:0", halo="recompute")
sg.launch = "pass"


def timeit():
    for _ in range(3):
        sg.step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sg.run_pass()
    torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / args.steps * 1e3
    sg.pass_events = []
    for _ in range(20):
        sg.run_pass()
    lt = sg.pass_launch_times_ms()
    sg.pass_events = None
    timeit.launches = "A %.1f B %.1f C %.1f us" % (lt["pass_a"]["ms"] * 1e3, lt["pass_b"]["ms"] * 1e3, lt["pass_tail"]["ms"] * 1e3)
    return t


print("default: %.4f ms" % timeit(), timeit.launches)
key = "OGG_PASS_LL_WG_SMALL" if args.wg_small else "OGG_PASS_LL_WG"
wgs = (args.wg_small or args.wg).split(",")
for wg in wgs:
    os.environ[key] = wg
    print("%s=%s: %.4f ms" % (key, wg, timeit()), timeit.launches, flush=True)
