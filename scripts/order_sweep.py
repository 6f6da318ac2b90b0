"""Sweep of launch B's dispatch order (OGG_PASS_ORDER) and of the quadrature chunking (OGG_QUAD_TARGET_WAVES, OGG_DPQUAD_TARGET_WAVES) on
one GPU.  usage: python scripts/order_sweep.py [--workload r8]"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from ocean_model_grid_generator_amd import supergrid  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="r8")
ap.add_argument("--steps", type=int, default=100)
args = ap.parse_args()
plan = supergrid.SupergridPlan(dp_arc="chord", **bench.WORKLOADS[args.workload])
sg = supergrid.Supergrid(plan, device="cuda:0")
sg.launch = "pass"


def timeit():
    for _ in range(20):
        sg.run_pass()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(args.steps):
            sg.run_pass()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / args.steps * 1e3)
    return best


for _ in range(100):
    sg.run_pass()
print("default %.4f" % timeit(), flush=True)
for order in ("01234", "23401", "32401", "23014", "20314", "23410", "42301", "24301"):
    os.environ["OGG_PASS_ORDER"] = order
    print("order %s %.4f" % (order, timeit()), flush=True)
os.environ.pop("OGG_PASS_ORDER")
for tw in (4096, 8192, 12288, 16384, 24576, 32768):
    os.environ["OGG_QUAD_TARGET_WAVES"] = str(tw)
    print("bp quad target waves %d %.4f" % (tw, timeit()), flush=True)
os.environ.pop("OGG_QUAD_TARGET_WAVES")
if any(s.kind == "dpole" for s in plan.subs):
    for tw in (4096, 8192, 16384, 32768):
        os.environ["OGG_DPQUAD_TARGET_WAVES"] = str(tw)
        print("dp quad target waves %d %.4f" % (tw, timeit()), flush=True)
