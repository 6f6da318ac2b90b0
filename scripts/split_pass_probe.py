"""Is a pass over a LARGE grid slower per byte than the same bands run as several smaller passes?  One GPU, one process.
The W band shares of a workload (the split bench.py uses at W ranks) are timed (a) each alone, 200 passes of the same share back to back
(what scripts/rank_sweep.py does: the share's own output is rewritten every time), (b) in sequence share 0, 1, .. W-1, 0, 1, ..: the bytes
written between two visits of one share are then the whole grid's, as in the one-launch pass.

usage: python scripts/split_pass_probe.py --workload r16 --world 4"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from ocean_model_grid_generator_amd import supergrid  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="r16")
ap.add_argument("--world", type=int, default=4)
ap.add_argument("--steps", type=int, default=100)
ap.add_argument("--copies", type=int, default=0, help="instead of the band shares: this many WHOLE grids with buffers of their own, each alone and then "
                "round-robin -- between two passes over one copy the others' bytes are written (how fast is a pass that finds nothing of its "
                "predecessor on the chip?)")
args = ap.parse_args()
if args.copies > 0:
    grids = []
    for k in range(args.copies):
        g = supergrid.Supergrid(supergrid.SupergridPlan(**bench.WORKLOADS[args.workload]), rank=0, world=1, device="cuda:0", halo="recompute")
        g.launch = "pass"
        grids.append(g)

    def t_of(fn, n):
        for _ in range(20):
            fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / n * 1e3

    def robin():
        for g in grids:
            g.run_pass()

    for k, g in enumerate(grids):   # where the copy's biggest band lies (virtual addresses of its six fields, GiB)
        big = max(g.buf.values(), key=lambda b: b["x"].numel())
        print("copy %d: %s" % (k, " ".join("%s %.3f" % (f, big[f].data_ptr() / 2.0**30) for f in supergrid.FIELDS)), flush=True)
    for rnd in range(2):
        alone = [t_of(g.run_pass, args.steps) for g in grids]
        rr = t_of(robin, args.steps) / args.copies
        print("%s, %d copies, round %d: one copy over and over %s ms per pass; round-robin over the copies %.4f ms per pass"
              % (args.workload, args.copies, rnd, " ".join("%.4f" % a for a in alone), rr), flush=True)
    raise SystemExit(0)
plan1 = supergrid.SupergridPlan(**bench.WORKLOADS[args.workload])
whole = supergrid.Supergrid(plan1, rank=0, world=1, device="cuda:0", halo="recompute")
whole.launch = "pass"
plan = supergrid.SupergridPlan(**bench.WORKLOADS[args.workload])
shares = []
for r in range(args.world):
    g = supergrid.Supergrid(plan, rank=r, world=args.world, device="cuda:0", halo="recompute")
    g.launch = "pass"
    shares.append(g)


def timed(fn, n):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def sequence():
    for g in shares:
        g.run_pass()


for rnd in range(2):
    t_whole = timed(whole.run_pass, args.steps)
    alone = [timed(g.run_pass, 2 * args.steps) for g in shares]
    t_seq = timed(sequence, args.steps)
    print("%s world %d round %d: whole grid in one pass %.4f ms; shares alone %s sum %.4f; shares in sequence %.4f ms"
          % (args.workload, args.world, rnd, t_whole, " ".join("%.4f" % a for a in alone), sum(alone), t_seq), flush=True)
