"""Time every rank's share of a W-way band split on ONE GPU, one after the other (rehearsal of the multi-GPU run).

usage: python scripts/rank_sweep.py --world 8 [--workload r8] [--cost 2.6,1.35 ...]
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
ap.add_argument("--world", type=int, default=8)
ap.add_argument("--workload", default="r8")
ap.add_argument("--cost", nargs="*", default=["default"], help="OGG_BP_ROW_COST values to sweep (fix,guard,lump); `default`: the plan's own weights")
ap.add_argument("--rounds", type=int, default=2, help="every rank is timed this many times (forwards, then backwards, ...); the best time counts")
ap.add_argument("--dp-arc", default="chord")
ap.add_argument("--json", default=None, help="append one JSON line per (world, cost) to this file")
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--launch", default="pass")
ap.add_argument("--refine", type=int, default=1, help="after the first sweep, one rebalancing step from the ranks' own times (SupergridPlan.refine_split, as bench.py does at N > 1) and a second sweep")
ap.add_argument("--events", type=int, default=0, help="also print every rank's launch durations (HIP events around the launches of 20 extra passes)")
args = ap.parse_args()
for cost in args.cost:
    if cost == "default":
        os.environ.pop("OGG_BP_ROW_COST", None)
    else:
        os.environ["OGG_BP_ROW_COST"] = cost
    plan = supergrid.SupergridPlan(dp_arc=args.dp_arc, **bench.WORKLOADS[args.workload])
    if os.environ.get("OGG_SPLIT_CALIBRATE", "1") != "0":   # the split bench.py uses: tail / pass timed on this box (one process: no broadcast)
        plan.calibrate_split("cuda:0", rank=0, world=args.world, broadcast=False)
    def sweep(first):
        ts = []
        order = [args.world - 1]   # the first entry is a throw-away (clock ramp, allocator warm-up)
        for k in range(args.rounds):
            order += list(range(args.world)) if k % 2 == 0 else list(range(args.world - 1, -1, -1))
        best = {}
        for r in order:
            sg = supergrid.Supergrid(plan, rank=r, world=args.world, device="cuda:0", halo="recompute")
            sg.launch, sg.overlap = args.launch, False
            for _ in range(100 if not ts else 20):
                sg.step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.steps):
                sg.run_pass()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / args.steps * 1e3)
            if len(ts) > 1:
                best[r] = min(best.get(r, 1e9), ts[-1])
            if first and args.events and args.launch == "pass" and len(ts) > 1 and len(ts) <= args.world + 1:
                sg.reserve_pass_events(20)
                sg.pass_events = []
                for _ in range(20):
                    sg.run_pass()
                lt = sg.pass_launch_times_ms()
                sg.pass_events = None
                print("  rank %d: pass %.4f ms; launches A %.4f  B %.4f  tail %.4f  D %.4f" % (
                    r, ts[-1], lt["pass_a"]["ms"], lt["pass_b"]["ms"], lt["pass_tail"]["ms"], lt["pass_dpquad"]["ms"]), flush=True)
            del sg
        return [best[r] for r in range(args.world)]

    ts = sweep(True)
    ts_first = None
    if args.refine and args.world > 1 and plan.split_times is not None and plan.refine_split([t * 1e3 for t in ts], args.world):
        ts_first, ts = ts, sweep(False)
    if args.json:
        import json
        with open(args.json, "a") as f:
            f.write(json.dumps({"workload": args.workload, "dp_arc": args.dp_arc, "world": args.world, "row_cost": cost, "launch": args.launch,
                                "steps": args.steps, "band_split": plan.split_times, "ms_per_rank": ts, "ms_slowest_rank": max(ts),
                                **({"ms_per_rank_before_rebalancing": ts_first} if ts_first else {}),
                                "note": "each rank's share of the band split timed on ONE GPU, one after the other (rehearsal; no "
                                        "multi-GPU hardware curve exists yet)"}) + "\n")
    if ts_first:
        print("world %d cost %s: max %.4f ms  [%s]  (before the rebalancing step)" % (args.world, cost, max(ts_first), " ".join("%.4f" % t for t in ts_first)), flush=True)
    print("world %d cost %s: max %.4f ms  [%s]" % (args.world, cost, max(ts), " ".join("%.4f" % t for t in ts)), flush=True)
