"""When do the workgroups of each role of launch B start and end?  Needs a library built with -DOGG_TIMELINE=1:
    scripts/ab_build.sh WORK tl -DOGG_TIMELINE=1
    OGG_LIB_PATH=$PWD/ab/libogg_hip_tl.so python scripts/pass_timeline.py --workload r16 [--config OGG_CAP_SYMMETRY=0] [--as-rank 0 --as-world 1]
Per role: workgroups, first start, last start, mean end, last end (microseconds from the launch's first workgroup; the device's 100 MHz
constant clock, so 0.01 us resolution), median over `--reps` single passes, each after 20 untimed ones."""
import argparse
import ctypes
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from ocean_model_grid_generator_amd import _lib, supergrid  # noqa: E402

ROLES = ["latlon strips", "latlon helpers", "bipolar mesh", "dpole mesh", "bipolar quad guard", "bipolar quad fast", "dpole quad", "next tables"]
ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="r8")
ap.add_argument("--config", action="append", default=[])
ap.add_argument("--reps", type=int, default=7)
ap.add_argument("--strips", action="store_true")
ap.add_argument("--calibrate", action="store_true", help="the band split bench.py uses (calibrate_split on this box) instead of the fitted constants")
ap.add_argument("--as-rank", type=int, default=0)
ap.add_argument("--as-world", type=int, default=1)
args = ap.parse_args()
lib = _lib.load()
tl = lib.ogg_timeline
lib.ogg_timeline_strips.argtypes, lib.ogg_timeline_strips.restype = [ctypes.POINTER(ctypes.c_ulonglong)], ctypes.c_int
tl.argtypes, tl.restype = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int], ctypes.c_int
plan = supergrid.SupergridPlan(**bench.WORKLOADS[args.workload])
if args.calibrate and args.as_world > 1:
    plan.calibrate_split("cuda:0", rank=0, world=args.as_world, broadcast=False)
    print("split:", plan.split_times.get("top_capacity"), {s.name: supergrid.Supergrid.rows_of(s, args.as_rank, args.as_world) for s in plan.subs})
sg = supergrid.Supergrid(plan, rank=args.as_rank, world=args.as_world, device="cuda:0", halo="recompute")
sg.launch = "pass"
for _ in range(100):
    sg.run_pass()
torch.cuda.synchronize()
for cfg in (args.config or [""]):
    kv = dict(x.split("=", 1) for x in cfg.split(",") if x)
    os.environ.update(kv)
    sg.replan()
    rows = [[] for _ in ROLES]
    for _ in range(args.reps):
        for _ in range(20):
            sg.run_pass()
        torch.cuda.synchronize()
        assert tl(None, 1) == 0
        sg.run_pass()
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 40)()
        assert tl(buf, 0) == 0
        first, last_start, end, sum_end, n = (list(buf[8 * k:8 * k + 8]) for k in range(5))
        t0 = min(f for f, c in zip(first, n) if c)
        if args.strips:
            sb = (ctypes.c_ulonglong * 512)()
            assert lib.ogg_timeline_strips(sb) == 0
            last_strips = [(v - t0) / 100.0 for v in list(sb)[:n[0]]]
        for r in range(8):
            if n[r]:
                rows[r].append((n[r], (first[r] - t0) / 100.0, (last_start[r] - t0) / 100.0, (sum_end[r] / n[r] - t0) / 100.0, (end[r] - t0) / 100.0))
    print("%s %s" % (args.workload, cfg or "(defaults)"))
    print("  %-20s %8s %10s %10s %10s %10s" % ("role", "wgs", "first", "last start", "mean end", "last end"))
    for r, name in enumerate(ROLES):
        if rows[r]:
            med = [statistics.median(v[k] for v in rows[r]) for k in range(5)]
            print("  %-20s %8d %10.1f %10.1f %10.1f %10.1f" % (name, med[0], med[1], med[2], med[3], med[4]), flush=True)
    if args.strips:   # resident strip workgroups by index (workgroup b runs on XCD b % 8), last repetition
        for x in range(8):
            print("  strips of XCD %d end at: %s" % (x, " ".join("%.0f" % v for v in last_strips[x::8])), flush=True)
    for k in kv:
        os.environ.pop(k, None)
