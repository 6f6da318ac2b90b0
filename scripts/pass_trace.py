"""Kernel timeline of a few passes: python3 scripts/pass_trace.py run <workload> [world rank]   (under rocprofv3 --kernel-trace)
                                    python3 scripts/pass_trace.py show <kernel_trace.csv>         (start / end of the last passes' kernels)"""
import csv
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if sys.argv[1] == "run":
    import torch
    import bench
    from ocean_model_grid_generator_amd import supergrid
    world = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    rank = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    plan = supergrid.SupergridPlan(**bench.WORKLOADS[sys.argv[2]])
    sg = supergrid.Supergrid(plan, rank=rank, world=world, device="cuda:0", halo="recompute")
    sg.launch = "pass"
    for _ in range(60):
        sg.run_pass()
    torch.cuda.synchronize()
else:
    rows = list(csv.DictReader(open(sys.argv[2])))
    rows = [r for r in rows if "pass_" in r["Kernel_Name"] or "tail" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    rows = rows[-12:]
    t0 = int(rows[0]["Start_Timestamp"])
    prev_end = None
    for r in rows:
        s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
        name = r["Kernel_Name"].split("(")[0][-40:]
        print("%-42s queue %s start %9.2f us  end %9.2f us  dur %7.2f  gap to previous end %6.2f" % (
            name, r.get("Queue_Id", "?"), s / 1e3, e / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3 if prev_end is not None else 0.0))
        prev_end = e
