"""Eager launches of the fused pass against a HIP-graph replay of the same three launches, for a whole grid and for one rank's
share of an 8-way split, on one GPU.  usage: python scripts/graph_vs_eager.py [out.json]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from ocean_model_grid_generator_amd import supergrid  # noqa: E402

out = []
for wl in ("r8", "r2"):
    plan = supergrid.SupergridPlan(**bench.WORKLOADS[wl])
    for world, rank in ((1, 0), (8, 3), (8, 7)):
        sg = supergrid.Supergrid(plan, rank=rank, world=world, device="cuda:0", halo="recompute")
        sg.launch = "pass"
        for _ in range(100):
            sg.run_pass()
        torch.cuda.synchronize()
        res = {"workload": wl, "world": world, "rank": rank}
        for mode in ("eager", "graph", "eager", "graph"):
            if mode == "graph" and not hasattr(sg, "graph"):
                sg.capture()
            fn = sg.replay if mode == "graph" else sg.run_pass
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(300):
                fn()
            torch.cuda.synchronize()
            res.setdefault(mode + "_ms", []).append((time.perf_counter() - t0) / 300 * 1e3)
        out.append(res)
        print(res, flush=True)
        del sg
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
