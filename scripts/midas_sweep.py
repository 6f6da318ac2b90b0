"""The generic stencil kernel (ogg_grid_metrics_midas_dev) on the three lat-lon sub-grids of the 1/8 (and 1/16) degree grid: ms per launch
for tile heights OGG_MIDAS_TILE_ROWS = 0 (the streaming kernel of rounds 1-3), 4, 6, 8, 12 -- interleaved, three rounds, on one box --
and bit-identity of every tile height with the streaming kernel.   usage: python3 scripts/midas_sweep.py [r8|r16] [out.json]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from ocean_model_grid_generator_amd import supergrid  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "r8"
plan = supergrid.SupergridPlan(**bench.WORKLOADS[wl])
sg = supergrid.Supergrid(plan, device="cuda:0", latlon="stencil", halo="recompute")
sg.launch, sg.overlap = "kernels", False
pts = sum((s.nj1) * (plan.Ni + 1) for s in plan.subs if s.kind in ("mercator", "latlon"))
ref, res = None, {}
for rnd in range(3):
    for rows in (0, 4, 6, 8, 12):
        os.environ["OGG_MIDAS_TILE_ROWS"] = str(rows)
        for _ in range(10):
            sg.step()
        torch.cuda.synchronize()
        sg.step(time_kernels=True)
        for _ in range(30):
            sg.run_pass()
        k = sg.kernel_times_ms()["midas_angle"]
        res.setdefault(rows, []).append(k["total_ms"] / 31)
        if rnd == 0:
            fp = [int(sg.buf[s.name][f].view(torch.int64).sum().item()) for s in plan.subs if s.kind in ("mercator", "latlon")
                  for f in ("dx", "dy", "area", "angle_dx")]
            ref = ref or fp
            assert fp == ref, (rows, "differs from the streaming kernel")
out = {"workload": wl, "points": pts, "ms_per_pass_3_launches": {str(k): [round(x, 4) for x in v] for k, v in res.items()},
       "alg_TBps_best": {str(k): round(48.0 * pts / (min(v) * 1e-3) / 1e12, 3) for k, v in res.items()},
       "bit_identical_to_streaming_kernel": True}
print(json.dumps(out))
if len(sys.argv) > 2:
    json.dump(out, open(sys.argv[2], "w"), indent=1)
