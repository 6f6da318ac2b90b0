"""The (field, row)-ordered lat-lon kernel (OGG_LATLON_ROWS=1) against the column-tile kernel, stand-alone, in one process on one box:
1/8, 1/16 and 1/2 degree."""
import os, sys
sys.path.insert(0, "/root/repo")
import torch
from ocean_model_grid_generator_amd import supergrid as SG
import bench
for wl in ("r8", "r16", "r2"):
    plan = SG.SupergridPlan(**bench.WORKLOADS[wl])
    res = {}
    for mode in ("0", "1"):
        os.environ["OGG_LATLON_ROWS"] = mode
        g = SG.Supergrid(plan, device="cuda:0")
        g.launch, g.overlap = "kernels", False
        for s in plan.subs:
            for f in SG.FIELDS:
                g.buf[s.name][f].fill_(float("nan"))
        g._events = None
        for _ in range(30):
            g.phase_a(kinds=("mercator", "latlon"))
        torch.cuda.synchronize()
        for wg in ((None,) if mode == "0" else (4096, 8192, 16384, 65536)):
            if wg:
                os.environ["OGG_ROWS_MAX_WG"] = str(wg)
            ts = []
            for rep in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    g.phase_a(kinds=("mercator", "latlon"))
                e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 20)
            pts = sum(g.buf[s.name]["n"] for s in plan.subs if s.kind in ("mercator", "latlon")) * (plan.Ni + 1)
            print(wl, "rows" if mode == "1" else "tiles", wg, "%.4f ms  %.0f GB/s" % (min(ts), 48 * pts / min(ts) / 1e6), flush=True)
        res[mode] = {s.name: {f: g.buf[s.name][f].clone() for f in SG.FIELDS} for s in plan.subs if s.kind in ("mercator", "latlon")}
        del g
    same = all(torch.equal(res["0"][n][f], res["1"][n][f]) for n in res["0"] for f in SG.FIELDS)
    print(wl, "bit-identical:", same, flush=True)
