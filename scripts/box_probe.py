"""What kind of box is this?  Two MI355X boxes of the pool differ by up to 25 % in what their WRITE path sustains while their clocks and
VALU rates agree, and every number of this repository that rides on the write path inherits that (the 1/16 degree pass: 0.91 or 1.18 ms).
One line: plain 1 GiB fill rate, the stand-alone lat-lon kernel at 1/16 degree in its two orderings (column tiles / (field, row)-ordered
workgroups), and the fused 1/16 degree pass.  usage: python3 scripts/box_probe.py [out.json]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from ocean_model_grid_generator_amd import _lib as L, supergrid as SG  # noqa: E402


def ev_ms(fn, reps):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


out = {"device": L.device_name()}
n = 1 << 27
a = torch.empty(n, dtype=torch.float64, device="cuda")
st = torch.cuda.current_stream().cuda_stream
for _ in range(5):
    a.fill_(1.5)
out["fill_1GiB_TBps"] = round(n * 8 / min(ev_ms(lambda: a.fill_(1.5), 20) for _ in range(3)) / 1e9, 3)
out["ogg_fill_1GiB_TBps"] = round(n * 8 / min(ev_ms(lambda: L.call("ogg_fill_dev", n, 1.5, a.data_ptr(), st), 20) for _ in range(3)) / 1e9, 3)
del a
plan = SG.SupergridPlan(**bench.WORKLOADS["r16"])
pts = None
for mode, name in (("0", "latlon_tiles"), ("1", "latlon_rows")):
    os.environ["OGG_LATLON_ROWS"] = mode
    g = SG.Supergrid(plan, device="cuda:0")
    g.launch, g.overlap, g._events = "kernels", False, None
    fn = lambda: g.phase_a(kinds=("mercator", "latlon"))
    for _ in range(30):
        fn()
    torch.cuda.synchronize()
    ms = min(ev_ms(fn, 20) for _ in range(3))
    pts = sum(g.buf[s.name]["n"] for s in plan.subs if s.kind in ("mercator", "latlon")) * (plan.Ni + 1)
    out[name + "_r16"] = {"ms": round(ms, 4), "TBps": round(48 * pts / ms / 1e9, 3)}
    if mode == "0":   # where the Mercator band's six arrays start: offset within a 2 MiB page (KiB) and 2 MiB page number mod 64
        out["merc_bases"] = {f: [g.buf["Merc"][f].data_ptr() % (1 << 21) // 1024, g.buf["Merc"][f].data_ptr() // (1 << 21) % 64] for f in SG.FIELDS}
    g.close()
    del g
os.environ["OGG_LATLON_ROWS"] = "0"
for wl in ("r16", "r8"):
    plan = SG.SupergridPlan(**bench.WORKLOADS[wl])
    for rep in range(2):
        g = SG.Supergrid(plan, device="cuda:0")
        g.launch = "pass"
        for _ in range(60):
            g.run_pass()
        torch.cuda.synchronize()
        ms = min(ev_ms(g.run_pass, 50) for _ in range(3))
        out.setdefault("fused_pass_%s" % wl, []).append(round(ms, 4))
        g.close()
        del g
print(json.dumps(out))
if len(sys.argv) > 1:
    with open(sys.argv[1], "a") as f:
        f.write(json.dumps(out) + "\n")
