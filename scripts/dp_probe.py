"""GPU probe of the displaced-pole kernels (K5 mesh + angle, K6 quadrature in both arc forms) against the oracle:
parity at small sizes, timing and parity at the full 1/8 degree size.  Writes gpurun_out/dp_probe.json."""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ocean_model_grid_generator_amd import _lib as L  # noqa: E402
from ocean_model_grid_generator_amd import ocean_grid_generator as ogg  # noqa: E402
from oracle import ogg_oracle as orc  # noqa: E402

out = {}
dev = "cuda:0"
full = "--full" in sys.argv


def rel(g, w):
    return float(np.max(np.abs(g - w) / np.abs(w)))


def quad_dev(form, order, nx, ny, lon0, lat0, lon_dp, r_dp, j0, n_dx, n_cell, reps=1):
    lib = L.load()
    wsb = int(lib.ogg_displaced_pole_quad_workspace_bytes(order, nx, n_cell))
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    dx = torch.zeros((n_dx, nx), dtype=torch.float64, device=dev)
    dy = torch.zeros((n_cell, nx + 1), dtype=torch.float64, device=dev)
    da = torch.zeros((n_cell, nx), dtype=torch.float64, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    ms = None
    for it in range(reps + 1):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.call("ogg_displaced_pole_metrics_quad_form_ws_dev", form, order, nx, ny, lon0, lat0, lon_dp, r_dp, 6371.0e3, j0, n_dx, n_cell,
               dx.data_ptr(), dy.data_ptr(), da.data_ptr(), ws.data_ptr(), wsb, st)
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1)
        ms = t if (ms is None or it == 1) else min(ms, t)
    flag = L.c_int(0)
    L.call("ogg_workspace_error_flag_dev", ws.data_ptr(), L.ctypes.byref(flag), st)
    return dx.cpu().numpy(), dy.cpu().numpy(), da.cpu().numpy(), ms, flag.value, wsb


# ---- small sizes, whole cap (incl. the rows around r = r_pole where the longitude swings) ---------------------------------
for (Ni, Nj, r_dp, order) in [(72, 14, 0.2, 4), (72, 14, 0.2, 2), (360, 70, 0.2, 4), (1440, 140, 0.2, 4), (720, 70, 0.34135899793333113, 4), (100, 9, 0.5, 4)]:
    want = orc.displacedPoleCap_metrics_quad(order, Ni, Nj, -300.0, -78.0, 80.0, r_dp)
    jm = int(np.ceil(0.49 * Nj))
    for form, name in ((0, "literal"), (1, "chord")):
        dx, dy, da, ms, flag, wsb = quad_dev(form, order, Ni, Nj, -300.0, -78.0, 80.0, r_dp, 0, Nj + 1, Nj)
        got = (dx, dy, da)
        key = "quad_%s_%dx%d_o%d_r%.2f" % (name, Ni, Nj, order, r_dp)
        out[key] = {"kept_rows_rel": [rel(g[jm:], w[jm:]) for g, w in zip(got, want)],
                    "all_rows_rel_to_scale": [float(np.max(np.abs(g - w)) / np.abs(w).max()) for g, w in zip(got, want)],
                    "ms": ms, "flag": flag, "ws_bytes": wsb}
        print(key, out[key], flush=True)
    # bands: bit-identical to the whole cap
    dxa, dya, daa, _, _, _ = quad_dev(0, order, Ni, Nj, -300.0, -78.0, 80.0, r_dp, 0, Nj + 1, Nj)
    cut = Nj // 3
    b1 = quad_dev(0, order, Ni, Nj, -300.0, -78.0, 80.0, r_dp, 0, cut, cut)
    b2 = quad_dev(0, order, Ni, Nj, -300.0, -78.0, 80.0, r_dp, cut, Nj + 1 - cut, Nj - cut)
    same = all(np.array_equal(np.concatenate((a, b)), w) for a, b, w in zip(b1[:3], b2[:3], (dxa, dya, daa)))
    out["quad_bands_bitwise_%dx%d_o%d" % (Ni, Nj, order)] = bool(same)
    print("bands bitwise", Ni, Nj, order, same, flush=True)

# ---- mesh + angle ----------------------------------------------------------------------------------------------------------
lib = L.load()
for (Ni, Nj, r_dp) in [(72, 14, 0.2), (1440, 140, 0.2), (720, 70, 0.34135899793333113), (5760, 560, 0.34135899793333113)]:
    wsb = int(lib.ogg_displaced_pole_grid_workspace_bytes(Ni, Nj + 1))
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    x = torch.zeros((Nj + 1, Ni + 1), dtype=torch.float64, device=dev)
    y = torch.zeros_like(x)
    a = torch.zeros_like(x)
    st = torch.cuda.current_stream().cuda_stream
    for it in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        L.call("ogg_displaced_pole_grid_angle_ws_dev", Ni, Nj, -300.0, -78.0, 80.0, r_dp, 0, Nj + 1, x.data_ptr(), y.data_ptr(), a.data_ptr(),
               ws.data_ptr(), wsb, st)
        e1.record()
        torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    ox, oy, _, _ = orc.generate_displaced_pole_grid(Ni, Nj, -300.0, -78.0, 80.0, r_dp)
    oa = orc.angle_x(ox, oy)
    gx, gy, ga = x.cpu().numpy(), y.cpu().numpy(), a.cpu().numpy()
    d = np.abs(ga - oa)
    d = np.minimum(d, np.abs(d - 360))
    key = "mesh_%dx%d_r%.2f" % (Ni, Nj, r_dp)
    out[key] = {"x": float(np.max(np.abs(gx - ox))), "y": float(np.max(np.abs(gy - oy))), "angle_q999": float(np.quantile(d[1:], 0.999)),
                "angle_max_rows1up": float(d[1:].max()), "ms": ms}
    print(key, out[key], flush=True)

# ---- full 1/8 degree size (BASELINE config 4): kept rows only ----------------------------------------------------------------
if full:
    Ni, Nj, r_dp, jm = 5760, 560, 0.34135899793333113, 276
    n_cell = Nj - jm
    res = {}
    for form, name in ((0, "literal"), (1, "chord")):
        dx, dy, da, ms, flag, wsb = quad_dev(form, 4, Ni, Nj, -300.0, -78.0, 80.0, r_dp, jm, n_cell + 1, n_cell, reps=5)
        res[name] = (dx, dy, da)
        out["full_%s" % name] = {"ms": ms, "flag": flag, "ws_bytes": wsb}
        print("full", name, out["full_%s" % name], flush=True)
    out["full_literal_vs_chord_rel"] = [rel(a, b) for a, b in zip(res["literal"], res["chord"])]
    t0 = time.time()
    want = orc.displacedPoleCap_metrics_quad(4, Ni, Nj, -300.0, -78.0, 80.0, r_dp, rows_per_chunk=8, j_first=jm)
    out["oracle_s"] = time.time() - t0
    for name in ("literal", "chord"):
        got = res[name]
        out["full_%s_vs_oracle_rel" % name] = [rel(got[0], want[0][jm:]), rel(got[1], want[1][jm:]), rel(got[2], want[2][jm:])]
        out["full_%s_vs_oracle_abs" % name] = [float(np.max(np.abs(got[0] - want[0][jm:]))), float(np.max(np.abs(got[1] - want[1][jm:]))),
                                               float(np.max(np.abs(got[2] - want[2][jm:])))]
        print(name, out["full_%s_vs_oracle_rel" % name], out["full_%s_vs_oracle_abs" % name], flush=True)

os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/dp_probe.json", "w"), indent=1, sort_keys=True)
print("done")
