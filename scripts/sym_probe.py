"""Mirrored columns of the caps (OGG_SYM_MIRROR) against every-column evaluation (OGG_SYM_NONE) and against the oracle, on one GPU:
    python scripts/sym_probe.py [r8] [--rows-step 8] [--json out.json]
For the bipolar quadrature (OGG:136-188), the bipolar mesh + angle (OGG:103-122, 719-729) and the displaced-pole quadrature in the chord
form (OGG:565-601): max relative (metrics) / absolute (coordinates) difference of either evaluation from the oracle, of the two from each
other, where they differ, and how many values are bit-identical.  The oracle is evaluated on every `--rows-step`-th cell row."""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ocean_model_grid_generator_amd import _lib as L   # noqa: E402
from ocean_model_grid_generator_amd import ocean_grid_generator as ogg   # noqa: E402
from oracle import ogg_oracle as orc   # noqa: E402

CFG = {"r2": (1440, 238, 64.97316302279852), "r4": (2880, 480, 64.0589597296948), "r8": (5760, 960, 64.03160594077568),
       "r16": (11520, 1920, 64.04528618884338), "r1": (720, 120, 64.05895973), "r0.5": (360, 60, 64.05895973)}
ap = argparse.ArgumentParser()
ap.add_argument("size", nargs="?", default="r8")
ap.add_argument("--rows-step", type=int, default=8)
ap.add_argument("--what", default="quad,mesh,dpole")
ap.add_argument("--json", default=None)
args = ap.parse_args()
nx, ny, lat0 = CFG[args.size]
lon_bp = -300.0
rp = float(np.tan(0.5 * (90 - lat0) * np.pi / 180))
out = {"size": args.size, "nx": nx, "ny": ny, "lib": L.load().ogg_version().decode()}


def with_sym(flag, fn):
    os.environ["OGG_CAP_SYMMETRY"] = "1" if flag else "0"
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        return fn()


def rel(a, b):
    with np.errstate(divide="ignore", invalid="ignore"):
        r = np.abs(a - b) / np.abs(b)
    r[~np.isfinite(r)] = 0.0
    return r


if "quad" in args.what:
    A = with_sym(False, lambda: ogg.bipolar_cap_metrics_quad_fast(5, nx, ny, lat0, lon_bp, rp))
    B = with_sym(True, lambda: ogg.bipolar_cap_metrics_quad_fast(5, nx, ny, lat0, lon_bp, rp))
    rows = list(range(0, ny, args.rows_step)) + list(range(ny - 12, ny))
    rep = {}
    for name, a, b in zip(("dx", "dy", "area"), A, B):
        same = a == b
        rep[name] = {"bit_identical_frac": float(same.mean()), "max_rel_mirror_vs_none": float(rel(b, a).max()),
                     "rows_all_identical": int(same.all(axis=1).sum()), "rows": int(a.shape[0])}
    wa = {k: 0.0 for k in ("dx", "dy", "area")}
    wb = dict(wa)
    for j in rows:
        O = orc.bipolar_cap_metrics_quad_fast(5, nx, ny, lat0, lon_bp, rp, j_first=j, j_last=j + 1)
        for name, a, b, o in zip(("dx", "dy", "area"), A, B, O):
            wa[name] = max(wa[name], float(rel(a[j], o[j]).max()))
            wb[name] = max(wb[name], float(rel(b[j], o[j]).max()))
    for name in wa:
        rep[name]["none_vs_oracle"] = wa[name]
        rep[name]["mirror_vs_oracle"] = wb[name]
    rep["area_abs_m2_mirror_vs_none"] = float(np.abs(A[2] - B[2]).max())
    out["bipolar_quad"] = rep
    print(json.dumps({"bipolar_quad": rep}), flush=True)

if "mesh" in args.what:
    def mesh():
        x, y, _, _ = ogg.generate_bipolar_cap_mesh(nx, ny, lat0, lon_bp)
        return x, y, ogg.angle_x(x, y)
    # (the function-level mesh has no fused angle; the pass has: compared through supergrid below when asked for)
    A = with_sym(False, mesh)
    B = with_sym(True, mesh)
    xo, yo, _, _ = orc.generate_bipolar_cap_mesh(nx, ny, lat0, lon_bp)
    ao = orc.angle_x(xo, yo)
    rep = {}
    for name, a, b, o in zip(("x", "y", "angle"), A, B, (xo, yo, ao)):
        d_ab = np.abs(a - b)
        da, db = np.abs(a - o), np.abs(b - o)
        if name == "angle":
            d_ab, da, db = (np.minimum(d, np.abs(d - 360)) for d in (d_ab, da, db))
        rep[name] = {"bit_identical_frac": float((a == b).mean()), "max_abs_mirror_vs_none": float(d_ab.max()),
                     "none_vs_oracle": float(da.max()), "mirror_vs_oracle": float(db.max()),
                     "none_vs_oracle_p999": float(np.quantile(da, 0.999)), "mirror_vs_oracle_p999": float(np.quantile(db, 0.999))}
    out["bipolar_mesh"] = rep
    print(json.dumps({"bipolar_mesh": rep}), flush=True)

if "dpole" in args.what:
    dnx = nx
    dny = int(nx / 720 * 40) * 7 // 4
    lon0, dlat0, lon_dp = -300.0, -78.0, 80.0
    r_dp = float(np.tan((90 - 85.85) * np.pi / 180) / np.tan((90 + dlat0) * np.pi / 180)) if args.size != "r4" else 0.2
    jmin = int(np.ceil(0.49 * dny))
    jmin += jmin % 2
    A = with_sym(False, lambda: ogg.displacedPoleCap_metrics_quad(4, dnx, dny, lon0, dlat0, lon_dp, r_dp))
    B = with_sym(True, lambda: ogg.displacedPoleCap_metrics_quad(4, dnx, dny, lon0, dlat0, lon_dp, r_dp))
    rep = {}
    for name, a, b in zip(("dx", "dy", "area"), A, B):
        rep[name] = {"bit_identical_frac": float((a[jmin:] == b[jmin:]).mean()), "max_rel_mirror_vs_none": float(rel(b[jmin:], a[jmin:]).max())}
    wa = {k: 0.0 for k in ("dx", "dy", "area")}
    wb = dict(wa)
    for j in range(jmin, dny, max(1, 4 * args.rows_step)):
        O = orc.displacedPoleCap_metrics_quad(4, dnx, dny, lon0, dlat0, lon_dp, r_dp, j_first=j, j_last=j + 1)
        for name, a, b, o in zip(("dx", "dy", "area"), A, B, O):
            wa[name] = max(wa[name], float(rel(a[j], o[j]).max()))
            wb[name] = max(wb[name], float(rel(b[j], o[j]).max()))
    for name in wa:
        rep[name]["none_vs_oracle"] = wa[name]
        rep[name]["mirror_vs_oracle"] = wb[name]
    out["dpole_quad_chord"] = rep
    print(json.dumps({"dpole_quad_chord": rep}), flush=True)

if args.json:
    json.dump(out, open(args.json, "w"), indent=1)
