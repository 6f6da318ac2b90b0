#!/usr/bin/env python3
"""Generate the golden fixtures in this directory FROM THE REFERENCE ITSELF.

Run in the build container only (needs /root/reference; never on the GPU box):

    python tests/golden/make_golden.py [--big]

The reference script imports ``numpypi.numpypi_series`` (OGG:5), which is not installed and cannot be fetched.
This script therefore puts a three-line *scratch* module on sys.path (under a temp dir, outside the repo) that
re-exports plain numpy under that name, imports the UNMODIFIED reference file from /root/reference, stubs its
NetCDF writer to capture arrays, and records:

  * ``ref_small_*.npz``      full output arrays of small CLI configurations (inputs = the flags);
  * ``ref_functions.npz``    per-function input/output vectors for every hot-path function;
  * ``ref_hashes.json``      sha256 of every output field for the larger configurations, shapes, scalars,
                             and a fingerprint of this platform's numpy transcendental results.

Everything recorded is data (inputs and outputs).  Numbers are numpy/glibc arithmetic, NOT numpypi arithmetic.
While generating, the oracle (oracle/ogg_oracle.py) is checked bit-for-bit against the same runs.
"""
import argparse
import contextlib
import hashlib
import io
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import ogg_oracle as orc  # noqa: E402

FIELDS = ("x", "y", "dx", "dy", "area", "angle_dx")

ARGPARSE_DEFAULTS = dict(gridfilename="G", r_dp=0.0, exfracdp=0.49, lon_dp=80.0, lat_dp=-99.0, south_cutoff_ang=-90.0,
                         south_cutoff_row=0, bipolar_lower_lat=-90.0, mercator_lower_lat=-90.0,
                         mercator_upper_lat=-99.0, south_ocean_lower_lat=-90.0, south_ocean_upper_lat=-99.0,
                         no_south_cap=False, match_dy=[], ensure_nj_even=False, plotem=False, skip_metrics=False,
                         write_subgrid_files=False, no_changing_meta=True, enhanced_equatorial=0,
                         shift_equator_to_u_point=True, grids="all")

# name -> flags (differences from the argparse defaults).  The first six are the reference's own test
# configurations (t/test_ocean_grid_gen.py:25-173); the rest cover BASELINE.json's configs and edge cases.
CONFIGS = {
    "r0.25_even": dict(inverse_resolution=0.25, ensure_nj_even=True),
    "r1_cut2": dict(inverse_resolution=1.0, south_cutoff_row=2),
    "r2": dict(inverse_resolution=2.0),
    "r2_equenh4": dict(inverse_resolution=2.0, enhanced_equatorial=4),
    "r4_om4": dict(inverse_resolution=4.0, r_dp=0.2, south_cutoff_row=83),
    "r4_om5proto": dict(inverse_resolution=4.0, south_ocean_lower_lat=-88.57, match_dy=["so"], no_south_cap=True),
    "r2_skip_metrics": dict(inverse_resolution=2.0, skip_metrics=True),                       # BASELINE config 0
    "r0.5_dp": dict(inverse_resolution=0.5, r_dp=0.2, ensure_nj_even=True),
    "r0.5_latdp": dict(inverse_resolution=0.5, lon_dp=80.0, lat_dp=-85.85, ensure_nj_even=True),
    "r1_dp_cutang": dict(inverse_resolution=1.0, r_dp=0.2, south_cutoff_ang=-81.0, ensure_nj_even=True),
    "r1_matchdy": dict(inverse_resolution=1.0, r_dp=0.2, south_cutoff_row=5, match_dy=["bp", "so", "p125sc"],
                       ensure_nj_even=True),
}
BIG = {
    "r8": dict(inverse_resolution=8.0),                                                       # BASELINE config 2
    "r8_latdp": dict(inverse_resolution=8.0, lon_dp=80.0, lat_dp=-85.85),                     # BASELINE config 3
    # the reference's own 1/8 degree test configuration (t/test_ocean_grid_gen.py:176-185, extras/Makefile:40-41): 4481 x 5761
    "r8_p125": dict(inverse_resolution=8.0, r_dp=0.2, south_cutoff_row=5, match_dy=["bp", "so", "p125sc"], ensure_nj_even=True),
}
SAVE_FULL = ("r0.25_even", "r0.5_dp")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype=np.float64).tobytes()).hexdigest()


def platform_fingerprint():
    """sha256 of a fixed battery of numpy transcendental results: tells a test whether bit-for-bit
    comparison with these fixtures is meaningful on the machine it runs on."""
    t = np.linspace(-3.0, 3.0, 4097)
    parts = [np.sin(t), np.cos(t), np.tan(t), np.arctan(t), np.arcsin(t / 3.0), np.arccos(t / 3.0),
             np.arctan2(t, t[::-1] + 0.1), np.sinh(t), np.log(t * t + 0.5), np.hypot(t, 1.0 - t), np.mod(t * 200, 360.0)]
    return hashlib.sha256(b"".join(p.tobytes() for p in parts)).hexdigest()


def load_reference():
    shim = tempfile.mkdtemp(prefix="oggshim_")
    os.makedirs(os.path.join(shim, "numpypi"))
    open(os.path.join(shim, "numpypi", "__init__.py"), "w").close()
    with open(os.path.join(shim, "numpypi", "numpypi_series.py"), "w") as f:
        f.write("from numpy import *\nimport numpy as _n\nfloat64=_n.float64\n")
    sys.dont_write_bytecode = True
    sys.path[:0] = [shim, "/root/reference"]
    import ocean_grid_generator as ogg
    return ogg


def run_reference(ogg, flags):
    cap = {}
    ogg.write_nc = lambda x, y, dx, dy, area, angle_dx, fnam=None, **k: cap.__setitem__(str(fnam), (x, y, dx, dy, area, angle_dx))
    kw = dict(ARGPARSE_DEFAULTS)
    kw.update(flags)
    with contextlib.redirect_stdout(io.StringIO()):
        ogg.main(**kw)
    return dict(zip(FIELDS, cap["G"]))


def oracle_flags(flags):
    kw = {k: v for k, v in flags.items()}
    r = kw.pop("inverse_resolution")
    return r, kw


def function_vectors(ogg):
    """Per-function golden vectors, small enough to commit."""
    rng = np.random.default_rng(20240807)
    out = {}
    # mdist (OGG:682)
    a = rng.uniform(-400, 400, 257)
    b = rng.uniform(-400, 400, 257)
    a[:4] = [0.0, 360.0, -300.0, 60.0]
    b[:4] = [0.0, 0.0, 60.0, -300.0]
    out["mdist_a"], out["mdist_b"], out["mdist_out"] = a, b, ogg.mdist(a, b)
    # Mercator scalars/axis (OGG:292-311)
    for Ni in (180, 1440, 5760):
        phi = np.array([-66.85954725 * ogg.PI_180, 64.05895973 * ogg.PI_180])
        out["ymr_%d" % Ni] = ogg.y_mercator_rounded(Ni, phi).astype(np.int64)
        ys = out["ymr_%d" % Ni]
        out["phiM_%d" % Ni] = ogg.phi_mercator(Ni, np.arange(ys[0], ys[1] + 1))
    # bipolar projection on an irregular mesh incl. the special meridians and the pole row (OGG:33-100)
    lon_bp, lat0 = -300.0, 64.05895973
    rp = np.tan(0.5 * (90 - lat0) * ogg.PI_180)
    lamg = np.tile(lon_bp + np.array([0, 1, 44.5, 89.999, 90, 90.001, 135, 180, 180.5, 269, 270, 271, 359, 360, 361.0]), (7, 1))
    phig = np.tile(np.array([lat0, 65, 70.3, 80, 89, 89.999, 90.0]).reshape(7, 1), (1, lamg.shape[1]))
    o = ogg.bipolar_projection(lamg.copy(), phig.copy(), lon_bp, rp)
    out.update(bp_lamg=lamg, bp_phig=phig, bp_lon_bp=lon_bp, bp_rp=rp, bp_lams=o[0], bp_phis=o[1], bp_hi=o[2], bp_hj=o[3])
    o = ogg.bipolar_projection(lamg.copy(), phig[:-1].copy() if False else phig.copy(), lon_bp, rp, metrics_only=True)
    out.update(bp_hi_mo=o[0], bp_hj_mo=o[1])
    # cap mesh + quadrature metrics at a tiny size (OGG:103-188)
    Ni, Nj = 48, 10
    lams, phis, hi, hj = ogg.generate_bipolar_cap_mesh(Ni, Nj, lat0, lon_bp, ensure_nj_even=False)
    out.update(bpm_Ni=Ni, bpm_Nj=Nj, bpm_lat0=lat0, bpm_lams=lams, bpm_phis=phis, bpm_hi=hi, bpm_hj=hj)
    for order in (2, 3, 4, 5):
        dxq, dyq, daq = ogg.bipolar_cap_metrics_quad_fast(order, Ni, Nj, lat0, lon_bp, rp)
        out.update({"bpq%d_dx" % order: dxq, "bpq%d_dy" % order: dyq, "bpq%d_da" % order: daq})
    # displaced pole mesh at integer and fractional indices (OGG:447-518)
    Ni, Nj, lon0, lat0sc, lon_dp, r_dp = 72, 14, -300.0, -78.0, 80.0, 0.2
    x, y, londp, latdp = ogg.generate_displaced_pole_grid(Ni, Nj, lon0, lat0sc, lon_dp, r_dp)
    out.update(dp_Ni=Ni, dp_Nj=Nj, dp_lon0=lon0, dp_lat0=lat0sc, dp_lon_dp=lon_dp, dp_r_dp=r_dp, dp_x=x, dp_y=y,
               dp_pole=np.array([londp, latdp]))
    fi = np.sort(rng.uniform(0, Ni + 1, 40))
    fj = np.sort(rng.uniform(-0.002, Nj + 1, 9))
    lam, phi, _, _ = ogg.displacedPoleCap_mesh(fi, fj, Ni, Nj, lon0, lat0sc, lon_dp, r_dp)
    out.update(dpf_i=fi, dpf_j=fj, dpf_lam=lam, dpf_phi=phi)
    out["dp_gad"] = ogg.great_arc_distance(fj, fi + 1e-3, fj, fi - 1e-3, Ni, Nj, lon0, lat0sc, lon_dp, r_dp)
    for order in (2, 4):
        dxq, dyq, daq = ogg.displacedPoleCap_metrics_quad(order, Ni, Nj, lon0, lat0sc, lon_dp, r_dp)
        out.update({"dpq%d_dx" % order: dxq, "dpq%d_dy" % order: dyq, "dpq%d_da" % order: daq})
    out["dp_hi6"] = ogg.numerical_hi(fj, fi, Ni, Nj, lon0, lat0sc, lon_dp, r_dp, eps=1e-3, order=6)
    out["dp_hj6"] = ogg.numerical_hj(fj, fi, Ni, Nj, lon0, lat0sc, lon_dp, r_dp, eps=1e-3, order=6)
    # MIDAS metrics + angle on a distorted (non lat-lon) mesh so that every stencil term matters (OGG:687-729)
    xx = np.tile(-300 + np.arange(33) * 360.0 / 32, (17, 1)) + rng.normal(0, 0.5, (17, 33))
    yy = np.tile((-80 + np.arange(17) * 10.0).reshape(17, 1), (1, 33)) + rng.normal(0, 0.5, (17, 33))
    dx, dy, ar = ogg.generate_grid_metrics_MIDAS(xx, yy)
    dx2, dy2, ar2 = ogg.generate_grid_metrics_MIDAS(xx, yy, latlon_areafix=False)
    out.update(md_x=xx, md_y=yy, md_dx=dx, md_dy=dy, md_area=ar, md_area_nofix=ar2, md_angle=ogg.angle_x(xx, yy))
    # lat-lon builder (OGG:832-846)
    lx, ly = ogg.generate_latlon_grid(24, 5, -300.0, 360, -78.0, 11.3, ensure_nj_even=True)
    out.update(ll_x=lx, ll_y=ly)
    # metrics_error self check (OGG:732-770)
    out["md_err"] = np.array(ogg.metrics_error(dx, dy, ar, 32, yy[0, 0], yy[-1, 0]))
    return out


def check_oracle_functions(v):
    """Bit-for-bit check of the oracle against the per-function vectors just generated."""
    eq = np.array_equal
    assert eq(orc.mdist(v["mdist_a"], v["mdist_b"]), v["mdist_out"])
    for Ni in (180, 1440, 5760):
        phi = np.array([-66.85954725 * orc.PI_180, 64.05895973 * orc.PI_180])
        ys = orc.y_mercator_rounded(Ni, phi)
        assert eq(ys, v["ymr_%d" % Ni])
        assert eq(orc.phi_mercator(Ni, np.arange(ys[0], ys[1] + 1)), v["phiM_%d" % Ni])
    o = orc.bipolar_projection(v["bp_lamg"], v["bp_phig"], v["bp_lon_bp"], v["bp_rp"])
    for a, k in zip(o, ("bp_lams", "bp_phis", "bp_hi", "bp_hj")):
        assert eq(a, v[k], equal_nan=True), k
    o = orc.generate_bipolar_cap_mesh(int(v["bpm_Ni"]), int(v["bpm_Nj"]), float(v["bpm_lat0"]), v["bp_lon_bp"], False)
    for a, k in zip(o, ("bpm_lams", "bpm_phis", "bpm_hi", "bpm_hj")):
        assert eq(a, v[k]), k
    for order in (2, 3, 4, 5):
        o = orc.bipolar_cap_metrics_quad_fast(order, int(v["bpm_Ni"]), int(v["bpm_Nj"]), float(v["bpm_lat0"]),
                                              v["bp_lon_bp"], v["bp_rp"], rows_per_chunk=3)
        for a, k in zip(o, ("dx", "dy", "da")):
            assert eq(a, v["bpq%d_%s" % (order, k)]), (order, k)
    dp = [int(v["dp_Ni"]), int(v["dp_Nj"]), float(v["dp_lon0"]), float(v["dp_lat0"]), float(v["dp_lon_dp"]), float(v["dp_r_dp"])]
    o = orc.generate_displaced_pole_grid(*dp)
    assert eq(o[0], v["dp_x"]) and eq(o[1], v["dp_y"])
    o = orc.displacedPoleCap_mesh(v["dpf_i"], v["dpf_j"], *dp)
    assert eq(o[0], v["dpf_lam"]) and eq(o[1], v["dpf_phi"])
    assert eq(orc.great_arc_distance(v["dpf_j"], v["dpf_i"] + 1e-3, v["dpf_j"], v["dpf_i"] - 1e-3, *dp), v["dp_gad"])
    for order in (2, 4):
        o = orc.displacedPoleCap_metrics_quad(order, *dp, rows_per_chunk=4)
        for a, k in zip(o, ("dx", "dy", "da")):
            assert eq(a, v["dpq%d_%s" % (order, k)]), (order, k)
    assert eq(orc.numerical_hi(v["dpf_j"], v["dpf_i"], *dp, eps=1e-3, order=6), v["dp_hi6"])
    assert eq(orc.numerical_hj(v["dpf_j"], v["dpf_i"], *dp, eps=1e-3, order=6), v["dp_hj6"])
    o = orc.generate_grid_metrics_MIDAS(v["md_x"], v["md_y"])
    assert eq(o[0], v["md_dx"]) and eq(o[1], v["md_dy"]) and eq(o[2], v["md_area"])
    assert eq(orc.generate_grid_metrics_MIDAS(v["md_x"], v["md_y"], latlon_areafix=False)[2], v["md_area_nofix"])
    assert eq(orc.angle_x(v["md_x"], v["md_y"]), v["md_angle"])
    o = orc.generate_latlon_grid(24, 5, -300.0, 360, -78.0, 11.3, ensure_nj_even=True)
    assert eq(o[0], v["ll_x"]) and eq(o[1], v["ll_y"])
    assert eq(np.array(orc.metrics_error(v["md_dx"], v["md_dy"], v["md_area"], 32, v["md_y"][0, 0], v["md_y"][-1, 0])), v["md_err"])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--big", action="store_true", help="also hash the 1/8 degree configurations (minutes of CPU)")
    ap.add_argument("--only", nargs="*", default=None, help="run just these configurations (names of CONFIGS / BIG); the others keep their "
                    "recorded hashes (same platform fingerprint required)")
    args = ap.parse_args()
    ogg = load_reference()

    with contextlib.redirect_stdout(io.StringIO()):
        vec = function_vectors(ogg)
    check_oracle_functions(vec)
    np.savez_compressed(os.path.join(HERE, "ref_functions.npz"), **vec)
    print("ref_functions.npz: %d arrays, oracle bit-identical" % len(vec))

    hashes_path = os.path.join(HERE, "ref_hashes.json")
    record = {"platform_fingerprint": platform_fingerprint(), "numpy": np.__version__, "configs": {}}
    if os.path.exists(hashes_path):
        old = json.load(open(hashes_path))
        if old.get("platform_fingerprint") == record["platform_fingerprint"]:
            record["configs"].update({k: v for k, v in old["configs"].items() if k in BIG or args.only is not None})
        elif args.only is not None:
            raise SystemExit("--only needs the recorded platform fingerprint to match this machine's")
    todo = dict(CONFIGS)
    if args.big:
        todo.update(BIG)
    if args.only is not None:
        todo = {k: v for k, v in {**CONFIGS, **BIG}.items() if k in args.only}
    for name, flags in todo.items():
        ref = run_reference(ogg, flags)
        r, kw = oracle_flags(flags)
        mine = orc.make_supergrid(r, **kw)
        for f in FIELDS:
            assert ref[f].shape == mine[f].shape, (name, f, ref[f].shape, mine[f].shape)
            assert np.array_equal(ref[f], mine[f]), (name, f, np.abs(ref[f] - mine[f]).max())
        record["configs"][name] = {"flags": flags, "shapes": {f: list(ref[f].shape) for f in FIELDS},
                                   "sha256": {f: sha(ref[f]) for f in FIELDS},
                                   "sub_rows": {k: int(p[0].shape[0]) for k, p in mine["sub"].items()}}
        if name in SAVE_FULL:
            np.savez_compressed(os.path.join(HERE, "ref_small_%s.npz" % name), **ref)
        print("%-16s %s oracle bit-identical" % (name, tuple(ref["x"].shape)))
    json.dump(record, open(hashes_path, "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
