#!/usr/bin/env python3
"""Extended-precision TRUTH for the reference's own formulas at their ill-conditioned loci.

Build container only, CPU only (mpmath, 50 digits):

    python scripts/truth_table.py [--procs 8] [--quick]      ->  tests/golden/truth_table.npz

What "truth" means here: the value of the reference's formula AS CODED -- the same operations, the same fp64 constants
(PI_180 = the double numpy.pi / 180, eps = the double 1e-3, Re = 6371e3, the fp64 Lobatto nodes) -- applied to the same
fp64 INPUTS, with every arithmetic operation and every transcendental exact (50 digits, rounded once at the end).  It is
NOT a better formula (not the analytic metric, not a finer finite difference): it is what an fp64 evaluation of the
reference's lines would return if fp64 arithmetic did not round.  |fp64 reference - truth| is therefore the reference's
own rounding error, and it is the yardstick the GPU kernels are held to (tests/test_gpu_truth.py): a kernel that is no
further from the truth than the fp64 reference is cannot be told apart from "the reference on another libm".

Three groups ("next round" item 1):

  dp   OGG:522-601  displacedPoleCap_metrics_quad(4): numerical_hi / numerical_hj (4th-order central differences of the
                    haversine distance of re-projected probes, eps = 1e-3) + 4 x 4 Lobatto means, on cells of the cap
                    of BASELINE config 4 (1/8 degree, --lat_dp -85.85 --lon_dp 80: nx = 5760, ny = 560): 20 cell rows
                    of the kept rows 276..559 x every 11th column (10 500 cells), plus the three rows around
                    r = r_pole (row 193.7, where the longitude swings by 180 degrees between two columns; main()
                    discards them, OGG:1177-1186) x every 44th column.  Two truths: `A` = everything exact from the
                    lattice nodes on; `B` = the probe positions i +- k eps, j +- k eps rounded to fp64 first, as the
                    reference forms them (OGG:538-541), everything after that exact.  A - B is the part of the
                    reference's error that comes from rounding the probe POSITIONS (ulp(5760) / 2e-3 = 4.5e-10), which
                    every fp64 implementation shares bit for bit; B isolates the transcendental part.
  md   OGG:695-713  generate_grid_metrics_MIDAS (dx, dy, area with latlon_areafix) on the Mercator sub-grid at
                    Ni = 5760 and 11520: every row, 9 sample columns.  Inputs: the fp64 axes (stored in the fixture).
  bp   OGG:41-70    bipolar_projection (lams, phis) on the 1/8 and 1/16 degree caps: the five columns around each
                    symmetry meridian, every row; and the pole row, every column.  Inputs: lamg, phig, lon_bp, rp as
                    generate_bipolar_cap_mesh (OGG:103-122) forms them in fp64.

  bq   OGG:125-188  bipolar_cap_metrics_quad_fast(5) over bipolar_projection(metrics_only) (OGG:33-100) on the 1/8 degree cap: 14 cell
                    rows (the joint, mid-cap, the ten rows under the pole incl. the j = ny - 0.001 row) x every 37th column and the ten
                    columns around each symmetry meridian (the cells that touch the two pole points).

Stored per value: hi = fp64(truth), lo = fp64(truth - hi)  (so that v - truth = (v - hi) - lo to 1e-32 relative), and
the oracle's distance from the truth as measured when the fixture was made (`*_eref`), which the tests use as the unit
of their tolerances.
"""
import argparse
import os
import sys
import time
from multiprocessing import Pool

import mpmath as mp
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
from oracle import ogg_oracle as orc  # noqa: E402

mp.mp.dps = 50
RE = 6371.0e3


def M(x):
    """fp64 -> mpf, exactly."""
    return mp.mpf(float(x))


PI180 = M(np.pi / 180.0)           # OGG:13: the DOUBLE, not pi / 180
DEG = M(180.0 / np.pi)             # numpy.angle(deg=True) multiplies by the double 180 / pi


def split(t):
    hi = float(t)
    return hi, float(t - M(hi))


# ---------------------------------------------------------------------------------------------------------------
# dp: OGG:447-467, 478-506, 522-601 in exact arithmetic
# ---------------------------------------------------------------------------------------------------------------
class Cap:
    def __init__(self, nx, ny, lon0, lat0, lon_dp, r_dp):
        self.nx, self.ny = nx, ny
        self.lon0, self.lat0 = M(lon0), M(lat0)
        self.rj = mp.tan((90 + self.lat0) * PI180)                      # OGG:494
        a = M(lon_dp) * PI180
        self.z0 = M(r_dp) * mp.mpc(mp.cos(a), mp.sin(a))                # OGG:495
        self.z0c = mp.conj(self.z0)

    def project(self, i, j):
        """(lam, phi) in RADIANS as great_arc_distance sees them (OGG:524-527).  The 360-degree unwrap (OGG:470-475) moves lam
        by a multiple of 360 degrees, which sin^2(dlam / 2) does not see in exact arithmetic: omitted."""
        lon = self.lon0 + i * 360 / self.nx                              # OGG:480
        lat = -90 + j * (self.lat0 + 90) / self.ny                       # OGG:482-483
        r = mp.tan((90 + lat) * PI180) / self.rj                         # OGG:448
        t = lon * PI180
        e = mp.mpc(mp.cos(t), mp.sin(t))                                 # OGG:451
        ep = (e - self.z0) / (1 - self.z0c * e)                          # OGG:452
        z = r * ep                                                       # OGG:454
        w = (z + self.z0) / (1 + self.z0c * z)                           # OGG:455
        lam = mp.arg(w) * DEG                                            # OGG:457
        phi = -90 + mp.atan(abs(w) * self.rj) / PI180                    # OGG:465-466
        return lam * PI180, phi * PI180

    def arc(self, i0, j0, i1, j1):
        lam0, phi0 = self.project(i0, j0)
        lam1, phi1 = self.project(i1, j1)
        dphi, dlam = phi1 - phi0, lam1 - lam0
        d = mp.sin(dphi / 2) ** 2 + mp.sin(dlam / 2) ** 2 * mp.cos(phi0) * mp.cos(phi1)   # OGG:531
        return 2 * mp.asin(mp.sqrt(d))


EPS64 = 1e-3


def probes(node, exact):
    """The four probe positions node +- eps, node +- 2 eps of OGG:538,541 / 553,556: exact, or rounded to fp64 as the reference forms them."""
    if exact:
        n, e = M(node), M(EPS64)
        return n + e, n - e, n + 2 * e, n - 2 * e
    return M(node + EPS64), M(node - EPS64), M(node + 2.0 * EPS64), M(node - 2.0 * EPS64)


def h_pair(cap, jn, inn, exact):
    """numerical_hi, numerical_hj (order 4) at lattice node (jn, inn) (fp64 values).  OGG:535-562."""
    reps = 1 / M(EPS64)
    ip1, im1, ip2, im2 = probes(inn, exact)
    jp1, jm1, jp2, jm2 = probes(jn, exact)
    jm, im = M(jn), M(inn)
    hi = (8 * cap.arc(ip1, jm, im1, jm) - cap.arc(ip2, jm, im2, jm)) * (M(1.0) / 12) * reps
    hj = (8 * cap.arc(im, jp1, im, jm1) - cap.arc(im, jp2, im, jm2)) * (M(1.0) / 12) * reps
    return hi, hj


def dp_cell(args):
    (nx, ny, lon0, lat0, lon_dp, r_dp), j, i, jn, inn = args
    cap = Cap(nx, ny, lon0, lat0, lon_dp, r_dp)
    w = [1, 5, 5, 1]
    out = []
    for exact in (True, False):
        hi = [[None] * 4 for _ in range(4)]
        hj = [[None] * 4 for _ in range(4)]
        for a in range(4):
            for b in range(4):
                hi[a][b], hj[a][b] = h_pair(cap, jn[a], inn[b], exact)
        d = M(1.0) / 12
        area = d * d * sum(w[a] * w[b] * hi[a][b] * hj[a][b] for a in range(4) for b in range(4))      # OGG:239-245
        dx = d * (5 * (hi[0][1] + hi[0][2]) + (hi[0][0] + hi[0][3]))                                   # OGG:217-218
        dy = d * (5 * (hj[1][0] + hj[2][0]) + (hj[0][0] + hj[3][0]))
        out += [split(dx * RE), split(dy * RE), split(area * RE * RE)]
    return j, i, out


def dp_group(pool, quick, tag="dp"):
    """tag "dp": the cap of BASELINE config 4 (1/8 degree, --lat_dp -85.85); "dp4": the cap of config 2 (OM4 1/4 degree, --r_dp 0.2), the
    rows main() keeps after --south_cutoff_row 83 (native rows 220..279 of the 280) x every 11th column."""
    if tag == "dp":
        nx, ny, lon0, lat0, lon_dp, lat_dp = 5760, 560, -300.0, -78.0, 80.0, -85.85
        r_dp = float(np.tan((90 + lat_dp) * orc.PI_180) / np.tan((90 + lat0) * orc.PI_180))           # OGG:1160 (main's fp64 value: an INPUT)
        jpole = (lat_dp + 90.0) / ((lat0 + 90.0) / ny)
        kept = [276, 277, 278, 290, 310, 330, 350, 370, 390, 410, 430, 450, 470, 490, 510, 530, 545, 557, 558, 559]
    else:
        nx, ny, lon0, lat0, lon_dp, r_dp = 2880, 280, -300.0, -78.0, 80.0, 0.2
        jpole = float(np.arctan(r_dp * np.tan(12.0 * orc.PI_180)) / orc.PI_180 / (12.0 / ny))          # the row where r = r_pole
        kept = [220, 221, 230, 240, 250, 260, 270, 277, 278, 279]
    polar = [int(jpole) - 1, int(jpole), int(jpole) + 1]
    cols_k = list(range(0, nx, 11))
    cols_p = list(range(0, nx, 44))
    if quick:
        kept, polar, cols_k, cols_p = kept[::7], polar[1:2], cols_k[::40], cols_p[::40]
    j1d = orc._lattice_1d(ny, 4).reshape(ny + 1, 4)
    i1d = orc._lattice_1d(nx, 4).reshape(nx + 1, 4)
    par = (nx, ny, lon0, lat0, lon_dp, r_dp)
    jobs = [(par, j, i, j1d[j], i1d[i]) for j in kept for i in cols_k] + [(par, j, i, j1d[j], i1d[i]) for j in polar for i in cols_p]
    t0 = time.time()
    res = pool.map(dp_cell, jobs, chunksize=16)
    print("%s: %d cells in %.0f s" % (tag, len(jobs), time.time() - t0), flush=True)
    jj = np.array([r[0] for r in res])
    ii = np.array([r[1] for r in res])
    vals = np.array([r[2] for r in res])                      # (n, 6, 2): A dx dy area, B dx dy area; hi / lo
    out = {tag + "_params": np.array([nx, ny, lon0, lat0, lon_dp, r_dp, 4.0]), tag + "_j": jj, tag + "_i": ii, tag + "_kept": np.isin(jj, kept),
           tag + "_jpole": np.array(jpole)}
    for k, name in enumerate(("A_dx", "A_dy", "A_area", "B_dx", "B_dy", "B_area")):
        out[tag + "_" + name] = vals[:, k, :].copy()
    # the oracle's own distance from the truth, row by row (whole lattice rows: the unwrap scan runs along i)
    eref = {n: 0.0 for n in ("A_dx", "A_dy", "A_area", "B_dx", "B_dy", "B_area")}
    for j in sorted(set(jj.tolist())):
        o = orc.displacedPoleCap_metrics_quad(4, nx, ny, lon0, lat0, lon_dp, r_dp, j_first=j, j_last=j + 1)
        m = (jj == j) & out[tag + "_kept"]
        if not m.any():
            continue
        for f, k in (("dx", 0), ("dy", 1), ("area", 2)):
            v = o[k][j, ii[m]]
            for T in "AB":
                t = out["%s_%s_%s" % (tag, T, f)][m]
                eref["%s_%s" % (T, f)] = max(eref["%s_%s" % (T, f)], float(np.max(np.abs((v - t[:, 0]) - t[:, 1]) / np.abs(t[:, 0]))))
    for n, v in eref.items():
        out["%s_%s_eref" % (tag, n)] = np.array(v)
        print("%s oracle vs truth %s: rel %.3e" % (tag, n, v))
    return out


# ---------------------------------------------------------------------------------------------------------------
# md: OGG:682-716 in exact arithmetic
# ---------------------------------------------------------------------------------------------------------------
def mdist_exact(a, b):
    def pymod(v):                      # numpy.mod: result has the sign of the divisor
        return v - 360 * mp.floor(v / 360)
    return min(pymod(a - b), pymod(b - a))


def md_rows(args):
    xcols, yrows = args                # xcols: (ncol, 2) fp64 x[i], x[i+1]; yrows: fp64 y[j], y[j+1], y[j+2]-or-nan
    y0, y1 = M(yrows[0]), M(yrows[1])
    out = []
    for xa, xb in xcols:
        dxi = mdist_exact(M(xb), M(xa)) * PI180                      # same on both rows of a lat-lon mesh
        lv0 = (M(0.5) * (y0 + y0)) * PI180                           # OGG:695 (y does not depend on i)
        lv1 = (M(0.5) * (y1 + y1)) * PI180
        dx = RE * mp.sqrt(((y0 - y0) * PI180) ** 2 + (dxi * mp.cos(lv0)) ** 2)          # OGG:696-698
        lu = (M(0.5) * (y1 + y0)) * PI180
        dxj = mdist_exact(M(xa), M(xa)) * PI180
        dy = RE * mp.sqrt(((y1 - y0) * PI180) ** 2 + (dxj * mp.cos(lu)) ** 2)           # OGG:699-702
        area = (M(RE) ** 2) * ((M(0.5) * (dxi + dxi)) * (mp.sin(lv1) - mp.sin(lv0)))    # OGG:711-713
        out.append([split(dx), split(dy), split(area)])
    return out


def md_group(pool, quick):
    out = {}
    for Ni, refine in ((5760, 8.0), (11520, 16.0)):
        x, y = orc.generate_mercator_grid(Ni, -66.85954725, 64.05895973, -300.0, 360, refine, True, False)
        xa, ya = x[0].copy(), y[:, 0].copy()
        cols = np.array([0, 1, Ni // 4 - 1, Ni // 4, Ni // 2 - 1, Ni // 2, 3 * Ni // 4 + 1, Ni - 2, Ni - 1])
        rows = np.arange(0, ya.size - 1, 40 if quick else 1)
        jobs = [(np.stack([xa[cols], xa[cols + 1]], 1), ya[j:j + 2]) for j in rows]
        t0 = time.time()
        res = np.array(pool.map(md_rows, jobs, chunksize=32))       # (nrow, ncol, 3, 2)
        print("md %d: %d rows in %.0f s" % (Ni, len(jobs), time.time() - t0), flush=True)
        tag = "md%d_" % Ni
        out.update({tag + "xaxis": xa, tag + "yaxis": ya, tag + "cols": cols, tag + "rows": rows})
        odx, ody, oar = orc.generate_grid_metrics_MIDAS(np.tile(xa, (ya.size, 1)), np.tile(ya[:, None], (1, xa.size)))
        for k, (f, o) in enumerate((("dx", odx), ("dy", ody), ("area", oar))):
            t = res[:, :, k, :]
            out[tag + f] = t.copy()
            v = o[rows][:, cols]
            err = np.abs((v - t[..., 0]) - t[..., 1])
            out[tag + f + "_eref_abs"] = np.array(err.max())
            out[tag + f + "_eref_rel"] = np.array((err / np.abs(t[..., 0])).max())
            print("md %d oracle vs truth %s: abs %.3e rel %.3e" % (Ni, f, err.max(), (err / np.abs(t[..., 0])).max()))
    return out


# ---------------------------------------------------------------------------------------------------------------
# bp: OGG:41-70 in exact arithmetic
# ---------------------------------------------------------------------------------------------------------------
HUGE = M(1.0e30)


def bp_point(args):
    lamg, phig, lon_bp, rp = (M(v) for v in args)
    phig = 90 - 2 * mp.atan(mp.tan(M(0.5) * (90 - phig) * PI180) / rp) / PI180        # OGG:41
    tmp = mdist_exact(lamg, lon_bp) * PI180                                              # OGG:42
    sinla = mp.sin(tmp)
    sphig = mp.sin(phig * PI180)
    alpha2 = mp.cos(tmp) ** 2
    t = phig * PI180
    c = mp.cos(t)
    guard = (c == 0) or (mp.tan(t) ** 2 > HUGE)                                          # OGG:46, 52 (tan = inf counts as > HUGE)
    if guard:
        B = M(0.0)
    else:
        beta2_inv = mp.tan(t) ** 2
        B = sinla * mp.sqrt(1 / (1 + alpha2 * beta2_inv))                                # OGG:47-50
    lamc = mp.asin(B) / PI180                                                            # OGG:53
    dl = lamg - lon_bp
    if 90 < dl <= 180:
        lamc = 180 - lamc
    if 180 < dl <= 270:
        lamc = 180 + lamc
    if dl > 270:
        lamc = 360 - lamc
    if dl == 90:
        lamc = M(90.0)
    if dl == 270:
        lamc = M(270.0)
    lams = lamc + lon_bp                                                                 # OGG:58-64
    A = sinla * sphig
    chic = mp.acos(A)
    phis = 90 - 2 * mp.atan(rp * mp.tan(chic / 2)) / PI180                               # OGG:68-70
    return split(lams), split(phis), bool(guard)


def bp_group(pool, quick):
    out = {}
    for Ni, Nj, lat0 in ((5760, 960, 64.03160594077568), (11520, 1920, 64.04528618884338)):
        lon_bp = -300.0
        lon_g = lon_bp + np.arange(Ni + 1) * 360.0 / float(Ni)                           # OGG:108-115, fp64 as the reference forms them
        latg0 = lat0 + np.arange(Nj + 1) * (90 - lat0) / float(Nj)
        rp = float(np.tan(0.5 * (90 - lat0) * orc.PI_180))
        pts = []
        for c0 in (Ni // 4, 3 * Ni // 4):
            for c in range(c0 - 2, c0 + 3):
                pts += [(j, c) for j in range(0, Nj + 1, 50 if quick else 1)]
        pts += [(Nj, c) for c in range(0, Ni + 1, 50 if quick else 1)]
        pts = sorted(set(pts))
        jj = np.array([p[0] for p in pts])
        ii = np.array([p[1] for p in pts])
        t0 = time.time()
        res = pool.map(bp_point, [(lon_g[i], latg0[j], lon_bp, rp) for j, i in pts], chunksize=64)
        print("bp %d: %d points in %.0f s" % (Ni, len(pts), time.time() - t0), flush=True)
        tag = "bp%d_" % Ni
        lam_t = np.array([r[0] for r in res])
        phi_t = np.array([r[1] for r in res])
        guard = np.array([r[2] for r in res])
        lamg = np.tile(lon_g, (Nj + 1, 1))
        phig = np.tile(latg0[:, None], (1, Ni + 1))
        ol, op, _, _ = orc.bipolar_projection(lamg, phig, lon_bp, rp)
        # the fp64 evaluation's own guard state (OGG:52), to make sure truth and fp64 took the same branch
        ph2 = 90 - 2 * np.arctan(np.tan(0.5 * (90 - phig[jj, ii]) * orc.PI_180) / rp) / orc.PI_180
        g64 = np.abs(np.tan(ph2 * orc.PI_180) ** 2) > 1e30
        same = g64 == guard
        print("bp %d: guard taken at %d points in fp64, %d in exact arithmetic, same branch at %d of %d" %
              (Ni, g64.sum(), guard.sum(), same.sum(), same.size))
        out.update({tag + "params": np.array([Ni, Nj, lat0, lon_bp, rp]), tag + "j": jj, tag + "i": ii, tag + "lams": lam_t,
                    tag + "phis": phi_t, tag + "same_branch": same})
        pole_row = jj == Nj
        for f, o, t in (("lams", ol, lam_t), ("phis", op, phi_t)):
            err = np.abs((o[jj, ii] - t[:, 0]) - t[:, 1])
            for nm, m in (("meridian", ~pole_row & same), ("polerow", pole_row & same)):
                out[tag + f + "_eref_" + nm] = np.array(err[m].max())
                print("bp %d oracle vs truth %s %s: %.3e deg" % (Ni, f, nm, err[m].max()))
    return out


# ---------------------------------------------------------------------------------------------------------------
# bq: OGG:125-188 (bipolar_cap_metrics_quad_fast, order 5) over OGG:33-100 (metrics_only) in exact arithmetic
# ---------------------------------------------------------------------------------------------------------------
def bq_metrics(lamg, phig, lon_bp, rp):
    """h_i_inv, h_j_inv of bipolar_projection(metrics_only=True) at one point.  OGG:41-47, 66-95."""
    phig = 90 - 2 * mp.atan(mp.tan(M(0.5) * (90 - phig) * PI180) / rp) / PI180        # OGG:41
    tmp = mdist_exact(lamg, lon_bp) * PI180
    sinla = mp.sin(tmp)
    sphig = mp.sin(phig * PI180)
    alpha2 = mp.cos(tmp) ** 2
    t = phig * PI180
    huge = (mp.cos(t) == 0) or (mp.tan(t) ** 2 > HUGE)
    A = sinla * sphig
    chic = mp.acos(A)
    tc = mp.tan(chic / 2)
    phis = 90 - 2 * mp.atan(rp * tc) / PI180
    M_inv = rp * (1 + tc ** 2) * (1 / (1 + (rp * tc) ** 2))                            # OGG:72-73
    chig = (90 - phig) * PI180
    tg = mp.tan(chig / 2)
    N = rp * (1 + tg ** 2) * (1 / (1 + (rp * tg) ** 2))                                # OGG:76-77
    N_inv = 1 / N
    cos2phis = mp.cos(phis * PI180) ** 2
    if huge:                                                                           # OGG:86, 94
        hj2, hi2 = M_inv * M_inv, M_inv * M_inv
    else:
        b2 = mp.tan(t) ** 2
        rden = 1 / (1 + alpha2 * b2)
        hj2 = cos2phis * alpha2 * (1 - alpha2) * b2 * (1 + b2) * rden ** 2 + M_inv * M_inv * (1 - alpha2) * rden
        hi2 = cos2phis * (1 + b2) * rden ** 2 + M_inv * M_inv * alpha2 * b2 * rden
    return mp.sqrt(hi2), mp.sqrt(hj2) * N_inv


def bq_cell(args):
    (nx, ny, lat0, lon_bp, rp), j, i, jn, inn = args
    lat0m, lonm, rpm = M(lat0), M(lon_bp), M(rp)
    hi = [[None] * 5 for _ in range(5)]
    hj = [[None] * 5 for _ in range(5)]
    for a in range(5):
        latg = lat0m + M(jn[a]) * (90 - lat0m) / ny                                    # OGG:127
        for b in range(5):
            long = lonm + M(inn[b]) * 360 / nx                                         # OGG:126
            x, y = bq_metrics(long, latg, lonm, rpm)
            hi[a][b] = x * 2 * M(np.pi) / nx                                           # OGG:131-132
            hj[a][b] = y * (90 - lat0m) * PI180 / ny
    w = [9, 49, 64, 49, 9]
    d = M(1.0) / 180
    area = d * d * sum(w[a] * w[b] * hi[a][b] * hj[a][b] for a in range(5) for b in range(5))
    dx = d * (64 * hi[0][2] + 49 * (hi[0][1] + hi[0][3]) + 9 * (hi[0][0] + hi[0][4]))   # OGG:219-220
    dy = d * (64 * hj[2][0] + 49 * (hj[1][0] + hj[3][0]) + 9 * (hj[0][0] + hj[4][0]))
    return j, i, [split(dx * RE), split(dy * RE), split(area * RE * RE)]


def bq_group(pool, quick):
    out = {}
    for Ni, Nj, lat0 in ((5760, 960, 64.03160594077568),):
        lon_bp = -300.0
        rp = float(np.tan(0.5 * (90 - lat0) * orc.PI_180))
        rows = [0, 1, 2, 240, 480, 720, 900, 940, 950, 955, 956, 957, 958, 959]
        cols = sorted(set(list(range(0, Ni, 37)) + [c for m in (Ni // 4, 3 * Ni // 4) for c in range(m - 5, m + 5)] + [Ni // 2 - 1, Ni // 2, Ni - 1]))
        if quick:
            rows, cols = rows[::5] + [959], cols[::20] + [Ni // 4 - 1, Ni // 4]
        nodes = orc._lattice_1d(Nj, 5).reshape(Nj + 1, 5)
        nodes[:, -1] = np.where(nodes[:, -1] == Nj, Nj - 0.001, nodes[:, -1])           # OGG:146-147
        i1d = orc._lattice_1d(Ni, 5).reshape(Ni + 1, 5)
        par = (Ni, Nj, lat0, lon_bp, rp)
        jobs = [(par, j, i, nodes[j], i1d[i]) for j in rows for i in cols]
        t0 = time.time()
        res = pool.map(bq_cell, jobs, chunksize=8)
        print("bq %d: %d cells in %.0f s" % (Ni, len(jobs), time.time() - t0), flush=True)
        tag = "bq%d_" % Ni
        jj = np.array([r[0] for r in res])
        ii = np.array([r[1] for r in res])
        vals = np.array([r[2] for r in res])
        out.update({tag + "params": np.array([Ni, Nj, lat0, lon_bp, rp]), tag + "j": jj, tag + "i": ii})
        o = [np.zeros(jj.size) for _ in range(3)]
        for j in sorted(set(jj.tolist())):
            r = orc.bipolar_cap_metrics_quad_fast(5, Ni, Nj, lat0, lon_bp, rp, j_first=j, j_last=j + 1)
            m = jj == j
            for k in range(3):
                o[k][m] = r[k][j, ii[m]]
        for k, f in enumerate(("dx", "dy", "area")):
            t = vals[:, k, :]
            out[tag + f] = t.copy()
            err = np.abs((o[k] - t[:, 0]) - t[:, 1])
            # dy on the fold lines i = 0 and i = Ni/2 is 0 up to rounding (1 - cos^2 of 0 or pi: exactly 0 at i = 0, 1e-9 m at Ni/2): such
            # values are compared absolutely (they are 1e-12 of the field), the relative figures are over the rest
            nz = np.abs(t[:, 0]) > 1e-6 * np.abs(t[:, 0]).max()
            assert np.all(np.abs(o[k][~nz]) < 1e-6) and err[~nz].max(initial=0.0) < 1e-6
            pole = (jj == Nj - 1) & np.isin(ii, (Ni // 4 - 1, Ni // 4, 3 * Ni // 4 - 1, 3 * Ni // 4))   # the four cells that touch a pole point
            # cells within 6 columns of a symmetry meridian (acos(A) at A -> 1 towards the pole points) or next to the fold lines
            # i = 0, Ni/2 (1 - cos^2 of an angle next to 0 or pi: cancellation in the reference's own formula, OGG:82-84)
            edge = ((np.abs(ii - Ni // 4) <= 6) | (np.abs(ii - 3 * Ni // 4) <= 6) | (np.abs(ii - Ni // 2) <= 1) | (ii <= 1) | (ii >= Ni - 2)) & ~pole
            out[tag + "pole_cells"], out[tag + "edge_cells"] = pole, edge
            for nm, m in (("", nz & ~pole & ~edge), ("_edgecells", nz & edge), ("_polecells", nz & pole)):
                out[tag + f + "_eref_rel" + nm] = np.array((err[m] / np.abs(t[m, 0])).max())
                out[tag + f + "_eref_abs" + nm] = np.array(err[m].max())
                print("bq %d oracle vs truth %s%s: rel %.3e abs %.3e" % (Ni, f, nm, (err[m] / np.abs(t[m, 0])).max(), err[m].max()))
    return out


# ---------------------------------------------------------------------------------------------------------------
# round 5: the rows of the table that had no truth yet
#   ax   OGG:719-729  angle_x over the EXACT mesh (bipolar: the pole row and the three rows under it; displaced pole: kept rows): what the
#                     chain "mesh, then angle_x" returns without rounding anywhere.  The fp64 reference's distance from it is dominated by
#                     the last-bit errors of its x, y, divided by the mesh spacing (5e-11 degrees at 1/8 degree; unbounded at the pole points)
#   dm   OGG:447-467  displaced-pole mesh x, y on the three rows around r = r_pole and on two kept rows, every column
#   bq11520           the bipolar quadrature at 1/16 degree: the four cells that touch a pole point + 200 regular cells
#   mdso / mdsc       MIDAS metrics on the Southern Ocean and regular southern-cap sub-grids of the 1/8 degree grid
# ---------------------------------------------------------------------------------------------------------------
def angle_exact(xl, xr, yl, yr, yc):
    """angle_x at an interior column from exact neighbours (degrees).  OGG:725."""
    return mp.atan2(yr - yl, (xr - xl) * mp.cos(yc * PI180)) / PI180


def ax_bp_point(args):
    (Ni, Nj, lat0, lon_bp, rp), j, i, lon3, lat = args
    pts = [bp_point((lo, lat, lon_bp, rp)) for lo in lon3]           # columns i-1, i, i+1 (fp64 inputs as the reference forms them)
    x = [M(p[0][0]) + M(p[0][1]) for p in pts]
    y = [M(p[1][0]) + M(p[1][1]) for p in pts]
    if i == 0:
        a = mp.atan2(y[2] - y[1], (x[2] - x[1]) * mp.cos(y[1] * PI180)) / PI180          # OGG:726
    elif i == Ni:
        a = mp.atan2(y[1] - y[0], (x[1] - x[0]) * mp.cos(y[1] * PI180)) / PI180          # OGG:727
    else:
        a = angle_exact(x[0], x[2], y[0], y[2], y[1])
    return j, i, split(a)


def ax_dp_point(args):
    par, j, i, x64 = args                                            # x64: the oracle's unwrapped longitudes at i-1, i, i+1 (fixes the 360 k)
    nx = par[0]
    cap = Cap(*par)
    x, y = [], []
    for k, c in enumerate((i - 1, i, i + 1)):
        c = min(max(c, 0), nx)
        lam, phi = cap.project(M(float(c)), M(float(j)))
        lam, phi = lam / PI180, phi / PI180
        lam = lam + 360 * mp.nint((M(x64[k]) - lam) / 360)          # monotonic_bounding moves lam by multiples of 360 (OGG:470-475)
        x.append(lam)
        y.append(phi)
    if i == 0:
        a = mp.atan2(y[2] - y[1], (x[2] - x[1]) * mp.cos(y[1] * PI180)) / PI180
    elif i == nx:
        a = mp.atan2(y[1] - y[0], (x[1] - x[0]) * mp.cos(y[1] * PI180)) / PI180
    else:
        a = angle_exact(x[0], x[2], y[0], y[2], y[1])
    return j, i, split(a), split(x[1]), split(y[1])


def ax_group(pool, quick):
    out = {}
    # bipolar caps
    for Ni, Nj, lat0 in ((5760, 960, 64.03160594077568), (11520, 1920, 64.04528618884338)):
        lon_bp = -300.0
        rp = float(np.tan(0.5 * (90 - lat0) * orc.PI_180))
        lon_g = lon_bp + np.arange(Ni + 1) * 360.0 / float(Ni)
        latg0 = lat0 + np.arange(Nj + 1) * (90 - lat0) / float(Nj)
        q = Ni // 4
        cols = sorted(set(list(range(0, Ni + 1, 64 if quick else 16)) + [c for m in (q, 3 * q) for c in range(m - 8, m + 9)] +
                          [0, 1, Ni - 1, Ni, 2 * q - 1, 2 * q, 2 * q + 1]))
        rows = [Nj - 3, Nj - 2, Nj - 1, Nj, Nj // 2]
        par = (Ni, Nj, lat0, lon_bp, rp)
        jobs = [(par, j, i, [lon_g[min(max(c, 0), Ni)] for c in (i - 1, i, i + 1)], latg0[j]) for j in rows for i in cols]
        t0 = time.time()
        res = pool.map(ax_bp_point, jobs, chunksize=16)
        print("ax bp %d: %d points in %.0f s" % (Ni, len(jobs), time.time() - t0), flush=True)
        tag = "axbp%d_" % Ni
        jj, ii = np.array([r[0] for r in res]), np.array([r[1] for r in res])
        t = np.array([r[2] for r in res])
        x, y, _, _ = orc.generate_bipolar_cap_mesh(Ni, Nj, lat0, lon_bp, ensure_nj_even=False)
        oa = orc.angle_x(x, y)[jj, ii]
        err = np.abs((oa - t[:, 0]) - t[:, 1])
        err = np.minimum(err, np.abs(err - 360.0))
        near = (np.abs(ii - q) <= 8) | (np.abs(ii - 3 * q) <= 8)          # the columns around the two pole points
        out.update({tag + "params": np.array([Ni, Nj, lat0, lon_bp, rp]), tag + "j": jj, tag + "i": ii, tag + "angle": t, tag + "near_pole_columns": near})
        for nm, m in (("away", ~near), ("near", near & (jj < Nj)), ("poleline", near & (jj == Nj))):
            out[tag + "eref_" + nm] = np.array(err[m].max())
            print("ax bp %d oracle vs truth, %s: max %.3e deg, median %.3e" % (Ni, nm, err[m].max(), np.median(err[m])))
    # displaced-pole caps (config 4 and OM4): kept rows
    for tag, par, rows in (("axdp_", (5760, 560, -300.0, -78.0, 80.0, float(np.tan((90 - 85.85) * orc.PI_180) / np.tan(12.0 * orc.PI_180))),
                            [276, 277, 300, 400, 500, 559, 560]),
                           ("axdp4_", (2880, 280, -300.0, -78.0, 80.0, 0.2), [220, 221, 250, 279, 280])):
        nx, ny = par[0], par[1]
        cols = sorted(set(list(range(0, nx + 1, 176 if quick else 44)) + [0, 1, nx - 1, nx]))
        x, y, _, _ = orc.generate_displaced_pole_grid(nx, ny, par[2], par[3], par[4], par[5])
        jobs = [(par, j, i, [x[j, min(max(c, 0), nx)] for c in (i - 1, i, i + 1)]) for j in rows for i in cols]
        t0 = time.time()
        res = pool.map(ax_dp_point, jobs, chunksize=16)
        print("ax %s: %d points in %.0f s" % (tag, len(jobs), time.time() - t0), flush=True)
        jj, ii = np.array([r[0] for r in res]), np.array([r[1] for r in res])
        t = np.array([r[2] for r in res])
        oa = orc.angle_x(x, y)[jj, ii]
        err = np.abs((oa - t[:, 0]) - t[:, 1])
        err = np.minimum(err, np.abs(err - 360.0))
        out.update({tag + "params": np.array(par), tag + "j": jj, tag + "i": ii, tag + "angle": t, tag + "eref": np.array(err.max())})
        print("ax %s oracle vs truth: max %.3e deg, median %.3e" % (tag, err.max(), np.median(err)))
    return out


def dm_row(args):
    par, j, cols, x64 = args
    cap = Cap(*par)
    out = []
    for i, xo in zip(cols, x64):
        lam, phi = cap.project(M(float(i)), M(float(j)))
        lam, phi = lam / PI180, phi / PI180
        lam = lam + 360 * mp.nint((M(xo) - lam) / 360)
        out.append([split(lam), split(phi)])
    return out


def dm_group(pool, quick):
    out = {}
    for tag, par in (("dm_", (5760, 560, -300.0, -78.0, 80.0, float(np.tan((90 - 85.85) * orc.PI_180) / np.tan(12.0 * orc.PI_180)))),
                     ("dm4_", (2880, 280, -300.0, -78.0, 80.0, 0.2))):
        nx, ny, lon0, lat0, lon_dp, r_dp = par
        jpole = float(np.arctan(r_dp * np.tan(12.0 * orc.PI_180)) / orc.PI_180 / (12.0 / ny))      # the row where r = r_pole
        rows = [int(jpole) - 1, int(jpole), int(jpole) + 1] + ([276, 559] if nx == 5760 else [220, 279])
        cols = np.arange(0, nx + 1, 64 if quick else 1)
        x, y, _, _ = orc.generate_displaced_pole_grid(nx, ny, lon0, lat0, lon_dp, r_dp)
        chunks = np.array_split(cols, 64)
        jobs = [(par, j, c, x[j, c]) for j in rows for c in chunks]
        t0 = time.time()
        res = pool.map(dm_row, jobs, chunksize=1)
        print("dm %s: %d points in %.0f s" % (tag, len(rows) * cols.size, time.time() - t0), flush=True)
        t = np.array([p for r in res for p in r]).reshape(len(rows), cols.size, 2, 2)
        out.update({tag + "params": np.array(par), tag + "rows": np.array(rows), tag + "cols": cols, tag + "x": t[:, :, 0, :].copy(), tag + "y": t[:, :, 1, :].copy(),
                    tag + "jpole": np.array(jpole)})
        for f, o, k in (("x", x, 0), ("y", y, 1)):
            v = o[rows][:, cols]
            err = np.abs((v - t[:, :, k, 0]) - t[:, :, k, 1])
            out[tag + f + "_eref_polar_rows"] = np.array(err[:3].max())
            out[tag + f + "_eref_kept_rows"] = np.array(err[3:].max())
            print("dm %s oracle vs truth %s: rows around r_pole %.3e deg, kept rows %.3e deg" % (tag, f, err[:3].max(), err[3:].max()))
    return out


def bq16_group(pool, quick):
    Ni, Nj, lat0, lon_bp = 11520, 1920, 64.04528618884338, -300.0
    rp = float(np.tan(0.5 * (90 - lat0) * orc.PI_180))
    q = Ni // 4
    cells = [(Nj - 1, c) for c in (q - 1, q, 3 * q - 1, 3 * q)]
    rng = np.random.default_rng(11)
    rows = [0, 1, 480, 960, 1440, 1780, 1800, 1880, 1910, 1917]
    for j in rows:
        cells += [(j, int(c)) for c in rng.integers(0, Ni, 6 if quick else 20)]
    nodes = orc._lattice_1d(Nj, 5).reshape(Nj + 1, 5)
    nodes[:, -1] = np.where(nodes[:, -1] == Nj, Nj - 0.001, nodes[:, -1])
    i1d = orc._lattice_1d(Ni, 5).reshape(Ni + 1, 5)
    par = (Ni, Nj, lat0, lon_bp, rp)
    t0 = time.time()
    res = pool.map(bq_cell, [(par, j, i, nodes[j], i1d[i]) for j, i in cells], chunksize=4)
    print("bq 11520: %d cells in %.0f s" % (len(cells), time.time() - t0), flush=True)
    tag = "bq11520_"
    jj, ii = np.array([r[0] for r in res]), np.array([r[1] for r in res])
    vals = np.array([r[2] for r in res])
    out = {tag + "params": np.array([Ni, Nj, lat0, lon_bp, rp]), tag + "j": jj, tag + "i": ii}
    o = [np.zeros(jj.size) for _ in range(3)]
    for j in sorted(set(jj.tolist())):
        r = orc.bipolar_cap_metrics_quad_fast(5, Ni, Nj, lat0, lon_bp, rp, j_first=j, j_last=j + 1)
        m = jj == j
        for k in range(3):
            o[k][m] = r[k][j, ii[m]]
    pole = (jj == Nj - 1) & np.isin(ii, (q - 1, q, 3 * q - 1, 3 * q))
    edge = ((np.abs(ii - q) <= 6) | (np.abs(ii - 3 * q) <= 6) | (np.abs(ii - Ni // 2) <= 1) | (ii <= 1) | (ii >= Ni - 2)) & ~pole
    out[tag + "pole_cells"], out[tag + "edge_cells"] = pole, edge
    for k, f in enumerate(("dx", "dy", "area")):
        t = vals[:, k, :]
        out[tag + f] = t.copy()
        err = np.abs((o[k] - t[:, 0]) - t[:, 1]) / np.abs(t[:, 0])
        for nm, m in (("", ~pole & ~edge), ("_polecells", pole)):
            out[tag + f + "_eref_rel" + nm] = np.array(err[m].max())
            print("bq 11520 oracle vs truth %s%s: rel %.3e" % (f, nm, err[m].max()))
    return out


def mdso_group(pool, quick):
    """MIDAS on the Southern Ocean sub-grid and the regular southern cap of the 1/8 degree grid (axes as main() forms them, OGG:1080-1157)."""
    out = {}
    Ni = 5760
    phiM = orc.generate_mercator_grid(Ni, -66.85954725, 64.05895973, -300.0, 360, 8.0, True, False)[1][:, 0]
    latUp = float(phiM[0])
    xso, yso = orc.generate_latlon_grid(Ni, 440, -300.0, 360.0, -78.0, latUp + 78.0, ensure_nj_even=False)
    xsc, ysc = orc.generate_latlon_grid(Ni, 192, -300.0, 360.0, -90.0, 12.0, ensure_nj_even=False)
    for tag, x, y in (("mdso_", xso, yso), ("mdsc_", xsc, ysc)):
        xa, ya = x[0].copy(), y[:, 0].copy()
        cols = np.array([0, 1, Ni // 4 - 1, Ni // 4, Ni // 2 - 1, Ni // 2, 3 * Ni // 4 + 1, Ni - 2, Ni - 1])
        rows = np.arange(0, ya.size - 1, 20 if quick else 1)
        res = np.array(pool.map(md_rows, [(np.stack([xa[cols], xa[cols + 1]], 1), ya[j:j + 2]) for j in rows], chunksize=16))
        out.update({tag + "xaxis": xa, tag + "yaxis": ya, tag + "cols": cols, tag + "rows": rows})
        odx, ody, oar = orc.generate_grid_metrics_MIDAS(np.tile(xa, (ya.size, 1)), np.tile(ya[:, None], (1, xa.size)))
        for k, (f, o) in enumerate((("dx", odx), ("dy", ody), ("area", oar))):
            t = res[:, :, k, :]
            out[tag + f] = t.copy()
            v = o[rows][:, cols]
            nz = np.abs(t[..., 0]) > 1e-6 * np.abs(t[..., 0]).max()         # dx on the pole row of the cap (cos(-90 degrees) ~ 6e-17): absolute only
            err = np.abs((v - t[..., 0]) - t[..., 1])
            out[tag + f + "_eref_abs"] = np.array(err.max())
            out[tag + f + "_eref_rel"] = np.array((err[nz] / np.abs(t[..., 0][nz])).max())
            print("%s oracle vs truth %s: abs %.3e rel %.3e" % (tag, f, err.max(), out[tag + f + "_eref_rel"]))
    return out


GROUPS = {"md": md_group, "bq": bq_group, "bp": bp_group, "dp": dp_group, "dp4": lambda pool, quick: dp_group(pool, quick, "dp4"),
          "ax": ax_group, "dm": dm_group, "bq16": bq16_group, "mdso": mdso_group}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--groups", nargs="+", default=list(GROUPS), help="which groups to (re)compute; the others are kept from the existing fixture")
    ap.add_argument("--procs", type=int, default=8)
    ap.add_argument("--quick", action="store_true", help="a thin sample (smoke run of this script; does not overwrite the fixture)")
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden", "truth_table.npz"))
    a = ap.parse_args()
    out = {"dps": np.array(mp.mp.dps)}
    if not a.quick and os.path.exists(a.out) and set(a.groups) != set(GROUPS):
        out.update({k: v for k, v in np.load(a.out).items()})              # keep the groups that are not recomputed
    with Pool(a.procs) as pool:
        for g in a.groups:
            out.update(GROUPS[g](pool, a.quick))
    path = a.out if not a.quick else "/tmp/truth_table_quick.npz"
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
