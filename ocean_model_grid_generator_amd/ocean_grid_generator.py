#!/usr/bin/env python
"""Host side of the MI355X supergrid generator: the function and command-line surface of the reference
``ocean_grid_generator.py`` (cited as OGG:<line>), with every array computation delegated to hand-written HIP
kernels in ``libogg_hip.so`` through ctypes (``_lib``).

Same function names, argument meaning and error behaviour as the reference, so it can be used as a drop-in::

    import ocean_model_grid_generator_amd.ocean_grid_generator as ogg
    dx, dy, area = ogg.generate_grid_metrics_MIDAS(x, y)

What stays on the host (as in BASELINE.json's north_star): flag parsing, sub-grid size selection, the 1-D
enhanced-equator splice, stitching of the four sub-grids, guards, NetCDF output.  There is no CPU implementation
of the array maths in this package: if the HIP library or a GPU is missing, calls raise.
"""
from __future__ import print_function

import argparse
import datetime
import os
import subprocess
import sys

import numpy as np

from . import _lib as L
from .plotting import cut_above, cut_below, displacedPoleCap_plot, plot_mesh_in_latlon, plot_mesh_in_xyz  # noqa: F401  (OGG:604-679)

# Constants (OGG:13-16)
PI_180 = np.pi / 180.0
_default_Re = 6371.0e3  # MIDAS
HUGE = 1.0e30


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _new(*shape):
    return np.empty(shape, dtype=np.float64)


# ----------------------------------------------------------------------------------------------------------------
# diagnostics (host)
# ----------------------------------------------------------------------------------------------------------------
def chksum(x, lbl):
    """sha256 + statistics line of OGG:19-30."""
    import hashlib

    if type(x) in (float, int, np.float64):
        y = np.array(x)
    else:
        y = np.zeros(x.shape)
        y[:] = x
    ymin, ymax, ymean = y.min(), y.max(), y.mean()
    ysd = np.sqrt(((y - ymean) ** 2).mean())
    print(hashlib.sha256(y).hexdigest(), "%10s" % lbl, "min = %.15f" % ymin, "max = %.15f" % ymax,
          "mean = %.15f" % ymean, "sd = %.15f" % ysd)


# ----------------------------------------------------------------------------------------------------------------
# bipolar cap (OGG:33-188)
# ----------------------------------------------------------------------------------------------------------------
def bipolar_projection(lamg, phig, lon_bp, rp, metrics_only=False):
    """OGG:33-100 on the GPU (kernel bipolar_projection_kernel)."""
    lamg, phig = _f64(lamg), _f64(phig)
    if lamg.shape != phig.shape:
        lamg, phig = [np.ascontiguousarray(a) for a in np.broadcast_arrays(lamg, phig)]
    shp = lamg.shape
    hi, hj = _new(*shp), _new(*shp)
    lams = phis = None
    if not metrics_only:
        lams, phis = _new(*shp), _new(*shp)
    L.call("ogg_bipolar_projection", lamg.size, L.ptr(lamg), L.ptr(phig), float(lon_bp), float(rp), int(bool(metrics_only)),
           L.ptr(lams), L.ptr(phis), L.ptr(hi), L.ptr(hj))
    if not metrics_only:
        return lams, phis, hi, hj
    return hi, hj


def _sym(symmetry):
    """``symmetry`` of the cap functions (not in the reference): None -> the library's default (mirrored columns unless OGG_CAP_SYMMETRY=0),
    True / "mirror" -> OGG_SYM_MIRROR, False / "none" -> OGG_SYM_NONE: every column evaluated, as the reference does."""
    if symmetry is None:
        return L.SYM_DEFAULT
    if symmetry in (True, "mirror"):
        return L.SYM_MIRROR
    if symmetry in (False, "none"):
        return L.SYM_NONE
    raise ValueError("symmetry must be None, True / 'mirror' or False / 'none', not %r" % (symmetry,))


def generate_bipolar_cap_mesh(Ni, Nj_ncap, lat0_bp, lon_bp, ensure_nj_even=True, symmetry=None):
    """OGG:103-122.  ``symmetry`` (not in the reference): see _sym; mirrored, a quarter of the columns (+ the neighbourhoods of the pole
    meridians and of the fold lines) is evaluated and written to its images (csrc/ogg_bipolar_dev.h, MeshCols)."""
    print("Generating bipolar grid bounded at latitude ", lat0_bp)
    if Nj_ncap % 2 != 0 and ensure_nj_even:
        print("   Supergrid has an odd number of area cells!")
        print("   The number of j's is not even. Fixing this by cutting one row.")
        Nj_ncap = Nj_ncap - 1
    Ni, Nj_ncap = int(Ni), int(Nj_ncap)
    lams, phis = _new(Nj_ncap + 1, Ni + 1), _new(Nj_ncap + 1, Ni + 1)
    h_i_inv, h_j_inv = _new(Nj_ncap + 1, Ni), _new(Nj_ncap, Ni + 1)
    L.call("ogg_bipolar_cap_mesh_sym", Ni, Nj_ncap, float(lat0_bp), float(lon_bp), _sym(symmetry), L.ptr(lams), L.ptr(phis), L.ptr(h_i_inv),
           L.ptr(h_j_inv))
    print("   number of js=", phis.shape[0])
    return lams, phis, h_i_inv, h_j_inv


def bipolar_cap_ij_array(i, j, Ni, Nj_ncap, lat0_bp, lon_bp, rp):
    """OGG:125-133."""
    i, j = _f64(i).reshape(-1), _f64(j).reshape(-1)
    hi, hj = _new(j.size, i.size), _new(j.size, i.size)
    L.call("ogg_bipolar_cap_ij_array", i.size, L.ptr(i), j.size, L.ptr(j), int(Ni), int(Nj_ncap), float(lat0_bp), float(lon_bp),
           float(rp), L.ptr(hi), L.ptr(hj))
    return hi, hj


def bipolar_cap_metrics_quad_fast(order, nx, ny, lat0_bp, lon_bp, rp, Re=_default_Re, symmetry=None):
    """OGG:136-188 (kernel bipolar_quad_kernel: the lattice is evaluated and reduced on chip).  ``symmetry`` (not in the reference): see
    _sym; mirrored, the rows below 88.2 degrees are evaluated on the cells [0, nx/4) and next to the fold lines and written to their images
    (csrc/ogg_bipolar_dev.h, QuadCols)."""
    print("   Calculating bipolar cap metrics via quadrature ...")
    nx, ny = int(nx), int(ny)
    dxq, dyq, daq = _new(ny + 1, nx), _new(ny, nx + 1), _new(ny, nx)
    L.call("ogg_bipolar_cap_metrics_quad_sym", int(order), nx, ny, float(lat0_bp), float(lon_bp), float(rp), float(Re), _sym(symmetry),
           L.ptr(dxq), L.ptr(dyq), L.ptr(daq))
    return dxq, dyq, daq


# ----------------------------------------------------------------------------------------------------------------
# quadrature helpers: a handful of scalars, host (OGG:191-255)
# ----------------------------------------------------------------------------------------------------------------
def quad_positions(n=3):
    """Lobatto node weights; same values as the kernels use (make_nodes in ogg_bipolar.hip)."""
    if n == 2:
        return np.array([0.0, 1.0]), np.array([1.0, 0.0])
    if n == 3:
        return np.array([0.0, 0.5, 1.0]), np.array([1.0, 0.5, 0.0])
    if n == 4:
        r5 = 0.5 / np.sqrt(5.0)
        return np.array([0.0, 0.5 - r5, 0.5 + r5, 1.0]), np.array([1.0, 0.5 + r5, 0.5 - r5, 0.0])
    if n == 5:
        r37 = 0.5 * np.sqrt(3.0 / 7.0)
        return np.array([0.0, 0.5 - r37, 0.5, 0.5 + r37, 1.0]), np.array([1.0, 0.5 + r37, 0.5, 0.5 - r37, 0.0])
    raise Exception("Uncoded order")


def quad_average(y):
    n = len(y)
    if n == 2:
        return (1.0 / 2.0) * (y[0] + y[1])
    if n == 3:
        return (1.0 / 6.0) * (4.0 * y[1] + (y[0] + y[2]))
    if n == 4:
        return (1.0 / 12.0) * (5.0 * (y[1] + y[2]) + (y[0] + y[3]))
    if n == 5:
        return (1.0 / 180.0) * (64.0 * y[2] + (49.0 * (y[1] + y[3])) + 9.0 * (y[0] + y[4]))
    raise Exception("Uncoded order")


def quad_average_2d(y):
    if y.shape[0] != y.shape[1]:
        raise Exception("Input array is not squared!")
    n = y.shape[0]
    if n == 2:
        d = 1.0 / 2.0
        return d * d * (y[0, 0] + y[0, 1] + y[1, 0] + y[1, 1])
    if n == 3:
        d = 1.0 / 6.0
        return d * d * (y[0, 0] + y[0, 2] + y[2, 0] + y[2, 2] + 4.0 * (y[0, 1] + y[1, 0] + y[1, 2] + y[2, 1] + 4.0 * y[1, 1]))
    if n in (4, 5):
        d, w = (1.0 / 12.0, [1.0, 5.0, 5.0, 1.0]) if n == 4 else (1.0 / 180.0, [9.0, 49.0, 64.0, 49.0, 9.0])
        ysum = 0.0
        for j in range(n):
            for i in range(n):
                ysum = ysum + w[i] * w[j] * y[j, i]
        return d * d * ysum
    raise Exception("Uncoded order")


def lagrange_interp(x, y, q):
    """4-point Lagrange interpolation used by the 1-D enhanced-equator splice (host, OGG:258-270)."""
    out = 0.0
    terms = []
    for k in range(4):
        n, d = 1.0, 1.0
        for m in range(4):
            if m != k:
                n = n * (q - x[m])
                d = d * (x[k] - x[m])
        terms.append((n / d) * y[k])
    out = (terms[0] + terms[3]) + (terms[1] + terms[2])
    return out


def lagrange_interp_6pt(x, y, q):
    """6-point variant (OGG:272-289)."""
    terms = []
    for k in range(6):
        n, d = 1.0, 1.0
        for m in range(6):
            if m != k:
                n = n * (q - x[m])
                d = d * (x[k] - x[m])
        terms.append((n / d) * y[k])
    return (terms[0] + terms[5]) + (terms[1] + terms[4]) + (terms[2] + terms[3])


# ----------------------------------------------------------------------------------------------------------------
# Mercator (OGG:292-441)
# ----------------------------------------------------------------------------------------------------------------
def y_mercator(Ni, phi):
    phi = _f64(np.atleast_1d(phi))
    out = _new(*phi.shape)
    L.call("ogg_y_mercator", int(Ni), phi.size, L.ptr(phi), L.ptr(out))
    return out


def phi_mercator(Ni, y):
    y = _f64(np.atleast_1d(y))
    out = _new(*y.shape)
    L.call("ogg_phi_mercator", int(Ni), y.size, L.ptr(y), L.ptr(out))
    return out


def y_mercator_rounded(Ni, phi):
    phi = _f64(np.atleast_1d(phi))
    out = np.empty(phi.shape, dtype=np.int64)
    L.call("ogg_y_mercator_rounded", int(Ni), phi.size, L.ptr(phi), L.ptr(out))
    return out.astype(int)


def _enhance_equator(phi_M, phi_n, refineR, enhanced_equatorial):
    """1-D host logic of OGG:349-428: pure Mercator | Lagrange shoulder | uniform band | mirror image."""
    print("   Enhancing the equator region resolution")
    use_4pt, use_6pt = True, False
    phi_enh_d, phi_cub_d = -5.0, -30
    N_cub = 132 * refineR / 2
    dphi_e = 0.13 * 2 / refineR
    N_enh = 40 * refineR / 2
    if refineR == 1 and enhanced_equatorial:  # closest to SPEAR
        phi_enh_d, phi_cub_d, N_cub, N_enh = -10, -20, 29, 55
        dphi_e = -phi_enh_d / N_enh / 0.981
    if refineR == 4 and enhanced_equatorial in (6, 8):
        phi_enh_d = -10
        N_enh = 2 * enhanced_equatorial * abs(phi_enh_d) + 1
        phi_cub_d, N_cub = -20, 101
        dphi_e = -phi_enh_d / N_enh
        if enhanced_equatorial == 8:
            use_4pt, use_6pt = False, True
    j_cub = np.where(phi_M < phi_cub_d)[0][-1]
    phi1 = phi_M[0:j_cub]
    if use_4pt:
        nodes = [0, 1, N_cub - 2, N_cub - 1]
        vals = [phi_M[j_cub - 1], phi_M[j_cub], phi_enh_d - dphi_e, phi_enh_d]
        phi2 = lagrange_interp(nodes, vals, np.arange(N_cub))
    elif use_6pt:
        N_cub = 111
        nodes = [0, 1, 2, N_cub - 3, N_cub - 2, N_cub - 1]
        vals = [phi_M[j_cub - 1], phi_M[j_cub], phi_M[j_cub + 1], phi_enh_d - dphi_e, phi_enh_d, phi_enh_d + dphi_e]
        phi2 = lagrange_interp_6pt(nodes, vals, np.arange(N_cub))
    print("   Meridional range of pure Mercator=(", phi1[0], ",", phi1[-2], ") U (", -phi1[-2], ",", -phi1[0], ").")
    print("   Meridional range of cubic interpolation=(", phi2[0], ",", phi2[-2], ") U (", -phi2[-2], ",", -phi2[0], ").")
    phi3 = np.concatenate((phi1[0:-1], phi2))
    phi4 = np.linspace(phi3[-1], 0, int(N_enh))
    print("   Meridional range of enhanced resolution=(", phi4[0], ",", -phi4[0], ").")
    print("   Meridional value of enhanced resolution=", phi4[1] - phi4[0])
    phi5 = np.concatenate((phi3[0:-1], phi4))
    sym = np.concatenate((phi5[0:-1], -phi5[::-1]))
    j_phi_n = np.where(sym < phi_n)[0][-1]
    return sym[0:j_phi_n]


def mercator_axis(Ni, phi_s, phi_n, refineR, shift_equator_to_u_point=True, ensure_nj_even=True, enhanced_equatorial=0,
                  return_y_star=False):
    """1-D latitude axis of the Mercator sub-grid (OGG:316-428): device kernels for y* and phi, host parity logic."""
    print("Requesting Mercator grid with phi range: phi_s,phi_n=", phi_s, phi_n)
    y_star = y_mercator_rounded(Ni, np.array([phi_s * PI_180, phi_n * PI_180]))
    print("   y*=", y_star, "nj=", y_star[1] - y_star[0] + 1)
    if y_star[0] % 2 == 0:
        print("  *Equator may not be a u-point!")
        if shift_equator_to_u_point:
            print("  *Fixing this by shifting the bounds!")
            y_star[0] = y_star[0] - 1
            y_star[1] = y_star[1] - 1
            print("   y*=", y_star, "nj=", y_star[1] - y_star[0] + 1)
    if (y_star[1] - y_star[0] + 1) % 2 == 0:
        print("  *Supergrid has an odd number of area cells!")
        if ensure_nj_even:
            print("  *Fixing this by shifting the y_star[1] ")
            y_star[1] = y_star[1] - 1
    print("   Generating Mercator grid with phi range: phi_s,phi_n=", phi_mercator(Ni, y_star))
    phi_M = phi_mercator(Ni, np.arange(y_star[0], y_star[1] + 1))
    equator_index = np.searchsorted(phi_M, 0.0)
    if equator_index == 0:
        raise Exception("   Ooops: Equator is not in the grid")
    print("   Equator is at j=", equator_index)
    if equator_index % 2 == 0:
        print("  *Equator is not going to be a u-point of this grid patch.")
    if enhanced_equatorial:
        phi_M = _enhance_equator(phi_M, phi_n, refineR, enhanced_equatorial)
    if return_y_star:
        return phi_M, y_star
    return phi_M


def generate_mercator_grid(Ni, phi_s, phi_n, lon0_M, lenlon_M, refineR, shift_equator_to_u_point=True, ensure_nj_even=True,
                           enhanced_equatorial=0):
    """OGG:314-441."""
    Ni = int(Ni)
    phi_M = mercator_axis(Ni, phi_s, phi_n, refineR, shift_equator_to_u_point, ensure_nj_even, enhanced_equatorial)
    if phi_M.shape[0] % 2 == 0 and ensure_nj_even:
        print("   The number of j's is not even. Fixing this by cutting one row at south.")
        phi_M = phi_M[1:]
    k = np.arange(Ni + 1, dtype=np.float64)
    lam_M = _new(Ni + 1)
    L.call("ogg_affine_index", Ni + 1, L.ptr(k), float(lon0_M), float(lenlon_M), float(Ni), L.ptr(lam_M))
    phi_M = _f64(phi_M)
    x, y = _new(phi_M.size, Ni + 1), _new(phi_M.size, Ni + 1)
    L.call("ogg_tile_latlon", phi_M.size, Ni + 1, L.ptr(phi_M), L.ptr(lam_M), L.ptr(x), L.ptr(y))
    print("   Final Mercator grid range=", y[0, 0], y[-1, 0])
    print("   number of js=", y.shape[0])
    return x, y


# ----------------------------------------------------------------------------------------------------------------
# displaced pole cap (OGG:447-601)
# ----------------------------------------------------------------------------------------------------------------
def displacedPoleCap_projection(lon_grid, lat_grid, z_0, r_joint):
    """OGG:447-467 on explicit 2-D grids."""
    lon_grid, lat_grid = _f64(lon_grid), _f64(lat_grid)
    nj, ni = lon_grid.shape
    lam, phi = _new(nj, ni), _new(nj, ni)
    z_0 = complex(z_0)
    L.call("ogg_displaced_pole_projection", nj, ni, L.ptr(lon_grid), L.ptr(lat_grid), z_0.real, z_0.imag, float(r_joint),
           float(lon_grid[0, 0]), L.ptr(lam), L.ptr(phi))
    return lam, phi


def monotonic_bounding(x, x_0):
    """OGG:470-475, in place."""
    buf = _f64(x)
    nj, ni = buf.shape
    L.call("ogg_monotonic_bounding", nj, ni, L.ptr(buf), float(x_0))
    if buf is not x:
        x[...] = buf
    return x


def displacedPoleCap_baseGrid(i, j, ni, nj, lon0, lat0):
    """OGG:478-485."""
    i, j = _f64(np.atleast_1d(i)), _f64(np.atleast_1d(j))
    u, v = _new(i.size), _new(j.size)
    L.call("ogg_affine_index", i.size, L.ptr(i), float(lon0), 360.0, float(ni), L.ptr(u))
    L.call("ogg_affine_index", j.size, L.ptr(j), -90.0, float(lat0) - (-90.0), float(nj), L.ptr(v))
    du = np.roll(u, shift=-1, axis=0) - u
    dv = np.roll(v, shift=-1, axis=0) - v
    return u, v, du, dv


def displacedPoleCap_mesh(i, j, ni, nj, lon0, lat0, lam_pole, r_pole, excluded_fraction=None):
    """OGG:488-506."""
    i, j = _f64(np.atleast_1d(i)), _f64(np.atleast_1d(j))
    lams, phis = _new(j.size, i.size), _new(j.size, i.size)
    L.call("ogg_displaced_pole_mesh", i.size, L.ptr(i), j.size, L.ptr(j), int(ni), int(nj), float(lon0), float(lat0),
           float(lam_pole), float(r_pole), L.ptr(lams), L.ptr(phis))
    londp, latdp = lams[0, 0], phis[0, 0]
    if excluded_fraction is not None:
        jmin = np.ceil(excluded_fraction * lams.shape[0])
        jmint = int(jmin + np.mod(jmin, 2))
        return lams[jmint:, :], phis[jmint:, :], londp, latdp
    return lams, phis, londp, latdp


def generate_displaced_pole_grid(Ni, Nj_scap, lon0, lat0, lon_dp, r_dp):
    """OGG:509-518."""
    print("Generating displaced pole grid bounded at latitude ", lat0)
    print("   requested displaced pole lon,rdp=", lon_dp, r_dp)
    x, y, londp, latdp = displacedPoleCap_mesh(np.arange(Ni + 1), np.arange(Nj_scap + 1), Ni, Nj_scap, lon0, lat0, lon_dp, r_dp)
    print("   generated displaced pole lon,lat=", londp, latdp)
    return x, y, londp, latdp


def great_arc_distance(j0, i0, j1, i1, nx, ny, lon0, lat0, lon_dp, r_dp):
    """OGG:522-532."""
    lam0, phi0, _, _ = displacedPoleCap_mesh(i0, j0, nx, ny, lon0, lat0, lon_dp, r_dp)
    lam1, phi1, _, _ = displacedPoleCap_mesh(i1, j1, nx, ny, lon0, lat0, lon_dp, r_dp)
    out = _new(*lam0.shape)
    L.call("ogg_haversine", lam0.size, L.ptr(lam0), L.ptr(phi0), L.ptr(lam1), L.ptr(phi1), L.ptr(out))
    return out


def _numerical_h(j, i, nx, ny, lon0, lat0, lon_dp, r_dp, eps, order, which):
    if order not in (2, 4, 6):
        raise Exception("order not coded")
    i, j = _f64(np.atleast_1d(i)), _f64(np.atleast_1d(j))
    out = _new(j.size, i.size)
    hi, hj = (out, None) if which == "i" else (None, out)
    L.call("ogg_displaced_pole_numerical_h", i.size, L.ptr(i), j.size, L.ptr(j), int(nx), int(ny), float(lon0), float(lat0),
           float(lon_dp), float(r_dp), float(eps), int(order), L.ptr(hi), L.ptr(hj))
    return out


def numerical_hi(j, i, nx, ny, lon0, lat0, lon_dp, r_dp, eps, order=6):
    """OGG:535-547."""
    return _numerical_h(j, i, nx, ny, lon0, lat0, lon_dp, r_dp, eps, order, "i")


def numerical_hj(j, i, nx, ny, lon0, lat0, lon_dp, r_dp, eps, order=6):
    """OGG:550-562."""
    return _numerical_h(j, i, nx, ny, lon0, lat0, lon_dp, r_dp, eps, order, "j")


def default_dp_arc():
    """Arc form of the displaced-pole quadrature where the caller names none: OGG_DP_ARC, else "chord" (closer to the exact value of
    the reference's formula than the fp64 reference itself; "literal" = the reference's operation sequence)."""
    arc = os.environ.get("OGG_DP_ARC", "chord")
    if arc not in ("chord", "literal"):
        raise ValueError("OGG_DP_ARC must be chord or literal, not %r" % arc)
    return arc


def displacedPoleCap_metrics_quad(order, nx, ny, lon0, lat0, lon_dp, r_dp, Re=_default_Re, arc_form=None, symmetry=None):
    """OGG:565-601 (kernels dpole_quad_tables / dpole_quad_kernel: the lattice is evaluated and reduced on chip).
    ``arc_form`` (not in the reference): how the great-arc distance of two probes of the finite-difference stencil is taken.
    "chord" (the default, also OGG_DP_ARC): from the probes' positions on the sphere (cross product; no longitude, no unwrap);
    "literal": the reference's great_arc_distance operation for operation (haversine of projected longitudes and latitudes).
    Same stencil, same quadrature.  Against the exact value of the reference's own formula (50-digit evaluation,
    tests/golden/truth_table.npz, 1/8 degree cap of BASELINE config 4) the fp64 reference itself is 1.4e-9 / 9.0e-10 / 1.2e-9
    (dx / dy / area, max relative) away, the literal form 1.4e-9 / 8.5e-10 / 1.2e-9 and the chord form 7.6e-10 / 2.7e-10 / 8.3e-10
    (profiles/r04_truth_table.json): the chord form is the more accurate one and ~6x cheaper, hence the default.
    ``symmetry`` (not in the reference): see _sym; mirrored (chord form, the displaced pole's meridian on a node column), the half of the
    columns on one side of that meridian is evaluated and written to its mirror images (csrc/ogg_dpole_dev.h, DpQuadParams)."""
    print("   Calculating displaced pole cap metrics via quadrature ...")
    nx, ny = int(nx), int(ny)
    form = {"literal": L.DP_ARC_LITERAL, "chord": L.DP_ARC_CHORD}[arc_form or default_dp_arc()]
    dxq, dyq, daq = _new(ny + 1, nx), _new(ny, nx + 1), _new(ny, nx)
    L.call("ogg_displaced_pole_metrics_quad_form_sym", form, _sym(symmetry), int(order), nx, ny, float(lon0), float(lat0), float(lon_dp),
           float(r_dp), float(Re), L.ptr(dxq), L.ptr(dyq), L.ptr(daq))
    return dxq, dyq, daq


# ----------------------------------------------------------------------------------------------------------------
# stencil metrics and angle (OGG:682-770)
# ----------------------------------------------------------------------------------------------------------------
def mdist(x1, x2):
    """OGG:682-684."""
    a, b = np.broadcast_arrays(_f64(x1), _f64(x2))
    a, b = np.ascontiguousarray(a), np.ascontiguousarray(b)
    out = _new(*a.shape)
    L.call("ogg_mdist", a.size, L.ptr(a), L.ptr(b), L.ptr(out))
    return out


def generate_grid_metrics_MIDAS(x, y, axis_units="degrees", Re=_default_Re, latlon_areafix=True):
    """OGG:687-716 (kernel midas_tile_kernel: an LDS-staged tile walk, csrc/ogg_midas.hip)."""
    x, y = _f64(x), _f64(y)
    if x.shape != y.shape:
        raise Exception("Input arrays do not have the same shape!")
    nj1, ni1 = x.shape
    if nj1 == 1 and ni1 >= 2:
        # a single row of points: numpy's slices give dx (1, ni1-1) and EMPTY dy, area.  dx of a row depends on that row alone, so the
        # kernel runs on the row taken twice and its first dx row is the answer (the library's own entry wants two rows of points)
        dx2, _, _ = generate_grid_metrics_MIDAS(np.concatenate((x, x), axis=0), np.concatenate((y, y), axis=0), axis_units, Re, latlon_areafix)
        return np.ascontiguousarray(dx2[:1]), _new(0, ni1), _new(0, ni1 - 1)
    dx, dy, area = _new(nj1, ni1 - 1), _new(nj1 - 1, ni1), _new(nj1 - 1, ni1 - 1)
    L.call("ogg_grid_metrics_midas", nj1, ni1, L.ptr(x), L.ptr(y), float(Re), int(bool(latlon_areafix)), L.ptr(dx), L.ptr(dy),
           L.ptr(area))
    return dx, dy, area


def angle_x(x, y):
    """OGG:719-729 (same kernel, angle-only instantiation)."""
    x, y = _f64(x), _f64(y)
    if x.shape != y.shape:
        raise Exception("Input arrays do not have the same shape!")
    nj1, ni1 = x.shape
    out = _new(nj1, ni1)
    L.call("ogg_angle_x", nj1, ni1, L.ptr(x), L.ptr(y), L.ptr(out))
    return out


def metrics_error_from_sums(area_sum, dy_col_a, dy_col_b, dx_first_row, dx_last_row, lat1, lat2=90, Re=_default_Re, bipolar=False,
                            displaced_pole=-999, excluded_fraction=None):
    """OGG:735-770 given the five sums it takes over the fields: sum(area), sum(dy[:, Ni//4]) (or, for a displaced pole,
    sum(dy[:, pole]) and sum(dy[:, antipole])), sum(dx[0, :]), sum(dx[-1, :]).  ``metrics_error`` forms them on the host from
    arrays; ``supergrid.Supergrid.metrics_error`` forms them on the device per band and adds them over the ranks."""
    exact_area = 2 * np.pi * (Re ** 2) * np.abs(np.sin(lat2 * PI_180) - np.sin(lat1 * PI_180))
    exact_lat_arc_length = np.abs(lat2 - lat1) * PI_180 * Re
    exact_lon_arc_length = np.cos(lat1 * PI_180) * 2 * np.pi * Re
    grid_lat_arc_length = dy_col_a
    grid_lon_arc_length = dx_first_row
    if lat1 > lat2:
        grid_lon_arc_length = dx_last_row
    if bipolar:
        lon_arc2_error = 100 * (dx_last_row / 4 - exact_lat_arc_length) / exact_lat_arc_length
    area_error = 100 * (area_sum - exact_area) / exact_area
    lat_arc_error = 100 * (grid_lat_arc_length - exact_lat_arc_length) / exact_lat_arc_length
    lon_arc_error = 100 * (grid_lon_arc_length - exact_lon_arc_length) / exact_lon_arc_length
    if displaced_pole != -999:
        grid_lat_arc_length = dy_col_a + dy_col_b
        lat_arc_error = 100 * (grid_lat_arc_length - 2.0 * exact_lat_arc_length) / exact_lat_arc_length
    if excluded_fraction:
        print("   Cannot estimate area and dy accuracies with excluded_fraction (doughnut)! ")
    if bipolar:
        return area_error, lat_arc_error, lon_arc_error, lon_arc2_error
    return area_error, lat_arc_error, lon_arc_error


def metrics_error_columns(Ni, displaced_pole=-999):
    """The dy columns OGG:739 / 755-760 sum: (Ni//4, none), or (pole, antipole) for a displaced pole."""
    if displaced_pole == -999:
        return Ni // 4, -1
    antipole = displaced_pole + Ni // 2
    if displaced_pole > Ni // 2:
        antipole = displaced_pole - Ni // 2
    return displaced_pole, antipole


def metrics_error(dx_, dy_, area_, Ni, lat1, lat2=90, Re=_default_Re, bipolar=False, displaced_pole=-999,
                  excluded_fraction=None):
    """Self-check of OGG:732-770: a few host reductions over device-produced fields, compared with the sphere."""
    col_a, col_b = metrics_error_columns(Ni, displaced_pole)
    return metrics_error_from_sums(np.sum(area_), np.sum(dy_[:, col_a]), np.sum(dy_[:, col_b]) if col_b >= 0 else 0.0,
                                   np.sum(dx_[0, :]), np.sum(dx_[-1, :]), lat1, lat2, Re, bipolar, displaced_pole, excluded_fraction)


# ----------------------------------------------------------------------------------------------------------------
# output (host; OGG:773-829).  NetCDF-3 64-bit-offset through scipy (netCDF4-python is not a dependency here).
# ----------------------------------------------------------------------------------------------------------------
def write_nc(x, y, dx, dy, area, angle_dx, axis_units="degrees", fnam=None, format="NETCDF3_64BIT", description=None,
             history=None, source=None, no_changing_meta=None, debug=False):
    """The reference's output layout (OGG:795-826): dimensions nyp, nxp, ny, nx, string(255); variables tile, y, x, dy,
    dx, area, angle_dx in that order; NetCDF-3 64-bit-offset.  Written by netcdf3.Dataset (no netCDF4 dependency)."""
    from . import netcdf3

    if fnam is None:
        fnam = "supergrid.nc"
    if format != "NETCDF3_64BIT":
        raise Exception("only format NETCDF3_64BIT is supported")
    if debug:
        for a, lbl in ((x, "x"), (y, "y"), (dx, "dx"), (dy, "dy"), (area, "area"), (angle_dx, "angle_dx")):
            chksum(a, lbl)
    ny, nx = area.shape
    print("   Writing netcdf file with ny,nx= ", ny, nx)
    gatts = []
    if not no_changing_meta:
        gatts = [("history", history or ""), ("description", description or ""), ("source", source or "")]
    ds = netcdf3.Dataset(str(fnam), [("nyp", ny + 1), ("nxp", nx + 1), ("ny", ny), ("nx", nx), ("string", 255)], gatts)
    ds.def_var("tile", netcdf3.NC_CHAR, ("string",), [], np.frombuffer(b"tile1".ljust(255, b"\0"), dtype="S1"))
    ds.def_var("y", netcdf3.NC_DOUBLE, ("nyp", "nxp"), [("units", "degrees")], y)
    ds.def_var("x", netcdf3.NC_DOUBLE, ("nyp", "nxp"), [("units", "degrees")], x)
    ds.def_var("dy", netcdf3.NC_DOUBLE, ("ny", "nxp"), [("units", "meters")], dy)
    ds.def_var("dx", netcdf3.NC_DOUBLE, ("nyp", "nx"), [("units", "meters")], dx)
    ds.def_var("area", netcdf3.NC_DOUBLE, ("ny", "nx"), [("units", "m2")], area)
    ds.def_var("angle_dx", netcdf3.NC_DOUBLE, ("nyp", "nxp"), [("units", "degrees")], angle_dx)
    ds.write()


def generate_latlon_grid(lni, lnj, llon0, llen_lon, llat0, llen_lat, ensure_nj_even=True):
    """OGG:832-846."""
    print("Generating regular lat-lon grid between latitudes ", llat0, llat0 + llen_lat)
    lni, lnj = int(lni), int(lnj)
    skip = 1 if ((lnj + 1) % 2 == 0 and ensure_nj_even) else 0
    if skip:
        print("   The number of j's is not even. Fixing this by cutting one row at south.")
    x, y = _new(lnj + 1 - skip, lni + 1), _new(lnj + 1 - skip, lni + 1)
    L.call("ogg_generate_latlon_grid", lni, lnj, float(llon0), float(llen_lon), float(llat0), float(llen_lat), skip, L.ptr(x),
           L.ptr(y))
    print("   generated regular lat-lon grid between latitudes ", y[0, 0], y[-1, 0])
    print("   number of js=", y.shape[0])
    return x, y


def usage():
    print("ocean_grid_generator.py -f <output_grid_filename> -r <inverse_degrees_resolution> [--r_dp=<displacement_factor/0.2> "
          "--exfracdp=0.5 --south_cutoff_ang=<degrees_south_to_start> --south_cutoff_row=<rows_south_to_cut> --match_dy bp so "
          "--ensure_nj_even --plotem --write_subgrid_files --enhanced_equatorial=<n> --skip_metrics --grids sc]")


# ----------------------------------------------------------------------------------------------------------------
# orchestrator (host; OGG:855-1449)
# ----------------------------------------------------------------------------------------------------------------
def _minus_ones(lam):
    nj1, ni1 = lam.shape
    return -np.ones([nj1, ni1 - 1]), -np.ones([nj1 - 1, ni1]), -np.ones([nj1 - 1, ni1 - 1])


def _script_metadata():
    """git hash / modification state of this script, as OGG:910-944 records them."""
    import socket

    host = str(socket.gethostname())
    scriptpath = sys.argv[0]
    dirname = os.path.dirname(scriptpath) or "."
    basename = os.path.basename(scriptpath)

    def sh(cmd):
        try:
            return subprocess.check_output(cmd, stderr=subprocess.STDOUT, shell=True).decode("ascii", "replace").rstrip("\n")
        except Exception:
            return ""

    githash = sh("cd " + dirname + ";git rev-parse HEAD; exit 0")
    gitmod = sh("cd " + dirname + ";git status --porcelain " + basename + " | awk '{print $1}' ; exit 0")
    if "M" in str(gitmod):
        gitmod = " , But was localy Modified!"
    return host, scriptpath, githash, gitmod


def _validate_flags(match_dy, r_dp, lat_dp):
    """OGG:881-888, 904-907"""
    known_options = ["bp", "so", "p125sc", ""]
    unknown = list(set(match_dy).difference(known_options))
    if len(unknown) != 0:
        print("Unknown options in match_dy: ", unknown)
        print("Known options are one or more of ", known_options)
        sys.exit(2)
    if r_dp != 0.0 and lat_dp > -90.0:
        print("Cannot specify both --rdp and --latdp for the displaced pole!")
        usage()
        sys.exit(2)


def _meta_strings(inverse_resolution, no_changing_meta):
    """history / source / description heads of OGG:946-963"""
    hist = "This grid file was generated via command " + " ".join(sys.argv)
    source = ""
    if not no_changing_meta:
        host, scriptpath, githash, gitmod = _script_metadata()
        hist = hist + " on " + str(datetime.date.today()) + " on platform " + host
        source = scriptpath + " had git hash " + githash + gitmod
        source = source + ". To obtain the grid generating code do: git clone  https://github.com/nikizadehgfdl/grid_generation.git ; cd grid_generation;  git checkout " + githash
    desc = ("This is an orthogonal coordinate grid for the Earth with a nominal resoution of " + str(1 / inverse_resolution)
            + " degrees along the equator. ")
    return hist, source, desc


def main(inverse_resolution, gridfilename="ocean_hgrid.nc", r_dp=0.0, lon_dp=80.0, lat_dp=-99.0, exfracdp=None,
         south_cutoff_row=0, south_cutoff_ang=-90.0, reproduce_MIDAS_grids=False, write_subgrid_files=False, plotem=False,
         no_changing_meta=False, enhanced_equatorial=0, debug=False, grids="all", match_dy=(), skip_metrics=False,
         ensure_nj_even=False, shift_equator_to_u_point=True, bipolar_lower_lat=-99.0, mercator_lower_lat=-99.0,
         mercator_upper_lat=-99.0, south_ocean_lower_lat=-99.0, south_ocean_upper_lat=-99.0, no_south_cap=False,
         return_arrays=False, path=None, dp_arc=None, cap_symmetry=None):
    """Build the supergrid and write it.  Same flags as the reference's main() (OGG:855-1449); the defaults of ``grids`` and
    ``match_dy`` are the argparse defaults (the reference's own function defaults fail its own validation, OGG:870-888).

    The sub-grid loop runs as ONE device-resident pass (supergrid.SupergridPlan + ogg_supergrid_pass_dev: three or four launches
    for the whole grid, all six fields of every sub-grid left in HBM); the fields then go from HBM straight into the NetCDF file
    (nc_stream: byte-swap-on-copy into a pinned ring, pwrite at each variable's offset) -- no per-function staging, no host
    stitching.  ``path="functions"`` (or OGG_MAIN_PATH=functions) runs the reference's own sequence of calls instead, one host-array
    function after the other (main_function_level): the same bits, for checking.  ``dp_arc`` (or OGG_DP_ARC): arc form of the
    displaced-pole quadrature, "chord" (default; DESIGN.md section 2: closer to the exact value of the reference's formula than the fp64
    reference is) or "literal" (the reference's operation sequence).  ``cap_symmetry`` (or OGG_CAP_SYMMETRY=0): None / True: the caps
    are evaluated on the columns that determine the rest and mirrored (_sym; DESIGN.md section 2: a mirrored value is as far from the exact
    value of the reference's formula as the reference's own), False: every column, as the reference does.  ``return_arrays=True``
    additionally returns the six stitched fields and the sub-grid pieces (used by tests)."""
    import time

    path = path or os.environ.get("OGG_MAIN_PATH", "pass")
    dp_arc = dp_arc or default_dp_arc()
    if path == "functions":
        return main_function_level(inverse_resolution, gridfilename, r_dp, lon_dp, lat_dp, exfracdp, south_cutoff_row, south_cutoff_ang,
                                   reproduce_MIDAS_grids, write_subgrid_files, plotem, no_changing_meta, enhanced_equatorial, debug, grids,
                                   match_dy, skip_metrics, ensure_nj_even, shift_equator_to_u_point, bipolar_lower_lat, mercator_lower_lat,
                                   mercator_upper_lat, south_ocean_lower_lat, south_ocean_upper_lat, no_south_cap, return_arrays, dp_arc,
                                   cap_symmetry)
    from . import supergrid as SG

    _validate_flags(match_dy, r_dp, lat_dp)
    hist, source, desc = _meta_strings(inverse_resolution, no_changing_meta)
    start_time = time.time()
    plan = SG.SupergridPlan(inverse_resolution, r_dp=r_dp, lon_dp=lon_dp, lat_dp=lat_dp, exfracdp=exfracdp, south_cutoff_row=south_cutoff_row,
                            south_cutoff_ang=south_cutoff_ang, skip_metrics=skip_metrics, ensure_nj_even=ensure_nj_even,
                            no_south_cap=no_south_cap, enhanced_equatorial=enhanced_equatorial, match_dy=match_dy, grids=grids,
                            shift_equator_to_u_point=shift_equator_to_u_point, bipolar_lower_lat=bipolar_lower_lat,
                            mercator_lower_lat=mercator_lower_lat, mercator_upper_lat=mercator_upper_lat,
                            south_ocean_lower_lat=south_ocean_lower_lat, dp_arc=dp_arc, cap_symmetry=cap_symmetry)
    import torch
    g = SG.Supergrid(plan, device="cuda:%d" % torch.cuda.current_device())
    for s in plan.subs:
        print("Generating %s sub-grid: %d rows x %d columns" % (s.name, s.nj1, plan.Ni + 1))
    g.run_pass()
    g.check_lookback_flags()
    if not skip_metrics:   # the reference's CHECK_metrics lines (OGG:1017,1070,1112,1157,1172) from sums taken on the device
        labels = {"Merc": "CHECK_metrics: % errors in (area, lat arc, lon arc)", "BP": "CHECK_metrics_hquad: % errors in (area, lat arc, lon arc1, lon arc2)",
                  "SO": "CHECK_metrics_MIDAS: % errors in (area, lat arc, lon arc)"}
        errs = g.metrics_error()
        for name in ("Merc", "BP", "SO", "SC"):
            if name in errs:
                sc_dp = name == "SC" and plan.subs[0].kind == "dpole"
                print("   " + labels.get(name, "CHECK_metrics_hquad: % errors in (area, lat arc, lon arc)" if sc_dp else labels["SO"]), errs[name])
    # south cuts (OGG:1268-1313) and the final guards (OGG:1371-1375, 1425-1436) need two columns of y only
    cut = g.south_cut()
    print("Stitching the grids together...")
    SG.check_guards(g.stitched_column("y", plan.Ni // 4, cut), any(s.name == "BP" for s in plan.subs))
    names = [s.name for s in plan.subs if not (s.name == "SC" and cut[2])]
    desc = desc + "It consists of; "
    merc = next((s for s in plan.subs if s.name == "Merc"), None)
    if merc is not None:
        ym = g.stitched_column("y", 0, (0, 0, False), only="Merc")
        desc = desc + "a Mercator grid spanning " + str(ym[0]) + " to " + str(ym[-1]) + " degrees; "
        if "BP" in names:
            desc = desc + "a bipolar northern cap north of " + str(ym[-1]) + " degrees; "
    if "SO" in names:
        desc = desc + "a regular lat-lon grid spanning " + str(plan.latUp_SO) + " to " + str(plan.lat0_SO) + " degrees; "
    if "SC" in names:
        desc = desc + "a " + ("displaced pole " if r_dp != 0.0 else "regular ") + "southern cap south of " + str(plan.lat0_SO) + " degrees."
    if south_cutoff_ang > -90:
        desc = desc + " It is cut south of " + str(south_cutoff_ang) + " degrees."
    if south_cutoff_row > 0:
        desc = desc + " The first " + str(south_cutoff_row) + " rows at south are deleted."
    nyp = g.stitched_rows(cut)
    print("shapes: ", (nyp, plan.Ni + 1), (nyp, plan.Ni + 1), (nyp, plan.Ni), (nyp - 1, plan.Ni + 1), (nyp - 1, plan.Ni), (nyp, plan.Ni + 1))
    out = None
    if return_arrays or write_subgrid_files or debug or plotem:
        out = SG.stitch(plan, [g.bands_to_host()])
    if write_subgrid_files:
        for name, piece in out["sub"].items():
            write_nc(piece["x"], piece["y"], piece["dx"], piece["dy"], piece["area"], piece["angle_dx"], axis_units="degrees",
                     fnam=gridfilename + name + ".nc", description=desc, history=hist, source=source, debug=debug)
        if "SC" not in out["sub"] and plan.Nj_scap != 0:
            print("There remained no South Pole cap grid because of the number of rows cut= ", cut)
    if gridfilename is not None:
        if debug:
            for lbl in ("x", "y", "dx", "dy", "area", "angle_dx"):
                chksum(out[lbl], lbl)
        g.write_nc(str(gridfilename), cut, description=desc, history=hist, source=source, no_changing_meta=no_changing_meta)
        print("Wrote the whole grid to file ", gridfilename)
    if plotem:   # OGG:1230-1234, 1445-1447, from the stitched host arrays (raises if matplotlib is missing: never ignored silently)
        _plot_grids(out, r_dp, lon_dp, lat_dp, plan.lon0, plan.lat0_SO, inverse_resolution, out["x"], out["y"])
    print("runtime(secs)  %s" % (time.time() - start_time))
    if return_arrays:
        return out


def _plot_grids(out, r_dp, lon_dp, lat_dp, lon0, lat0_SC, refineR, x, y):
    """--plotem (OGG:1230-1234, 1445-1447): the displaced-pole cap (with the Southern Ocean piece) on polar axes when there is one, then the
    whole grid south of -40 and north of 40 degrees."""
    sub = out.get("sub", {}) if isinstance(out, dict) else {}
    if "SC" in sub and (r_dp != 0.0 or lat_dp > -90):
        sc = sub["SC"]
        ax = displacedPoleCap_plot(_sub_xy(sc)[0], _sub_xy(sc)[1], lon0, lon_dp, lat0_SC, stride=int(refineR * 10), block=True, dplat=lat_dp)
        if "SO" in sub:
            plot_mesh_in_latlon(_sub_xy(sub["SO"])[0], _sub_xy(sub["SO"])[1], stride=int(refineR * 10), newfig=False, axis=ax)
    plot_mesh_in_xyz(x, y, stride=30, upperlat=-40, title="Grid south of -40 degrees")
    plot_mesh_in_xyz(x, y, stride=30, lowerlat=40, title="Grid north of 40 degrees")


def _sub_xy(piece):
    """(x, y) of a sub-grid record: a dict of fields (pass path) or a tuple starting with x, y (function-level path)."""
    return (piece["x"], piece["y"]) if isinstance(piece, dict) else (piece[0], piece[1])


def main_function_level(inverse_resolution, gridfilename="ocean_hgrid.nc", r_dp=0.0, lon_dp=80.0, lat_dp=-99.0, exfracdp=None,
                        south_cutoff_row=0, south_cutoff_ang=-90.0, reproduce_MIDAS_grids=False, write_subgrid_files=False, plotem=False,
                        no_changing_meta=False, enhanced_equatorial=0, debug=False, grids="all", match_dy=(), skip_metrics=False,
                        ensure_nj_even=False, shift_equator_to_u_point=True, bipolar_lower_lat=-99.0, mercator_lower_lat=-99.0,
                        mercator_upper_lat=-99.0, south_ocean_lower_lat=-99.0, south_ocean_upper_lat=-99.0, no_south_cap=False,
                        return_arrays=False, dp_arc=None, cap_symmetry=None):
    """The reference's own sequence of calls (OGG:855-1449), every callee a host-array function of this module (numpy in, numpy
    out, one staged device call each) and the stitching on the host: what a user gets who swaps the reference's module for this
    one function by function.  main() produces the same bits from one device-resident pass."""
    import time

    known_options = ["bp", "so", "p125sc", ""]
    unknown = list(set(match_dy).difference(known_options))
    if len(unknown) != 0:
        print("Unknown options in match_dy: ", unknown)
        print("Known options are one or more of ", known_options)
        sys.exit(2)
    doughnut = 0.28 * 7 / 4
    doughnut = exfracdp if (exfracdp is not None) else doughnut
    calculate_metrics = not skip_metrics
    if r_dp != 0.0 and lat_dp > -90.0:
        print("Cannot specify both --rdp and --latdp for the displaced pole!")
        usage()
        sys.exit(2)

    hist = "This grid file was generated via command " + " ".join(sys.argv)
    source = ""
    if not no_changing_meta:
        host, scriptpath, githash, gitmod = _script_metadata()
        hist = hist + " on " + str(datetime.date.today()) + " on platform " + host
        source = scriptpath + " had git hash " + githash + gitmod
        source = source + ". To obtain the grid generating code do: git clone  https://github.com/nikizadehgfdl/grid_generation.git ; cd grid_generation;  git checkout " + githash
    desc = ("This is an orthogonal coordinate grid for the Earth with a nominal resoution of " + str(1 / inverse_resolution)
            + " degrees along the equator. ")

    start_time = time.time()
    refineS = 2  # supergrid
    refineR = inverse_resolution
    lenlon = 360
    lon0 = -300.0
    Ni = int(refineR * refineS * lenlon)
    q = Ni // 4  # symmetry-meridian column

    def want(tok):
        return (tok in grids) or ("all" in grids)

    def nc(sub, tag):
        if write_subgrid_files:
            write_nc(sub[0], sub[1], sub[2], sub[3], sub[4], sub[5], axis_units="degrees", fnam=gridfilename + tag + ".nc",
                     description=desc, history=hist, source=source, debug=debug)

    # ---- Mercator (OGG:987-1030)
    phi_s_Merc, phi_n_Merc = -66.85954725, 64.05895973
    if mercator_upper_lat > -90:
        phi_n_Merc = mercator_upper_lat
    if mercator_lower_lat > -90:
        phi_s_Merc = mercator_lower_lat
    if refineR == 2:
        phi_s_Merc, phi_n_Merc = -68.0, 65.0
    if refineR == 1 and enhanced_equatorial:
        phi_s_Merc, phi_n_Merc = -77.8, 60.0
    sub = {}
    if want("mercator"):
        lamMerc, phiMerc = generate_mercator_grid(Ni, phi_s_Merc, phi_n_Merc, lon0, lenlon, refineR,
                                                  shift_equator_to_u_point=shift_equator_to_u_point,
                                                  ensure_nj_even=ensure_nj_even, enhanced_equatorial=enhanced_equatorial)
        angleMerc = angle_x(lamMerc, phiMerc)
        dxMerc, dyMerc, areaMerc = _minus_ones(lamMerc)
        if calculate_metrics:
            dxMerc, dyMerc, areaMerc = generate_grid_metrics_MIDAS(lamMerc, phiMerc)
            print("   CHECK_metrics: % errors in (area, lat arc, lon arc)",
                  metrics_error(dxMerc, dyMerc, areaMerc, Ni, phiMerc[0, 0], phiMerc[-1, 0]))
        sub["Merc"] = [lamMerc, phiMerc, dxMerc, dyMerc, areaMerc, angleMerc]
        nc(sub["Merc"], "Merc")
        DeltaPhiMerc_so = phiMerc[1, q] - phiMerc[0, q]
        DeltaPhiMerc_no = phiMerc[-1, q] - phiMerc[-2, q]
        lat0_bp = phiMerc[-1, q]

    # ---- bipolar cap (OGG:1035-1075)
    if want("bipolar"):
        lon_bp = lon0
        if bipolar_lower_lat > -90:
            lat0_bp = bipolar_lower_lat
        Nj_ncap = int(60 * refineR * refineS)
        if refineR == 2:
            Nj_ncap = 119 * refineS
        if refineR == 1 and enhanced_equatorial:
            Nj_ncap = 154
        if "bp" in match_dy:
            print("   Match dy at bipolar cap joint")
            Nj_ncap = int(0.5 + (90.0 - lat0_bp) / DeltaPhiMerc_no)
        lamBP, phiBP, _, _ = generate_bipolar_cap_mesh(Ni, Nj_ncap, lat0_bp, lon_bp, ensure_nj_even=ensure_nj_even, symmetry=cap_symmetry)
        rp = np.tan(0.5 * (90 - lat0_bp) * PI_180)
        dxBP, dyBP, areaBP = _minus_ones(lamBP)
        if calculate_metrics:
            dxBP, dyBP, areaBP = bipolar_cap_metrics_quad_fast(5, phiBP.shape[1] - 1, phiBP.shape[0] - 1, lat0_bp, lon_bp, rp, symmetry=cap_symmetry)
            print("   CHECK_metrics_hquad: % errors in (area, lat arc, lon arc1, lon arc2)",
                  metrics_error(dxBP, dyBP, areaBP, Ni, lat0_bp, 90.0, bipolar=True))
        angleBP = angle_x(lamBP, phiBP)
        sub["BP"] = [lamBP, phiBP, dxBP, dyBP, areaBP, angleBP]
        nc(sub["BP"], "BP")

    # ---- Southern Ocean (OGG:1080-1116); like the reference this needs the Mercator sub-grid
    lat0_SO = -78.0
    if south_ocean_lower_lat > -90:
        lat0_SO = south_ocean_lower_lat
    latUp_SO = phiMerc[0, q]
    lenlat_SO = latUp_SO - lat0_SO
    deltaPhiSO = 1.0 / refineR / refineS
    Nj_SO = int(refineR * 55)
    if refineR == 2 and enhanced_equatorial:
        Nj_SO = 109
    if refineR == 1 and enhanced_equatorial:
        Nj_SO = 0
    if "so" in match_dy:
        print("   Match dy at Southern Ocean joint")
        Nj_SO = int(0.5 + lenlat_SO / DeltaPhiMerc_so)
    if (Nj_SO != 0) and want("so"):
        lamSO, phiSO = generate_latlon_grid(Ni, Nj_SO, lon0, lenlon, lat0_SO, lenlat_SO, ensure_nj_even=ensure_nj_even)
        dxSO, dySO, areaSO = _minus_ones(lamSO)
        if calculate_metrics:
            dxSO, dySO, areaSO = generate_grid_metrics_MIDAS(lamSO, phiSO)
        angleSO = angle_x(lamSO, phiSO)
        print("   CHECK_metrics_MIDAS: % errors in (area, lat arc, lon arc)",
              metrics_error(dxSO, dySO, areaSO, Ni, phiSO[0, 0], phiSO[-1, 0]))
        sub["SO"] = [lamSO, phiSO, dxSO, dySO, areaSO, angleSO]
        nc(sub["SO"], "SO")

    # ---- southern cap (OGG:1122-1231)
    lat0_SC = phiSO[0, q]
    if "p125sc" in match_dy:
        print("   Warning: Choose SC latitude to reproduce CM4X grid!")
        lat0_SC = lat0_SO
    Nj_scap = int(refineR * 40) * 7 // 4
    if no_south_cap or (enhanced_equatorial and refineR in (1, 2)):
        Nj_scap = 0
    if (Nj_scap != 0) and want("sc"):
        if r_dp == 0.0 and lat_dp <= -90.0:  # regular pole
            Nj_scap = int((lat0_SC + 90.0) / deltaPhiSO)
            lamSC, phiSC = generate_latlon_grid(Ni, Nj_scap, lon0, lenlon, -90.0, 90 + lat0_SO, ensure_nj_even=ensure_nj_even)
            angleSC = angle_x(lamSC, phiSC)
            dxSC, dySC, areaSC = _minus_ones(lamSC)
            if calculate_metrics:
                dxSC, dySC, areaSC = generate_grid_metrics_MIDAS(lamSC, phiSC)
                print("   CHECK_metrics_MIDAS: % errors in (area, lat arc, lon arc)",
                      metrics_error(dxSC, dySC, areaSC, Ni, phiSC[-1, 0], phiSC[0, 0]))
            pieces = [lamSC, phiSC, dxSC, dySC, areaSC, angleSC]
        else:  # displaced pole
            if lat_dp > -90:
                r_dp = np.tan((90 + lat_dp) * PI_180) / np.tan((90 + lat0_SC) * PI_180)
            lamSC, phiSC, londp, latdp = generate_displaced_pole_grid(Ni, Nj_scap, lon0, lat0_SC, lon_dp, r_dp)
            angleSC = angle_x(lamSC, phiSC)
            dxSC, dySC, areaSC = _minus_ones(lamSC)
            if calculate_metrics:
                dxSC, dySC, areaSC = displacedPoleCap_metrics_quad(4, Ni, Nj_scap, lon0, lat0_SC, lon_dp, r_dp, arc_form=dp_arc, symmetry=cap_symmetry)
                poles_i = int(Ni * np.mod(lon_dp - lon0, 360) / 360.0)
                print("   CHECK_metrics_hquad: % errors in (area, lat arc, lon arc)",
                      metrics_error(dxSC, dySC, areaSC, Ni, lat1=lat0_SC, lat2=-90.0, displaced_pole=poles_i,
                                    excluded_fraction=doughnut))
            pieces = [lamSC, phiSC, dxSC, dySC, areaSC, angleSC]
            if doughnut != 0.0:
                jmin = np.ceil(doughnut * Nj_scap)
                jmint = int(jmin + np.mod(jmin, 2))
                pieces = [p[jmint:, :] for p in pieces]
            if pieces[1].shape[0] % 2 == 0 and ensure_nj_even:
                print("   The number of j's is not even. Fixing this by cutting one row at south.")
                pieces = [p[1:, :] for p in pieces]
            print("   number of js=", pieces[0].shape[0])
        sub["SC"] = pieces
        nc(sub["SC"], "SC")

    # ---- south cuts (OGG:1268-1313)
    cut, jcut = False, 0
    if south_cutoff_row > 0:
        cut, jcut = True, south_cutoff_row - 1
    elif south_cutoff_ang > -90:
        cut, jcut = True, 1 + np.nonzero(sub["SC"][1][:, 0] < south_cutoff_ang)[0][-1]
    sc_rows_before_cut = sub["SC"][0].shape[0] if "SC" in sub else 0
    if cut:
        if "SC" in sub and jcut < sub["SC"][0].shape[0]:
            print("   SC: shape[0], jcut", sub["SC"][0].shape[0], jcut)
            if (sub["SC"][1].shape[0] - jcut) % 2 == 0 and ensure_nj_even:
                print("   SC: The number of j's is not even. Fixing this by cutting one row at south.")
                jcut = jcut + 1
            print("   Cutting SC grid rows 0 to ", jcut)
            sub["SC"] = [p[jcut:, :] for p in sub["SC"]]
        elif "SO" in sub:
            print("   Whole SC and some of SO need to be cut!")
            jcut_SO = jcut - sub["SC"][0].shape[0]
            del sub["SC"]
            if (sub["SO"][4].shape[0] - jcut_SO - 1) % 2 == 0 and ensure_nj_even:
                print("   SO: The number of j's is not even. Fixing this by cutting one row at south.")
                jcut_SO = jcut_SO + 1
            print("   No SC grid remained. Cutting SO grid rows 0 to ", jcut_SO)
            sub["SO"] = [p[jcut_SO:, :] for p in sub["SO"]]

    # ---- stitch south -> north (OGG:1315-1377)
    def join(south, north):
        out = []
        for k, (s, n) in enumerate(zip(south, north)):
            # x, y, dx, angle drop the southern piece's last row; dy, area are concatenated whole
            out.append(np.concatenate((s if k in (3, 4) else s[:-1, :], n), axis=0))
        return out

    print("Stitching the grids together...")
    g = None
    if "SC" in sub and "SO" in sub:
        g = join(sub["SC"], sub["SO"])
    elif "SO" in sub:
        g = list(sub["SO"])
    if "SO" in sub and "Merc" in sub:
        g = join(g, sub["Merc"])
    elif "Merc" in sub:
        g = list(sub["Merc"])
    if "BP" in sub:
        g = join(g, sub["BP"])
        ycol = g[1][:, q]
        if np.any((np.roll(ycol, shift=-1, axis=0) - ycol) == 0):
            raise Exception("lattitude array has repeated values along symmetry meridian!")
    x3, y3, dx3, dy3, area3, angle3 = g

    if write_subgrid_files:
        if "SC" in sub:
            nc(sub["SC"], "SC")
        elif Nj_scap != 0:
            print("There remained no South Pole cap grid because of the number of rows cut= ", jcut, sc_rows_before_cut)

    # ---- description (OGG:1403-1423)
    desc = desc + "It consists of; "
    if "Merc" in sub:
        desc = desc + "a Mercator grid spanning " + str(phiMerc[0, 0]) + " to " + str(phiMerc[-1, 0]) + " degrees; "
    if "BP" in sub:
        desc = desc + "a bipolar northern cap north of " + str(phiMerc[-1, 0]) + " degrees; "
    if "SO" in sub:
        desc = desc + "a regular lat-lon grid spanning " + str(latUp_SO) + " to " + str(lat0_SO) + " degrees; "
    if "SC" in sub:
        desc = desc + "a " + ("displaced pole " if r_dp != 0.0 else "regular ") + "southern cap south of " + str(lat0_SO) + " degrees."
    if south_cutoff_ang > -90:
        desc = desc + " It is cut south of " + str(south_cutoff_ang) + " degrees."
    if south_cutoff_row > 0:
        desc = desc + " The first " + str(south_cutoff_row) + " rows at south are deleted."

    # ---- guards (OGG:1425-1436)
    equator_index = np.searchsorted(y3[:, q], 0.0)
    if equator_index == 0:
        raise Exception("   Ooops: Equator is not in the grid")
    print("   Equator is at j=", equator_index)
    if equator_index % 2 == 0:
        raise Exception("Ooops: Equator is not going to be a u-point. Use option --south_cutoff_row to one more or on less row from south.")
    if y3.shape[0] % 2 == 0:
        raise Exception("Ooops: The number of j's in the supergrid is not even. Use option --south_cutoff_row to one more or on less row from south.")

    print("shapes: ", x3.shape, y3.shape, dx3.shape, dy3.shape, area3.shape, angle3.shape)
    if gridfilename is not None:
        write_nc(x3, y3, dx3, dy3, area3, angle3, axis_units="degrees", fnam=gridfilename, description=desc, history=hist,
                 source=source, no_changing_meta=no_changing_meta, debug=debug)
        print("Wrote the whole grid to file ", gridfilename)
    if plotem:
        _plot_grids({"sub": sub}, r_dp, lon_dp, lat_dp, lon0, lat0_SO, inverse_resolution, x3, y3)
    print("runtime(secs)  %s" % (time.time() - start_time))
    if return_arrays:
        return {"x": x3, "y": y3, "dx": dx3, "dy": dy3, "area": area3, "angle_dx": angle3, "sub": sub}


def build_parser():
    """The reference's flag surface (OGG:1452-1524), flag for flag."""
    parser = argparse.ArgumentParser(description="create ocean hgrid")
    parser.add_argument("-r", "--inverse_resolution", type=float, required=True,
                        help="inverse of the horizontal resolution (e.g. 4 for 1/4 degree)")
    parser.add_argument("-f", "--gridfilename", type=str, required=False, default="ocean_hgrid.nc", help="name for output grid file")
    parser.add_argument("--r_dp", type=float, required=False, default=0.0,
                        help="displacement factor/0.2 for the displaced south pole, do not specify both r_dp and lat_dp!")
    parser.add_argument("--exfracdp", type=float, required=False, default=0.49,
                        help="exclusion factor that determines the size of the hole arount SP!")
    parser.add_argument("--lon_dp", type=float, required=False, default=80.0, help="longitude of the displaced south pole")
    parser.add_argument("--lat_dp", type=float, required=False, default=-99.0,
                        help="latitude of the displaced south pole, do not specify both r_dp and lat_dp!")
    parser.add_argument("--south_cutoff_ang", type=float, required=False, default=-90.0, help="degrees south to start")
    parser.add_argument("--south_cutoff_row", type=int, required=False, default=0, help="rows to cut from the grid at south")
    parser.add_argument("--bipolar_lower_lat", type=float, required=False, default=-90.0,
                        help="starting (lower) latitude of Northern Bipolar sub grid")
    parser.add_argument("--mercator_lower_lat", type=float, required=False, default=-90.0,
                        help="starting (lower) latitude of Mercator sub grid")
    parser.add_argument("--mercator_upper_lat", type=float, required=False, default=-99.0,
                        help="ending (upper) latitude of Mercator sub grid")
    parser.add_argument("--south_ocean_lower_lat", type=float, required=False, default=-90.0,
                        help="starting (lower) latitude of SO sub grid")
    parser.add_argument("--south_ocean_upper_lat", type=float, required=False, default=-99.0,
                        help="ending (upper) latitude of SO sub grid")
    parser.add_argument("--no_south_cap", action="store_true", help="do not generate a southern cap sub grid")
    parser.add_argument("--match_dy", type=str, nargs="+", required=False, default=[],
                        help="set the number of j-points of subgrid such that latitude resolution (dy) becomes continous at the "
                             "joints. bp: Bipolar Cap to Mercator stitch; so: Southern Ocean to Mercator stitch; p125sc: buggy "
                             "Southern Cap to Southern Ocean stitch to reproduce the CM4X grid")
    parser.add_argument("--ensure_nj_even", action="store_true", required=False, default=False,
                        help="make the number of j-points in every subgrid even by dropping rows")
    parser.add_argument("--plotem", action="store_true", help="make a rudimentary plot of the subgrids")
    parser.add_argument("--skip_metrics", action="store_true", help="skip generating the metrics, only for fast debugging purposes")
    parser.add_argument("--write_subgrid_files", action="store_true", help="write subgrids to separate files ")
    parser.add_argument("--no_changing_meta", action="store_true", help="do not write meta data to netcdf files that might change")
    parser.add_argument("--enhanced_equatorial", type=int, required=False, default=0,
                        help="generate a subgrid that has an enhanced resolution around the equator")
    parser.add_argument("--shift_equator_to_u_point", action="store_false", required=False, default=True,
                        help="if the equator is not a u point shift the Mercator subgrid by 1 j-point to make it a u point , default=True")
    parser.add_argument("--grids", type=str, nargs="+", required=False, default="all",
                        help="specify the subgrids to generate, choices are bipolar, mercator, so, sc, all. Default is all")
    return parser


if __name__ == "__main__":
    main(**vars(build_parser().parse_args()))
