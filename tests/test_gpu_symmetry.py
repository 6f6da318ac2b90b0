"""Mirrored columns of the two caps (OGG_SYM_MIRROR, the default since round 5) against every-column evaluation (OGG_SYM_NONE, what the
reference does: OGG:168-172, 583-584) and against the oracle.

The bipolar projection (OGG:33-100) is mirror-symmetric about its pole meridians and fold lines, the displaced-pole map (OGG:447-467) about
the meridian through lon_dp.  The reference's own results are mirror images of each other only up to its own rounding (lon_bp + i 360/Ni is
rounded at another magnitude at either end of a row): what a mirrored kernel returns at an image column is the REFERENCE'S value at the
source column.  These tests hold (i) the columns that are still evaluated one by one -- the source columns themselves, the neighbourhoods
of the fold lines and of the pole meridians, the rows next to the pole points -- to bit identity with OGG_SYM_NONE; (ii) the images to the
measured size of the reference's own asymmetry; (iii) both evaluations to the SAME bounds against the oracle as before (SUB_TOL)."""
import math

import numpy as np
import pytest

from oracle import ogg_oracle as orc

pytestmark = pytest.mark.gpu

PI_180 = np.pi / 180


@pytest.fixture(scope="module")
def ogg(hip):
    import ocean_model_grid_generator_amd.ocean_grid_generator as m
    return m


def rel(a, b):
    with np.errstate(divide="ignore", invalid="ignore"):
        r = np.abs(a - b) / np.abs(b)
    r[~np.isfinite(r)] = 0.0
    return r


CAPS = {"r0.5": (360, 60, 64.05895973), "r1": (720, 120, 64.05895973), "r2": (1440, 238, 64.97316302279852), "r4": (2880, 480, 64.0589597296948),
        "r8": (5760, 960, 64.03160594077568)}


@pytest.mark.parametrize("size", ["r0.5", "r1", "r2", "r4", "r8"])
def test_bipolar_quadrature_mirrored_vs_every_column(ogg, size):
    nx, ny, lat0 = CAPS[size]
    lon_bp, rp = -300.0, float(np.tan(0.5 * (90 - lat0) * PI_180))
    a = ogg.bipolar_cap_metrics_quad_fast(5, nx, ny, lat0, lon_bp, rp, symmetry=False)
    b = ogg.bipolar_cap_metrics_quad_fast(5, nx, ny, lat0, lon_bp, rp, symmetry=True)
    q, z = nx // 4, int(math.ceil(6.0 * nx / 360.0))
    assert 2 * z <= q
    # first cell row that carries the guard (plan_quad): evaluated at every column either way
    jg = max(int(math.floor(ny * (math.degrees(math.acos(2.0 / math.sqrt(4000.0))) - lat0) / (90.0 - lat0))) - 1, 0)
    own_cells = np.r_[0:q, 2 * q - z:2 * q + z, nx - z:nx]
    own_cols = np.r_[0:q + 1, 2 * q - z:2 * q + z + 1, nx - z:nx + 1]
    for name, va, vb, own in (("dx", a[0], b[0], own_cells), ("dy", a[1], b[1], own_cols), ("area", a[2], b[2], own_cells)):
        assert np.array_equal(va[:, own], vb[:, own]), (name, "a column that is evaluated at its own position changed")
        assert np.array_equal(va[jg:], vb[jg:]), (name, "a row that carries the guard changed")
        # the images: the reference's own asymmetry six degrees from a fold line (oracle: <= 7e-15 area, 3e-15 dx, 5e-15 dy at 1/8 and
        # 1/16 degree) + the kernel's own rounding (dy relative to the row's largest value: it vanishes on the fold lines)
        scale = np.abs(va).max(axis=1, keepdims=True) if name == "dy" else np.abs(va)
        assert (np.abs(va - vb) / scale).max() < 2e-14, (name, float((np.abs(va - vb) / scale).max()))
    # both against the oracle on a sample of rows: the bound of tests/test_gpu_parity.py (5e-14; measured 9e-15)
    for j in sorted(set(list(range(0, ny, max(ny // 6, 1))) + [jg - 1, ny - 1])):
        o = orc.bipolar_cap_metrics_quad_fast(5, nx, ny, lat0, lon_bp, rp, j_first=j, j_last=j + 1)
        for k, name in enumerate(("dx", "dy", "area")):
            scale = np.abs(o[k][j]).max() if name == "dy" else np.abs(o[k][j])
            for v in (a, b):
                assert (np.abs(v[k][j] - o[k][j]) / scale).max() < (5e-14 if j < ny - 1 else 2e-13), (name, j)


@pytest.mark.parametrize("size", ["r0.5", "r2", "r4", "r8"])
def test_bipolar_mesh_mirrored_vs_every_column(ogg, size):
    nx, ny, lat0 = CAPS[size]
    lon_bp = -300.0
    a = ogg.generate_bipolar_cap_mesh(nx, ny, lat0, lon_bp, symmetry=False)
    b = ogg.generate_bipolar_cap_mesh(nx, ny, lat0, lon_bp, symmetry=True)
    q, zf, zm = nx // 4, 2, max(int(math.ceil(2.0 * nx / 360.0)), 2)
    own = np.r_[0:q + zm + 1, 2 * q - zf:2 * q + zf + 1, 3 * q - zm:3 * q + zm + 1, nx - zf:nx + 1]
    for k, name in enumerate(("x", "y")):
        assert np.array_equal(a[k][:, own], b[k][:, own]), name
    # h_i_inv (Ni columns), h_j_inv: symmetric like the metrics
    assert np.array_equal(a[2][:, own[own < nx]], b[2][:, own[own < nx]]) and np.array_equal(a[3][:, own], b[3][:, own])
    assert rel(b[2], a[2]).max() < 1e-13 and (np.abs(b[3] - a[3]) / np.abs(a[3]).max(axis=1, keepdims=True)).max() < 1e-13
    xo, yo, _, _ = orc.generate_bipolar_cap_mesh(nx, ny, lat0, lon_bp)
    # the images: x within 1e-12 degrees of the ORACLE at every point (north_star's bound; two degrees from a pole meridian the reference's
    # own columns differ by 3.4e-13), y 1e-13
    for v in (a, b):
        assert np.abs(v[0] - xo).max() < 1e-12 and np.abs(v[1] - yo).max() < 1e-13, (float(np.abs(v[0] - xo).max()), float(np.abs(v[1] - yo).max()))
    assert np.abs(a[0] - b[0]).max() < 6e-13 and np.abs(a[1] - b[1]).max() < 6e-14


@pytest.mark.parametrize("case", [(360, 70, 80.0, 0.2, 4), (720, 70, 80.0, 0.34135899793333113, 4), (1440, 140, -100.0, 0.2, 4), (720, 70, 60.0, 0.3, 2),
                                  (2880, 280, 80.0, 0.2, 4), (720, 70, 80.3, 0.2, 4)])
def test_displaced_pole_quadrature_mirrored_vs_every_column(ogg, case):
    nx, ny, lon_dp, r_dp, order = case
    lon0, lat0 = -300.0, -78.0
    a = ogg.displacedPoleCap_metrics_quad(order, nx, ny, lon0, lat0, lon_dp, r_dp, arc_form="chord", symmetry=False)
    b = ogg.displacedPoleCap_metrics_quad(order, nx, ny, lon0, lat0, lon_dp, r_dp, arc_form="chord", symmetry=True)
    ic = ((lon_dp - lon0) % 360.0) * nx / 360.0
    if ic != int(ic):   # the pole's meridian between two columns: symmetry declined, every column evaluated
        assert all(np.array_equal(x, y) for x, y in zip(a, b))
        return
    c0 = int(ic) % nx
    if c0 > nx // 2:
        c0 -= nx // 2
    own_cells, own_cols = np.r_[c0:c0 + nx // 2], np.r_[c0:c0 + nx // 2 + 1]
    jk = int(math.ceil(0.49 * ny))   # the rows main() keeps (the rows around r = r_pole swing by 180 degrees between two columns)
    tol = 2e-12 * nx                 # dp_quad_rel_tol(Ni) of tests/test_gpu_parity.py: the reference's own finite-difference noise
    for name, va, vb, own in (("dx", a[0], b[0], own_cells), ("dy", a[1], b[1], own_cols), ("area", a[2], b[2], own_cells)):
        assert np.array_equal(va[:, own], vb[:, own]), name
        assert rel(vb[jk:], va[jk:]).max() < tol, (name, float(rel(vb[jk:], va[jk:]).max()))
    # the literal form never mirrors
    lit_a = ogg.displacedPoleCap_metrics_quad(order, nx, ny, lon0, lat0, lon_dp, r_dp, arc_form="literal", symmetry=False)
    lit_b = ogg.displacedPoleCap_metrics_quad(order, nx, ny, lon0, lat0, lon_dp, r_dp, arc_form="literal", symmetry=True)
    assert all(np.array_equal(x, y) for x, y in zip(lit_a, lit_b))
    o = orc.displacedPoleCap_metrics_quad(order, nx, ny, lon0, lat0, lon_dp, r_dp, j_first=jk)
    for k in range(3):
        for v in (a, b):
            assert rel(v[k][jk:], o[k][jk:]).max() < tol


@pytest.mark.parametrize("flags", [dict(inverse_resolution=0.5, r_dp=0.2, ensure_nj_even=True), dict(inverse_resolution=2.0),
                                   dict(inverse_resolution=0.5, lon_dp=80.0, lat_dp=-85.85, ensure_nj_even=True)])
@pytest.mark.parametrize("world", [1, 3])
def test_pass_with_mirrored_caps(hip, flags, world):
    """The fused pass, mirrored and not: the same relation as the function-level kernels, band by band (the images are written inside a
    row: a band split along rows does not see them); and the mirrored pass is bit-identical to the mirrored function-level entry points."""
    from ocean_model_grid_generator_amd import supergrid
    out = {}
    for sym in (False, True):
        plan = supergrid.SupergridPlan(cap_symmetry=sym, **flags)
        parts = []
        for r in range(world):
            g = supergrid.Supergrid(plan, rank=r, world=world, device="cuda:0", halo="recompute")
            g.step()
            parts.append(g.bands_to_host())
            g.close()
        out[sym] = supergrid.stitch(plan, parts, guards=True)
        g = supergrid.Supergrid(plan, rank=0, world=1, device="cuda:0", halo="recompute")
        g.launch, g.overlap = "kernels", False
        g.step()
        alone = supergrid.stitch(plan, [g.bands_to_host()], guards=True)
        g.close()
        for f in ("x", "y", "dx", "dy", "area", "angle_dx"):
            assert np.array_equal(out[sym][f], alone[f]), (sym, f)
    want = orc.make_supergrid(flags["inverse_resolution"], **{k: v for k, v in flags.items() if k != "inverse_resolution"}, skip_doughnut_rows=True)
    from test_gpu_parity import _check_supergrid
    for sym in (False, True):
        _check_supergrid(out[sym], want, "sym%d" % sym)
    assert any(not np.array_equal(out[False][f], out[True][f]) for f in ("dx", "dy", "area"))   # (the mirrored pass did mirror)
