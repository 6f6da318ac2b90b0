"""GPU parity on seeded RANDOM shapes and parameters: sizes that are not multiples of anything (fewer columns than one wave
strip, one cell row, odd Ni, bands that start and end anywhere), cap latitudes and pole displacements drawn at random -- the
shapes the fixed configurations of test_gpu_parity.py / test_gpu_pipeline.py never hit.  Tolerances are those of
test_gpu_parity.py (module docstring there)."""
import numpy as np
import pytest

from oracle import ogg_oracle as orc
from test_gpu_parity import _check_supergrid, dp_quad_rel_tol

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ogg(hip):
    import ocean_model_grid_generator_amd.ocean_grid_generator as m
    return m


def _rel(a, b):
    assert a.shape == b.shape, (a.shape, b.shape)
    m = b != 0
    assert np.array_equal(a[~m], b[~m])
    return float((np.abs(a - b)[m] / np.abs(b[m])).max()) if m.any() else 0.0


@pytest.mark.parametrize("seed", range(12))
def test_bipolar_mesh_and_quadrature_random(ogg, seed):
    rng = np.random.default_rng(1000 + seed)
    Ni = int(rng.choice([4, 8, 12, 36, 60, 64, 124, 128, 252, 500, 1000]))   # the cap needs Ni % 4 == 0 for its pole columns
    Nj = int(rng.integers(1, 40))
    lat0 = float(rng.uniform(40.0, 88.0))
    lon_bp = float(rng.choice([-300.0, -280.0, 0.0, 17.5]))
    rp = np.tan(0.5 * (90 - lat0) * orc.PI_180)
    lam, phi, hi, hj = ogg.generate_bipolar_cap_mesh(Ni, Nj, lat0, lon_bp, ensure_nj_even=False)
    olam, ophi, ohi, ohj = orc.generate_bipolar_cap_mesh(Ni, Nj, lat0, lon_bp, ensure_nj_even=False)
    assert np.abs(phi - ophi).max() < 2e-11 and np.abs(lam - olam).max() < 5e-9   # lam is ill-conditioned next to the symmetry meridian
    assert np.abs(lam - olam)[:, 1:Ni // 4 - 1].max(initial=0.0) < 2e-11
    order = int(rng.choice([2, 3, 4, 5]))
    got = ogg.bipolar_cap_metrics_quad_fast(order, Ni, Nj, lat0, lon_bp, rp)
    want = orc.bipolar_cap_metrics_quad_fast(order, Ni, Nj, lat0, lon_bp, rp)
    for g, w, name in zip(got, want, ("dx", "dy", "area")):
        assert _rel(g, w) < 5e-14, (name, Ni, Nj, lat0, order)


@pytest.mark.parametrize("seed", range(8))
def test_bipolar_quadrature_random_bands(ogg, hip, seed):
    """The band entry point (j0, rows) used by the sharded pass, on random bands: bit-identical to the whole cap."""
    import torch
    from ocean_model_grid_generator_amd import _lib as L
    rng = np.random.default_rng(2000 + seed)
    nx = int(rng.choice([8, 60, 64, 200, 632]))
    ny = int(rng.integers(2, 30))
    lat0 = float(rng.uniform(55.0, 80.0))
    rp = float(np.tan(0.5 * (90 - lat0) * orc.PI_180))
    whole = ogg.bipolar_cap_metrics_quad_fast(5, nx, ny, lat0, -300.0, rp)
    cuts = sorted(set([0, ny + 1] + [int(c) for c in rng.integers(0, ny + 2, size=3)]))
    st = torch.cuda.current_stream().cuda_stream
    wsb = int(L.load().ogg_bipolar_quad_workspace_bytes(5, nx, ny))
    ws = torch.empty(wsb, dtype=torch.uint8, device="cuda")
    for lo, hi in zip(cuts[:-1], cuts[1:]):
        n_dx = hi - lo
        n_cell = min(hi, ny) - lo
        dx = torch.full((n_dx, nx), float("nan"), dtype=torch.float64, device="cuda")
        dy = torch.full((max(n_cell, 0), nx + 1), float("nan"), dtype=torch.float64, device="cuda")
        da = torch.full((max(n_cell, 0), nx), float("nan"), dtype=torch.float64, device="cuda")
        L.call("ogg_bipolar_cap_metrics_quad_ws_dev", 5, nx, ny, lat0, -300.0, rp, orc.RE_DEFAULT, lo, n_dx, max(n_cell, 0), dx.data_ptr(),
               dy.data_ptr() if n_cell > 0 else None, da.data_ptr() if n_cell > 0 else None, ws.data_ptr(), wsb, st)
        torch.cuda.synchronize()
        assert np.array_equal(dx.cpu().numpy(), whole[0][lo:hi]), (lo, hi)
        if n_cell > 0:
            assert np.array_equal(dy.cpu().numpy(), whole[1][lo:lo + n_cell])
            assert np.array_equal(da.cpu().numpy(), whole[2][lo:lo + n_cell])


@pytest.mark.parametrize("seed", range(8))
def test_displaced_pole_random(ogg, seed):
    rng = np.random.default_rng(3000 + seed)
    Ni = int(rng.choice([16, 72, 100, 256, 300, 720]))
    Nj = int(rng.integers(2, 24))
    lon_dp = float(rng.uniform(0.0, 360.0))
    r_dp = float(rng.uniform(0.05, 0.6))
    lat0 = float(rng.uniform(-82.0, -60.0))
    x, y, londp, latdp = ogg.generate_displaced_pole_grid(Ni, Nj, -300.0, lat0, lon_dp, r_dp)
    ox, oy, olondp, olatdp = orc.generate_displaced_pole_grid(Ni, Nj, -300.0, lat0, lon_dp, r_dp)
    assert abs(londp - olondp) < 1e-10 and abs(latdp - olatdp) < 1e-12
    assert np.abs(y - oy).max() < 1e-11
    d = np.abs(x - ox)
    d = np.minimum(d, np.abs(d - 360.0))   # a longitude within 1 ulp of the unwrap threshold may land on the other branch
    assert d.max() < 1e-9 and np.median(d) < 1e-12
    want = orc.displacedPoleCap_metrics_quad(4, Ni, Nj, -300.0, lat0, lon_dp, r_dp)
    for form in ("literal", "chord"):
        got = ogg.displacedPoleCap_metrics_quad(4, Ni, Nj, -300.0, lat0, lon_dp, r_dp, arc_form=form)
        tol = max(dp_quad_rel_tol(Ni), 1e-10)
        for g, w, name in zip(got, want, ("dx", "dy", "area")):
            jm = Nj // 2    # rows towards the joint: relative; the rows around the pole (h -> 0): relative to the field's scale
            assert _rel(g[jm:], w[jm:]) < tol, (form, name, Ni, Nj, lon_dp, r_dp, _rel(g[jm:], w[jm:]))
            assert np.abs(g - w).max() <= tol * np.abs(w).max(), (form, name, Ni, Nj, lon_dp, r_dp)


@pytest.mark.parametrize("seed", range(10))
def test_midas_and_angle_random_meshes(ogg, seed):
    """generate_grid_metrics_MIDAS / angle_x on arbitrary (non lat-lon) meshes of random shape, incl. one row, two columns."""
    rng = np.random.default_rng(4000 + seed)
    nj1 = int(rng.choice([1, 2, 3, 17, 33, 100]))
    ni1 = int(rng.choice([2, 3, 63, 64, 65, 129, 300, 1000]))
    lon = -300.0 + np.sort(rng.uniform(0, 360, ni1))
    lat = np.sort(rng.uniform(-85, 85, nj1))
    x = np.tile(lon, (nj1, 1)) + rng.normal(0, 1e-3, (nj1, ni1))
    y = np.tile(lat[:, None], (1, ni1)) + rng.normal(0, 1e-3, (nj1, ni1))
    a = ogg.angle_x(x, y)
    oa = orc.angle_x(x, y)
    assert np.abs(a - oa).max() < 1e-9
    if nj1 >= 2:
        dx, dy, area = ogg.generate_grid_metrics_MIDAS(x, y)
        odx, ody, oarea = orc.generate_grid_metrics_MIDAS(x, y)
        assert dx.shape == odx.shape and dy.shape == ody.shape and area.shape == oarea.shape
        assert np.all(np.abs(dx - odx) <= 1e-9 + 5e-15 * np.abs(odx)) and np.all(np.abs(dy - ody) <= 1e-9 + 5e-15 * np.abs(ody))
        assert np.all(np.abs(area - oarea) <= 1e-3 + 5e-11 * np.abs(oarea))


@pytest.mark.parametrize("seed", range(6))
def test_tripolar_pass_random_plans(hip, seed):
    """Whole passes at odd resolutions and band counts: the fused pass against one launch per sub-grid, bitwise, and both
    against the oracle."""
    import torch
    import ocean_model_grid_generator_amd.supergrid as sg
    rng = np.random.default_rng(5000 + seed)
    r = float(rng.choice([0.25, 0.5, 0.75, 1.0, 1.5, 3.0]))
    even = bool(rng.integers(0, 2))
    world = int(rng.choice([1, 2, 3, 5, 7]))
    plan = sg.SupergridPlan(r, ensure_nj_even=even)
    a = [sg.Supergrid(plan, rank=k, world=world, device="cuda:0", halo="recompute") for k in range(world)]
    b = [sg.Supergrid(plan, rank=k, world=world, device="cuda:0", halo="recompute") for k in range(world)]
    for g in a:
        g.launch = "pass"
        g.step()
    for g in b:
        g.launch, g.overlap = "kernels", False
        g.step()
    torch.cuda.synchronize()
    ga = sg.stitch(plan, [g.bands_to_host() for g in a])
    gb = sg.stitch(plan, [g.bands_to_host() for g in b])
    for f in sg.FIELDS:
        assert np.array_equal(ga[f], gb[f]), (f, r, even, world)
    try:
        want = orc.make_supergrid(r, ensure_nj_even=even, skip_doughnut_rows=True)
    except Exception as exc:   # the reference's own guards (OGG:1425-1436) reject this flag set: nothing to compare with
        assert "Ooops" in str(exc) or "repeated values" in str(exc)
        return
    _check_supergrid(ga, want, "random_plan_%d" % seed)


@pytest.mark.parametrize("seed", range(40))
def test_main_random_flags_pass_path_vs_function_level_and_oracle(hip, seed):
    """main() with randomly drawn flag sets -- resolution, --ensure_nj_even, displaced pole by radius or by latitude (or none),
    south cuts by row or by angle, --match_dy, --no_south_cap, --exfracdp, latitude overrides, --skip_metrics -- through the
    device-resident pass against the function-level path (bitwise, bipolar angle_dx excepted) and against the oracle; flag sets
    the reference's own guards reject must be rejected with the same text."""
    import ocean_model_grid_generator_amd.ocean_grid_generator as ogg
    from test_gpu_pipeline import _same_as_function_level, FIELDS
    rng = np.random.default_rng(7000 + seed)
    flags = dict(inverse_resolution=float(rng.choice([0.25, 0.5, 0.75, 1.0])), ensure_nj_even=bool(rng.integers(0, 2)))
    pole = int(rng.integers(0, 3))
    if pole == 1:
        flags["r_dp"] = float(rng.choice([0.1, 0.2, 0.3]))
    elif pole == 2:
        flags.update(lat_dp=float(rng.uniform(-88.0, -84.0)), lon_dp=float(rng.uniform(0.0, 359.0)))
    if pole and rng.integers(0, 2):
        flags["exfracdp"] = float(rng.choice([0.0, 0.3, 0.49, 0.6]))
    cut = int(rng.integers(0, 4))
    if cut == 1:
        flags["south_cutoff_row"] = int(rng.integers(1, 12))
    elif cut == 2 and flags["inverse_resolution"] <= 0.5:
        flags["south_cutoff_row"] = int(rng.integers(20, 45))      # may consume the whole cap
    elif cut == 3:
        flags["south_cutoff_ang"] = float(rng.uniform(-86.0, -79.0))
    if rng.integers(0, 3) == 0:
        flags["match_dy"] = [["bp"], ["so"], ["bp", "so"], ["bp", "so", "p125sc"]][int(rng.integers(0, 4))]
    if rng.integers(0, 5) == 0:
        flags["no_south_cap"] = True
    if rng.integers(0, 4) == 0:
        flags["south_ocean_lower_lat"] = float(rng.uniform(-84.0, -75.0))
    if rng.integers(0, 6) == 0:
        flags["skip_metrics"] = True
    if seed >= 10:   # (the first ten seeds keep the flag sets they were first run with)
        extra = int(rng.integers(0, 8))
        if extra == 0:
            flags["grids"] = ["bipolar", "mercator", "so"]
        elif extra == 1:
            flags["grids"] = ["mercator", "so", "sc"]
        elif extra == 2:
            lat = float(rng.uniform(62.0, 68.0))
            flags.update(bipolar_lower_lat=lat, mercator_upper_lat=lat)
        elif extra == 3:
            flags["mercator_lower_lat"] = float(rng.uniform(-70.0, -60.0))
        elif extra == 4 and not pole:
            flags.update(inverse_resolution=2.0, enhanced_equatorial=4)
            flags.pop("south_cutoff_ang", None)
        elif extra == 5:
            flags["shift_equator_to_u_point"] = False
    r = flags["inverse_resolution"]
    oflags = {k: v for k, v in flags.items() if k != "inverse_resolution"}
    try:
        want = orc.make_supergrid(r, skip_doughnut_rows=True, **oflags)
    except BaseException as exc:   # rejected by the reference's own logic (guards, or a cut that needs a cap that is not there)
        with pytest.raises(BaseException) as ei:
            ogg.main(gridfilename=None, no_changing_meta=True, return_arrays=True, **flags)
        if isinstance(exc, Exception) and "Ooops" in str(exc):
            assert str(ei.value) == str(exc), flags
        return
    got = ogg.main(gridfilename=None, no_changing_meta=True, return_arrays=True, **flags)
    ref = ogg.main(gridfilename=None, no_changing_meta=True, return_arrays=True, path="functions", **flags)
    ref = dict(ref, sub={k: dict(zip(FIELDS, v)) for k, v in ref["sub"].items()})
    _same_as_function_level(got, ref, str(flags))
    _check_supergrid(got, want, "random_flags_%d" % seed)
