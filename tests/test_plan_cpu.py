"""CPU test of the host logic behind the device-resident pass: supergrid.SupergridPlan restates main()'s size selection (OGG:969-1197)
and south cuts (OGG:1268-1313); here its sub-grid sizes, cuts and stitched row count are compared with the oracle's restatement of
the same code on seeded random flag sets -- no GPU: the Mercator axis the plan would take from the device is handed in from the
oracle, and the oracle runs with --skip_metrics at coarse resolutions (a fraction of a second per case)."""
import numpy as np
import pytest

from oracle import ogg_oracle as orc


def _flags(seed):
    rng = np.random.default_rng(9000 + seed)
    f = dict(inverse_resolution=float(rng.choice([0.25, 0.5, 1.0])), ensure_nj_even=bool(rng.integers(0, 2)))
    pole = int(rng.integers(0, 3))
    if pole == 1:
        f["r_dp"] = float(rng.choice([0.1, 0.2, 0.3]))
    elif pole == 2:
        f.update(lat_dp=float(rng.uniform(-88.0, -84.0)), lon_dp=float(rng.uniform(0.0, 359.0)))
    if pole and rng.integers(0, 2):
        f["exfracdp"] = float(rng.choice([0.0, 0.3, 0.49, 0.6]))
    cut = int(rng.integers(0, 4))
    if cut == 1:
        f["south_cutoff_row"] = int(rng.integers(1, 14))
    elif cut == 2:
        f["south_cutoff_row"] = int(rng.integers(14, 60))
    elif cut == 3 and not pole:          # by angle on a regular cap: the plan knows the cap's latitudes without running a kernel
        f["south_cutoff_ang"] = float(rng.uniform(-88.0, -79.0))
    if rng.integers(0, 3) == 0:
        f["match_dy"] = [["bp"], ["so"], ["bp", "so"], ["bp", "so", "p125sc"]][int(rng.integers(0, 4))]
    if rng.integers(0, 5) == 0:
        f["no_south_cap"] = True
    if rng.integers(0, 4) == 0:
        f["south_ocean_lower_lat"] = float(rng.uniform(-84.0, -75.0))
    extra = int(rng.integers(0, 8))
    if extra == 0:
        f["grids"] = ["bipolar", "mercator", "so"]
    elif extra == 1:
        f["grids"] = ["mercator", "so", "sc"]
    elif extra == 2:
        lat = float(rng.uniform(62.0, 68.0))
        f.update(bipolar_lower_lat=lat, mercator_upper_lat=lat)
    elif extra == 3:
        f["mercator_lower_lat"] = float(rng.uniform(-70.0, -60.0))
    elif extra == 4:
        f["shift_equator_to_u_point"] = False
    return f


@pytest.mark.parametrize("seed", range(60))
def test_plan_sizes_and_cuts_follow_the_oracle(seed):
    import ocean_model_grid_generator_amd.supergrid as SG
    flags = _flags(seed)
    r = flags["inverse_resolution"]
    oflags = {k: v for k, v in flags.items() if k != "inverse_resolution"}
    Ni = int(r * 2 * 360)
    phi_s, phi_n = -66.85954725, 64.05895973
    if flags.get("mercator_upper_lat", -99.0) > -90:
        phi_n = flags["mercator_upper_lat"]
    if flags.get("mercator_lower_lat", -99.0) > -90:
        phi_s = flags["mercator_lower_lat"]
    try:
        want = orc.make_supergrid(r, skip_metrics=True, skip_doughnut_rows=True, **oflags)
    except Exception as exc:
        if "Ooops" not in str(exc):                 # e.g. a cut that needs a cap which is not there: the plan must refuse as well
            with pytest.raises(Exception):
                y0, y1 = orc.mercator_y_star(Ni, phi_s, phi_n, flags.get("shift_equator_to_u_point", True), flags["ensure_nj_even"])
                SG.SupergridPlan(r, mercator_axis=(y0, orc.phi_mercator(Ni, np.arange(y0, y1 + 1))), **oflags).south_cut()
        return                                      # (the final guards need the fields: tests/test_gpu_random_shapes.py)
    y0, y1 = orc.mercator_y_star(Ni, phi_s, phi_n, flags.get("shift_equator_to_u_point", True), flags["ensure_nj_even"])
    plan = SG.SupergridPlan(r, mercator_axis=(y0, orc.phi_mercator(Ni, np.arange(y0, y1 + 1))), **oflags)
    c_sc, c_so, gone = plan.south_cut()
    rows = {s.name: s.nj1 for s in plan.subs}
    if gone:
        rows.pop("SC")
        rows["SO"] = max(rows["SO"] - c_so, 0)
    elif "SC" in rows:
        rows["SC"] -= c_sc
    # (a sub-grid the cut removes altogether is an empty piece in the reference and simply absent from the plan)
    assert {k: v for k, v in rows.items() if v} == {k: v[1].shape[0] for k, v in want["sub"].items() if v[1].shape[0]}, (flags, rows)
    assert plan.nyp == want["y"].shape[0] and plan.cells == want["area"].size, flags
    for s in plan.subs:
        if s.kind == "bipolar":     # (the mesh's first row reproduces the joint latitude to an ulp of the projection, OGG:41)
            assert abs(s.lat0_bp - want["sub"]["BP"][1][0, 0]) < 1e-12


def test_plot_helpers_and_row_cuts_are_back():
    """cut_below / cut_above / plot_mesh_in_* / displacedPoleCap_plot are importable from the drop-in module again (the reference's
    OGG:604-679); the row cuts are plain numpy and follow the reference's loop, including its "last row when none exceeds" quirk."""
    import numpy as np
    from ocean_model_grid_generator_amd import ocean_grid_generator as ogg
    for name in ("cut_below", "cut_above", "plot_mesh_in_latlon", "plot_mesh_in_xyz", "displacedPoleCap_plot"):
        assert callable(getattr(ogg, name))
    phi = np.tile(np.array([-80.0, -60.0, -40.0, -20.0]).reshape(-1, 1), (1, 3))
    lam = np.tile(np.arange(3.0), (4, 1))
    a, b = ogg.cut_below(lam, phi, -50.0)
    assert a.shape == (2, 3) and b[0, 0] == -40.0
    a, b = ogg.cut_above(lam, phi, -50.0)
    assert a.shape == (2, 3) and b[-1, 0] == -60.0
    a, b = ogg.cut_below(lam, phi, 10.0)      # nothing exceeds: the reference's loop ends on the last row
    assert a.shape == (1, 3) and b[0, 0] == -20.0
    a, b = ogg.cut_above(lam, phi, 10.0)
    assert a.shape == (3, 3)
    try:
        import matplotlib  # noqa: F401
    except ImportError:
        import pytest
        with pytest.raises(Exception, match="matplotlib"):     # asked for a plot without matplotlib: an error, not silence
            ogg.plot_mesh_in_latlon(lam, phi)
