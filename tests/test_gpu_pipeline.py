"""GPU tests of the device-resident, band-sharded pass (supergrid.Supergrid): same numbers as main(), and the band
decomposition is bit-invariant (every output element depends only on (i, j) and at most one halo row)."""
import json
import os

import numpy as np
import pytest

from oracle import ogg_oracle as orc
from test_gpu_parity import _check_supergrid, FIELD_TOL  # noqa: F401

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
FIELDS = ("x", "y", "dx", "dy", "area", "angle_dx")


@pytest.fixture(scope="module")
def sg(hip):
    import ocean_model_grid_generator_amd.supergrid as m
    return m


def run(sg, plan, world=1, halo="local", latlon="stencil"):
    ranks = []
    for r in range(world):
        ranks.append(sg.Supergrid(plan, rank=r, world=world, device="cuda:0", halo=halo, peers=ranks, latlon=latlon))
    for g in ranks:
        g.phase_a()
    for g in ranks:
        g.exchange_halo()
    for g in ranks:
        g.phase_b()
    return sg.stitch(plan, [g.bands_to_host() for g in ranks])


CONFIGS = {
    "r1_cut2": dict(inverse_resolution=1.0, south_cutoff_row=2),
    "r2": dict(inverse_resolution=2.0),
    "r2_skip_metrics": dict(inverse_resolution=2.0, skip_metrics=True),
    "r0.25_even": dict(inverse_resolution=0.25, ensure_nj_even=True),
    "r0.5_dp": dict(inverse_resolution=0.5, r_dp=0.2, ensure_nj_even=True),
    "r0.5_latdp": dict(inverse_resolution=0.5, lon_dp=80.0, lat_dp=-85.85, ensure_nj_even=True),
}


def plan_for(sg, name, **kw):
    """SupergridPlan of CONFIGS[name]; "name+literal" / "name+chord" name the arc form of the displaced-pole quadrature explicitly (the
    default is the chord form since round 4; the literal form -- the reference's operation sequence, a fourth launch with its own
    look-back words -- keeps every pipeline test it had while it was the default)."""
    base, _, arc = name.partition("+")
    return sg.SupergridPlan(**CONFIGS[base], **({"dp_arc": arc} if arc else {}), **kw)


@pytest.mark.parametrize("name", sorted(CONFIGS))
def test_pipeline_vs_oracle(sg, name):
    flags = dict(CONFIGS[name])
    plan = sg.SupergridPlan(**flags)
    got = run(sg, plan)
    cfg = json.load(open(os.path.join(GOLD, "ref_hashes.json")))["configs"][name]
    for f in FIELDS:
        assert list(got[f].shape) == cfg["shapes"][f], (f, got[f].shape)       # shapes pinned by the reference run
    assert plan.nyp == cfg["shapes"]["x"][0] and plan.cells == cfg["shapes"]["area"][0] * cfg["shapes"]["area"][1]
    r = flags.pop("inverse_resolution")
    want = orc.make_supergrid(r, skip_doughnut_rows=True, **flags)
    _check_supergrid(got, want, "pipe_" + name)


@pytest.mark.parametrize("name", ["r1_cut2", "r0.5_dp", "r0.5_dp+literal"])
@pytest.mark.parametrize("world,halo", [(2, "local"), (3, "local"), (8, "local"), (5, "recompute")])
def test_band_decomposition_is_bit_invariant(sg, name, world, halo):
    plan = plan_for(sg, name)
    one = run(sg, plan, 1)
    many = run(sg, plan, world, halo)
    for f in FIELDS:
        assert np.array_equal(one[f], many[f]), (f, world, halo)


@pytest.mark.parametrize("name,world", [("r2", 8), ("r2", 3), ("r0.5_dp", 4), ("r1_cut2", 2)])
def test_measured_split_keeps_the_bits(sg, name, world):
    """SupergridPlan.calibrate_split on this GPU (one process: no broadcast): the last rank's share comes out of timings, so the band
    edges differ from run to run and from box to box -- the stitched fields must not.  The measured share is in effect (rows_of follows
    it), every row is covered once, and the fused pass over the measured bands gives the single-rank bits."""
    plan = plan_for(sg, name)
    one = run_pass_mode(sg, plan, 1)
    st = plan.calibrate_split("cuda:0", rank=0, world=world, broadcast=False, rounds=2)
    if plan.skip_metrics or not any(s.kind == "bipolar" for s in plan.subs):
        assert st is None
    else:
        assert "measured by rank 0" in st["source"] and st["top_capacity"]["world"] == world and 0.3 <= st["top_capacity"]["share_of_last_rank"] <= 1.3
        assert 0 < st["tail_us"] < st["pass_us"]
    for s in plan.subs:
        rows = []
        for r in range(world):
            lo, hi = sg.Supergrid.rows_of(s, r, world)
            rows += list(range(lo, hi))
        assert rows == list(range(s.nj1)), s.name
    many = run_pass_mode(sg, plan, world)
    for f in FIELDS:
        assert np.array_equal(one[f], many[f], equal_nan=False), (f, world)


@pytest.mark.parametrize("name", ["r1_cut2", "r2", "r2_skip_metrics", "r0.5_dp", "r0.25_even"])
@pytest.mark.parametrize("world", [1, 3, 8])
def test_fused_latlon_kernel_is_bit_identical_to_stencil(sg, name, world):
    """The fused lat-lon kernel (no HBM reads, no halo) against tile + generic stencil kernel, sharded or not."""
    plan = plan_for(sg, name)
    stencil = run(sg, plan, 1, latlon="stencil")
    fused = run(sg, plan, world, latlon="fused")
    for f in FIELDS:
        assert np.array_equal(stencil[f], fused[f]), (f, world)


@pytest.mark.parametrize("name", ["r1_cut2", "r2_skip_metrics", "r0.25_even"])
def test_row_ordered_latlon_kernel_is_bit_identical(sg, name, monkeypatch):
    """ogg_latlon_supergrid_rows_ws_dev ((field, row)-ordered workgroups fed from tables) against the column-tile kernel."""
    plan = plan_for(sg, name)
    monkeypatch.setenv("OGG_LATLON_ROWS", "0")
    tiles = run(sg, plan, 1, latlon="fused")
    monkeypatch.setenv("OGG_LATLON_ROWS", "1")
    rows = run(sg, plan, 3, latlon="fused")
    for f in FIELDS:
        assert np.array_equal(tiles[f], rows[f]), f
    # the resident-workgroup caps of the two stand-alone kernels (OGG_FUSED_MAX_WG, OGG_ROWS_MAX_WG) change who writes what, not what
    monkeypatch.setenv("OGG_ROWS_MAX_WG", "7")
    rows7 = run(sg, plan, 2, latlon="fused")
    monkeypatch.setenv("OGG_LATLON_ROWS", "0")
    monkeypatch.setenv("OGG_FUSED_MAX_WG", "5")
    tiles5 = run(sg, plan, 2, latlon="fused")
    for f in FIELDS:
        assert np.array_equal(tiles[f], rows7[f]) and np.array_equal(tiles[f], tiles5[f]), f


@pytest.mark.parametrize("name", ["r1_cut2", "r0.5_dp", "r0.5_dp+literal", "r2_skip_metrics"])
@pytest.mark.parametrize("world", [2, 5])
def test_band_sharded_ranks_write_one_netcdf_file(sg, name, world, tmp_path):
    """Every rank of a band-sharded run streams its own bands into ONE NetCDF file at their byte offsets (rank 0 writes the header;
    no gather): the file is byte-identical to the one a single rank writes."""
    import torch
    plan = plan_for(sg, name)
    one = sg.Supergrid(plan, device="cuda:0")
    one.step()
    cut = one.south_cut()
    one.write_nc(str(tmp_path / "one.nc"), cut, no_changing_meta=True)
    ranks = [sg.Supergrid(plan, rank=r, world=world, device="cuda:0", halo="recompute") for r in range(world)]
    for g in ranks:
        g.step()
    torch.cuda.synchronize()
    for g in ranks:      # rank 0 first: it creates the file (in a real run a barrier stands here)
        g.write_nc(str(tmp_path / "many.nc"), cut, no_changing_meta=True)
    a, b = open(tmp_path / "one.nc", "rb").read(), open(tmp_path / "many.nc", "rb").read()
    assert len(a) == len(b) and a == b
    got = sg.stitch(plan, [one.bands_to_host()])
    from scipy.io import netcdf_file
    nc = netcdf_file(str(tmp_path / "many.nc"), "r", mmap=False)
    for f in FIELDS:
        assert np.array_equal(nc.variables[f][:], got[f]), f
    nc.close()


def test_lookback_is_stable_under_load(sg):
    """The displaced-pole look-back (strip maps published and polled across workgroups of a launch that also carries every other
    role) over many passes and under uneven load: two grids of different size run their passes on two streams at once, 150 times;
    the cap's fields keep the bits of the first pass, and no wait ever times out."""
    import torch
    plans = [sg.SupergridPlan(2.0, r_dp=0.2), sg.SupergridPlan(1.0, lon_dp=123.0, lat_dp=-86.5, dp_arc="literal")]
    grids = [sg.Supergrid(p, device="cuda:0") for p in plans]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]

    def fingerprint(g):
        b = g.buf["SC"]
        return [int(b[f].view(torch.int64).sum().item()) for f in ("x", "y", "angle_dx", "dx", "dy", "area")]

    for g in grids:
        g.step()
    torch.cuda.synchronize()
    first = [fingerprint(g) for g in grids]
    for rep in range(150):
        for g, st in zip(grids, streams):
            with torch.cuda.stream(st):
                for f in ("x", "dx", "area"):
                    g.buf["SC"][f].fill_(float("nan"))
                g.run_pass()
        if rep % 50 == 49:
            torch.cuda.synchronize()
            for g, want in zip(grids, first):
                g.check_lookback_flags()
                assert fingerprint(g) == want, rep
    torch.cuda.synchronize()


def run_pass_mode(sg, plan, world=1):
    import torch
    out = []
    for r in range(world):
        g = sg.Supergrid(plan, rank=r, world=world, device="cuda:0", halo="recompute", latlon="fused")
        for f in FIELDS:  # poison: every element must be written by the pass
            for s in plan.subs:
                g.buf[s.name][f].fill_(float("nan"))
        g.launch = "pass"
        g.step()
        torch.cuda.synchronize()
        g.check_lookback_flags()
        out.append(g.bands_to_host())
    return sg.stitch(plan, out)


@pytest.mark.parametrize("name", ["r1_cut2", "r2", "r2_skip_metrics", "r0.5_dp", "r0.5_latdp", "r0.5_dp+literal", "r0.5_latdp+literal", "r0.25_even"])
@pytest.mark.parametrize("world", [1, 2, 8])
def test_tripolar_pass_is_bit_identical_to_separate_kernels(sg, name, world):
    """ogg_tripolar_pass_dev (lat-lon strips and cap workgroups sharing three launches) against one launch per sub-grid and phase."""
    plan = plan_for(sg, name)
    kernels = run(sg, plan, 1, latlon="fused")
    fused = run_pass_mode(sg, plan, world)
    for f in FIELDS:
        assert np.array_equal(kernels[f], fused[f], equal_nan=False), (f, world)


@pytest.mark.parametrize("name,world", [("r2", 1), ("r2", 8), ("r0.5_latdp", 1), ("r0.5_dp", 3), ("r0.5_latdp+literal", 1), ("r0.5_dp+literal", 3),
                                        ("r1_cut2", 2), ("r2_skip_metrics", 1)])
def test_tables_of_the_next_pass_ride_in_launch_b(sg, name, world, monkeypatch):
    """A plan keeps two workspaces per cap; the last workgroups of launch B of pass k build the tables (and clear the look-back words) of
    pass k + 1 in the other one, and pass k + 1 starts with launch B.  30 passes back to back per rank with EVERY output poisoned on the
    stream between two passes (nothing of pass k + 1 may reach an output array during pass k: the j = ny row of the bipolar dx goes
    through the workspace), passes that time their launches (and run launch A themselves) mixed in, against a plan with one slot: all
    fields bit for bit, no look-back flag, and all but the first pass and the timed ones really started with launch B."""
    import torch
    from ocean_model_grid_generator_amd import _lib as L
    plan = plan_for(sg, name)
    monkeypatch.setenv("OGG_PASS_SLOTS", "1")
    base = run_pass_mode(sg, plan, world)
    monkeypatch.delenv("OGG_PASS_SLOTS")
    out = []
    for r in range(world):
        g = sg.Supergrid(plan, rank=r, world=world, device="cuda:0", halo="recompute", latlon="fused")
        g.launch = "pass"
        g.reserve_pass_events(4)
        timed = 0
        for k in range(30):
            for f in FIELDS:
                for s in plan.subs:
                    g.buf[s.name][f].fill_(float("nan"))
            g.pass_events = [] if k in (7, 8, 20) else None
            timed += 1 if g.pass_events is not None else 0
            g.run_pass()
            if g.pass_events is not None:
                g.pass_launch_times_ms()
                g.pass_events = None
        torch.cuda.synchronize()
        g.check_lookback_flags()
        h = g._pass_args[4]
        lib = L.load()
        two = not (plan.skip_metrics and not any(s.kind == "dpole" for s in plan.subs))
        assert lib.ogg_supergrid_pass_plan_slots(h) == (2 if two else 1)
        assert lib.ogg_supergrid_pass_plan_carried_runs(h) == ((30 - 1 - timed) if two else 0)
        out.append(g.bands_to_host())
    got = sg.stitch(plan, out)
    for f in FIELDS:
        assert np.array_equal(base[f], got[f], equal_nan=False), f


@pytest.mark.parametrize("name", ["r0.5_dp", "r0.5_latdp+literal", "r0.5_dp+literal", "r2"])
def test_captured_pass_replays_from_a_clean_slot(sg, name):
    """A pass captured into a HIP graph (Supergrid.capture: one eager pass, then the capture) is REPLAYED many times: the captured pass
    must run its own launch A -- counters, tickets and look-back words reset on every replay -- and carry nobody's next tables
    (ogg_pass.hip run_pass_pipe: hipStreamIsCapturing).  Five replays with every output poisoned in between, then eager passes again
    (the plan must not believe that the captured pass prepared a slot): the eager bits every time, no look-back flag, and the plan's
    run counter unchanged by the capture."""
    import torch
    from ocean_model_grid_generator_amd import _lib as L
    plan = plan_for(sg, name)
    want = run_pass_mode(sg, plan, 1)
    g = sg.Supergrid(plan, device="cuda:0", halo="recompute", latlon="fused")
    g.launch = "pass"
    for _ in range(3):
        g.run_pass()
    torch.cuda.synchronize()
    lib = L.load()
    carried0 = lib.ogg_supergrid_pass_plan_carried_runs(g._pass_args[4])
    g.capture()                         # one more eager pass + the capture
    carried1 = lib.ogg_supergrid_pass_plan_carried_runs(g._pass_args[4])
    assert carried1 - carried0 <= 1     # the eager pass of capture() may have been carried; the captured one is not counted

    def poison():
        for f in FIELDS:
            for s_ in plan.subs:
                g.buf[s_.name][f].fill_(float("nan"))

    def check(tag):
        torch.cuda.synchronize()
        g.check_lookback_flags()
        got = sg.stitch(plan, [g.bands_to_host()])
        for f in FIELDS:
            assert np.array_equal(want[f], got[f]), (tag, f)

    for k in range(5):
        poison()
        g.replay()
        check("replay %d" % k)
    for k in range(3):                   # eager again, straight after replays: launch A must run (no slot is 'ready')
        poison()
        g.run_pass()
        check("eager %d after replays" % k)
    poison()
    g.replay()
    check("replay after eager passes")
    g.close()


@pytest.mark.parametrize("name", ["r0.5_dp", "r0.5_latdp", "r0.5_dp+literal", "r0.5_latdp+literal", "r2"])
def test_function_level_calls_between_two_passes_of_a_plan(sg, name):
    """What a plan's launch B builds for the next pass lives in workspaces of the plan's own: a caller that runs the stand-alone kernels on
    ITS workspaces between two passes (Supergrid in "kernels" mode shares them with the pass: tickets taken, look-back words published,
    counters left non-zero) does not disturb the next pass.  Passes and kernel-mode steps alternate in every phase of the two slots, all
    outputs poisoned before every pass: the pass's bits every time."""
    import torch
    plan = plan_for(sg, name)
    want = run_pass_mode(sg, plan, 1)
    g = sg.Supergrid(plan, device="cuda:0", halo="recompute", latlon="fused")
    for k, n_pass in enumerate((1, 2, 1, 3, 2)):
        g.launch, g.overlap = "kernels", False
        g.step()
        g.launch = "pass"
        for _ in range(n_pass):
            for f in FIELDS:
                for s in plan.subs:
                    g.buf[s.name][f].fill_(float("nan"))
            g.run_pass()
        torch.cuda.synchronize()
        g.check_lookback_flags()
        got = sg.stitch(plan, [g.bands_to_host()])
        for f in FIELDS:
            assert np.array_equal(want[f], got[f], equal_nan=False), (f, k)


@pytest.mark.parametrize("name,dp_arc", [("r0.5_dp", "literal"), ("r0.5_dp", "chord"), ("r1_cut2", "literal"), ("r0.25_even", "literal")])
def test_pass_does_not_depend_on_tiling_knobs(sg, name, dp_arc, monkeypatch):
    """Chunk lengths of both quadratures (one row ... the whole cap), rows per workgroup of both meshes, the number of resident lat-lon
    workgroups and of their helpers, strips handed out from per-tile counters or owned, the strips' row scalars read from the table the table
    launch builds (the default of a plan handle) or evaluated by the strips themselves (OGG_PASS_LL_TABLE=0), the dispatch order of the
    roles, the unused dynamic LDS that caps launch B's workgroups per CU, the weights of the band split (read when the PLAN is built): none
    of them may change a bit of the result."""
    plan = sg.SupergridPlan(dp_arc=dp_arc, **CONFIGS[name])
    base = run_pass_mode(sg, plan, 1)
    knobs = ({"OGG_QUAD_TARGET_WAVES": "1", "OGG_DPQUAD_TARGET_WAVES": "1", "OGG_MESH_ROWS": "32", "OGG_DPMESH_ROWS": "1",
              "OGG_PASS_LL_WG": "7", "OGG_PASS_LL_WG_MID": "7", "OGG_PASS_LL_WG_SMALL": "7", "OGG_PASS_ORDER": "43210", "OGG_PASS_LL_NT": "0", "OGG_PASS_LL_HELPERS": "0", "OGG_PASS_LL_POOL": "1", "OGG_LL_ROWS_PER_STRIP": "5",
              "OGG_PASS_LL_TABLE": "0"},
             {"OGG_QUAD_TARGET_WAVES": "1000000", "OGG_DPQUAD_TARGET_WAVES": "1000000", "OGG_MESH_ROWS": "1", "OGG_DPMESH_ROWS": "3",
              "OGG_PASS_LL_WG": "1000", "OGG_PASS_LL_WG_MID": "1000", "OGG_PASS_LL_WG_SMALL": "1000", "OGG_PASS_ORDER": "01234",
              "OGG_PASS_LL_HELPERS": "5", "OGG_BP_ROW_COST": "2,1.5,3", "OGG_TOP_RANK_TAIL_US": "20", "OGG_PASS_B_LDS_PAD": "41000"},
             {"OGG_QUAD_TARGET_WAVES": "300", "OGG_DPQUAD_TARGET_WAVES": "200", "OGG_MESH_ROWS": "5", "OGG_DPMESH_ROWS": "5",
              "OGG_PASS_LL_POOL": "1", "OGG_PASS_LL_POOL_HELPERS": "2", "OGG_PASS_LL_WG_SMALL": "40", "OGG_LL_ROWS_PER_STRIP": "32"})
    for world, kn in zip((1, 3, 2), knobs):
        for k, v in kn.items():
            monkeypatch.setenv(k, v)
        got = run_pass_mode(sg, sg.SupergridPlan(dp_arc=dp_arc, **CONFIGS[name]), world)
        for k in kn:
            monkeypatch.delenv(k)
        for f in FIELDS:
            assert np.array_equal(base[f], got[f], equal_nan=False), (f, world, kn)


@pytest.mark.parametrize("order,arc", [(2, "literal"), (2, "chord"), (4, "literal"), (4, "chord")])
def test_supergrid_pass_southern_cap_alone_matches_function_level(hip, order, arc):
    """ogg_supergrid_pass_dev with nothing but a displaced-pole band (no lat-lon bands, no bipolar cap), both quadrature orders and
    both arc forms, band in the middle of the cap: the same bits as ogg_displaced_pole_grid_angle_ws_dev +
    ogg_displaced_pole_metrics_quad_form_ws_dev."""
    import ctypes
    import torch
    from ocean_model_grid_generator_amd import _lib as L
    lib = L.load()
    Ni, Nj, j0, n = 200, 30, 7, 11
    form = {"literal": L.DP_ARC_LITERAL, "chord": L.DP_ARC_CHORD}[arc]
    geo = (-300.0, -78.0, 95.0, 0.27)
    st = torch.cuda.current_stream().cuda_stream

    def fields():
        return {f: torch.full(shp, float("nan"), dtype=torch.float64, device="cuda:0")
                for f, shp in (("x", (n, Ni + 1)), ("y", (n, Ni + 1)), ("angle", (n, Ni + 1)), ("dx", (n, Ni)), ("dy", (n, Ni + 1)), ("area", (n, Ni)))}

    a, b = fields(), fields()
    wsb = int(lib.ogg_dpole_band_workspace_bytes(order, Ni, n))
    ws = torch.zeros(wsb, dtype=torch.uint8, device="cuda:0")
    band = L.DpoleBand()
    band.Ni, band.Nj, band.lon0, band.lat0, band.lon_dp, band.r_dp, band.Re = Ni, Nj, geo[0], geo[1], geo[2], geo[3], 6371.0e3
    band.order, band.arc_form, band.j0, band.n_pt_rows, band.n_cell_rows = order, form, j0, n, n
    for f in a:
        setattr(band, f, a[f].data_ptr())
    band.workspace, band.workspace_bytes = ws.data_ptr(), wsb
    L.call("ogg_supergrid_pass_dev", 0, (L.LatlonBand * 1)(), Ni + 1, geo[0], 360.0, 6371.0e3, 1, None, ctypes.byref(band), None, None, st)
    mws = int(lib.ogg_displaced_pole_grid_workspace_bytes(Ni, n))
    qws = int(lib.ogg_displaced_pole_quad_workspace_bytes(order, Ni, n))
    w1 = torch.zeros(mws, dtype=torch.uint8, device="cuda:0")
    w2 = torch.zeros(qws, dtype=torch.uint8, device="cuda:0")
    L.call("ogg_displaced_pole_grid_angle_ws_dev", Ni, Nj, geo[0], geo[1], geo[2], geo[3], j0, n, b["x"].data_ptr(), b["y"].data_ptr(),
           b["angle"].data_ptr(), w1.data_ptr(), mws, st)
    L.call("ogg_displaced_pole_metrics_quad_form_ws_dev", form, order, Ni, Nj, geo[0], geo[1], geo[2], geo[3], 6371.0e3, j0, n, n,
           b["dx"].data_ptr(), b["dy"].data_ptr(), b["area"].data_ptr(), w2.data_ptr(), qws, st)
    torch.cuda.synchronize()
    for f in a:
        assert not bool(torch.isnan(a[f]).any()), f
        assert torch.equal(a[f], b[f]), (order, arc, f)
    want = orc.displacedPoleCap_metrics_quad(order, Ni, Nj, geo[0], geo[1], geo[2], geo[3])
    for f, w in zip(("dx", "dy", "area"), want):
        w = w[j0:j0 + n]
        assert np.abs(a[f].cpu().numpy() - w).max() <= 2e-12 * Ni * np.abs(w).max(), (order, arc, f)   # dp_quad_rel_tol(Ni) = 4e-10


def test_tripolar_pass_full_size_r8_bitwise(sg):
    """1/8 degree, the benchmark workload: the fused pass against one launch per sub-grid and phase, compared on the device."""
    import torch
    plan = sg.SupergridPlan(8.0)
    for world, rank in ((1, 0), (8, 7), (8, 2)):
        a = sg.Supergrid(plan, rank=rank, world=world, device="cuda:0", halo="recompute", latlon="fused")
        b = sg.Supergrid(plan, rank=rank, world=world, device="cuda:0", halo="recompute", latlon="fused")
        a.launch, b.launch, b.overlap = "pass", "kernels", False
        for s in plan.subs:
            for f in FIELDS:
                a.buf[s.name][f].fill_(float("nan"))
        a.step()
        b.step()
        torch.cuda.synchronize()
        for s in plan.subs:
            for f in FIELDS:
                assert torch.equal(a.buf[s.name][f], b.buf[s.name][f]), (world, rank, s.name, f)
        del a, b


def test_tripolar_pass_argument_errors(sg):
    import ctypes
    import torch
    from ocean_model_grid_generator_amd import _lib as L
    plan = sg.SupergridPlan(1.0)
    g = sg.Supergrid(plan, device="cuda:0")
    g.step()
    bands, arr, cap = g._pass_args[:3]
    st = torch.cuda.current_stream().cuda_stream
    bad = L.BipolarBand.from_buffer_copy(cap)
    bad.order = 7
    with pytest.raises(Exception, match="Uncoded order"):       # the reference's own exception and text (OGG:204)
        L.call("ogg_tripolar_pass_dev", len(bands), arr, plan.Ni + 1, plan.lon0, plan.lenlon, plan.Re, 1, ctypes.byref(bad), st)
    bad = L.BipolarBand.from_buffer_copy(cap)
    bad.workspace = None
    with pytest.raises(L.OggHipError, match="workspace"):
        L.call("ogg_tripolar_pass_dev", len(bands), arr, plan.Ni + 1, plan.lon0, plan.lenlon, plan.Re, 1, ctypes.byref(bad), st)
    bad = L.BipolarBand.from_buffer_copy(cap)
    bad.Ni = plan.Ni + 2
    with pytest.raises(Exception, match="columns"):
        L.call("ogg_tripolar_pass_dev", len(bands), arr, plan.Ni + 1, plan.lon0, plan.lenlon, plan.Re, 1, ctypes.byref(bad), st)
    # no cap at all (a rank without cap rows) and no lat-lon band at all are both fine
    L.call("ogg_tripolar_pass_dev", len(bands), arr, plan.Ni + 1, plan.lon0, plan.lenlon, plan.Re, 1, None, st)
    L.call("ogg_tripolar_pass_dev", 0, arr, plan.Ni + 1, plan.lon0, plan.lenlon, plan.Re, 1, ctypes.byref(cap), st)
    torch.cuda.synchronize()
    # the two-step form: a plan is refused for the same reasons as the one-shot call, and a run needs a plan
    handle = ctypes.c_void_p()
    bad = L.BipolarBand.from_buffer_copy(cap)
    bad.order = 7
    with pytest.raises(Exception, match="Uncoded order"):
        L.call("ogg_supergrid_pass_plan_dev", len(bands), arr, plan.Ni + 1, plan.lon0, plan.lenlon, plan.Re, 1, ctypes.byref(bad), None, ctypes.byref(handle))
    assert not handle.value
    with pytest.raises(L.OggHipError, match="null plan"):
        L.call("ogg_supergrid_pass_run_dev", None, None, None, st)
    # plan once, run twice, against the one-shot call: the same bits
    want = {f: g.buf["BP"][f].clone() for f in FIELDS}
    for f in FIELDS:
        g.buf["BP"][f].fill_(float("nan"))
    L.call("ogg_supergrid_pass_plan_dev", len(bands), arr, plan.Ni + 1, plan.lon0, plan.lenlon, plan.Re, 1, ctypes.byref(cap), None, ctypes.byref(handle))
    for _ in range(2):
        L.call("ogg_supergrid_pass_run_dev", handle, None, None, st)
    torch.cuda.synchronize()
    L.call("ogg_supergrid_pass_plan_destroy", handle)
    for f in FIELDS:
        assert torch.equal(g.buf["BP"][f], want[f]), f


def test_bench_json_contract(tmp_path):
    """bench.py prints ONE JSON line with the driver's keys, the roofline and the CPU baseline (smallest workload)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "r2", "--steps", "4", "--warmup", "1"],
                         capture_output=True, text=True, timeout=600, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 1 and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert 0 < d["ms_per_step_slowest_rank"] <= d["ms_per_step"] * 1.001 and "rehearsal" not in d
    assert d["higher_is_better"] is True and d["unit"] == "cells/s" and "workload" in d["config"]
    assert abs(d["value"] - d["config"]["cells"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and "traffic" in r
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] == 1 and c["value"] > 0 and c["unit"] == "cells/s" and c["sample"]


@pytest.mark.parametrize("name", ["r1_cut2", "r2", "r0.5_dp", "r0.5_latdp", "r0.5_dp+literal"])
@pytest.mark.parametrize("world", [1, 3])
def test_metrics_error_on_device(sg, name, world):
    """Supergrid.metrics_error (five device sums per band + one sum over the ranks) against the reference's function
    (oracle restatement) applied to the whole sub-grid arrays."""
    import torch
    plan = plan_for(sg, name)
    ranks = []
    for r in range(world):
        ranks.append(sg.Supergrid(plan, rank=r, world=world, device="cuda:0", halo="local", peers=ranks, latlon="fused"))
    for g in ranks:
        g.step()
    torch.cuda.synchronize()
    got = ranks[0].metrics_error()
    bands = [g.bands_to_host() for g in ranks]
    for s in plan.subs:
        f = {k: np.concatenate([b[s.name][k] for b in bands]) for k in ("y", "dx", "dy", "area")}
        if s.kind == "bipolar":
            want = orc.metrics_error(f["dx"], f["dy"], f["area"], plan.Ni, s.lat0_bp, 90.0, bipolar=True)
        elif s.kind == "dpole":
            want = orc.metrics_error(f["dx"], f["dy"], f["area"], plan.Ni, s.lat0, -90.0, displaced_pole=ranks[0]._pole_column(s))
        elif s.name == "SC":
            want = orc.metrics_error(f["dx"], f["dy"], f["area"], plan.Ni, f["y"][-1, 0], f["y"][0, 0])
        else:
            want = orc.metrics_error(f["dx"], f["dy"], f["area"], plan.Ni, f["y"][0, 0], f["y"][-1, 0])
        assert len(got[s.name]) == len(want)
        pairs = [(a, b) for a, b in zip(got[s.name], want) if a == a]   # NaN: not estimable without the doughnut rows (dpole)
        assert len(pairs) == len(want) or s.kind == "dpole"
        assert all(abs(a - b) < 1e-9 for a, b in pairs), (s.name, got[s.name], want)
        if s.kind in ("mercator", "latlon", "bipolar"):
            assert max(abs(e) for e in got[s.name]) < 0.05, (s.name, got[s.name])   # the grids are accurate: errors in %


def test_more_ranks_than_rows(sg):
    """Tiny sub-grids: some ranks own no rows of a sub-grid, the top band may own only the fold row."""
    plan = sg.SupergridPlan(0.25, ensure_nj_even=True)
    one = run(sg, plan, 1)
    many = run(sg, plan, 8)
    for f in FIELDS:
        assert np.array_equal(one[f], many[f]), f


def _same_as_function_level(got, ref, name):
    """Bit identity of the pass path with the function-level path, field by field; angle_dx on the bipolar cap excepted: the
    function-level path applies the generic angle_x kernel to the stored mesh, the pass takes the cap's angle inside the mesh kernel
    with cos(phi) in its algebraic form 2u/(1+u^2) -- < 1e-10 degrees apart away from the two pole points."""
    for f in FIELDS:
        assert got[f].shape == ref[f].shape, (name, f, got[f].shape, ref[f].shape)
        if f != "angle_dx":
            assert np.array_equal(got[f], ref[f]), (name, f)
    nbp = got["sub"]["BP"]["x"].shape[0] if "BP" in got["sub"] else 0
    n = got["angle_dx"].shape[0]
    assert np.array_equal(got["angle_dx"][: n - nbp], ref["angle_dx"][: n - nbp]), name
    if nbp:
        d = np.abs(got["angle_dx"][n - nbp:] - ref["angle_dx"][n - nbp:])
        d = np.minimum(d, np.abs(d - 360.0))
        assert np.quantile(d, 0.999) < 1e-10 and np.median(d) < 1e-12, (name, float(np.quantile(d, 0.999)), float(d.max()))


_GOLDEN_CONFIGS = sorted(json.load(open(os.path.join(GOLD, "ref_hashes.json")))["configs"])


@pytest.mark.parametrize("name", _GOLDEN_CONFIGS)
def test_main_pass_path_matches_function_level_path(hip, name, tmp_path):
    """main() (one device-resident pass, fields streamed from HBM into the file) against main(path="functions") (the reference's own
    sequence of host-array calls) on every golden configuration: the whole flag surface -- enhanced_equatorial (explicit axis,
    axis_kind = 2), match_dy, south cuts by row and by angle, latitude overrides, no_south_cap, ensure_nj_even, both displaced-pole
    spellings.  The file the pass path writes holds the same arrays."""
    import ocean_model_grid_generator_amd.ocean_grid_generator as ogg
    from scipy.io import netcdf_file
    cfg = json.load(open(os.path.join(GOLD, "ref_hashes.json")))["configs"][name]
    flags = dict(cfg["flags"])
    out = str(tmp_path / "g.nc")
    got = ogg.main(gridfilename=out, no_changing_meta=True, return_arrays=True, **flags)
    for f, shp in cfg["shapes"].items():
        assert list(got[f].shape) == shp, (f, got[f].shape, shp)          # shapes pinned by the reference run
    ref = ogg.main(gridfilename=None, no_changing_meta=True, return_arrays=True, path="functions", **flags)
    ref = dict(ref, sub={k: dict(zip(FIELDS, v)) for k, v in ref["sub"].items()})
    _same_as_function_level(got, ref, name)
    nc = netcdf_file(out, "r", mmap=True)
    assert list(nc.variables.keys()) == ["tile", "y", "x", "dy", "dx", "area", "angle_dx"] and nc.version_byte == 2
    for f in FIELDS:
        assert np.array_equal(nc.variables[f][:], got[f]), (name, f)
    del f
    nc.close()


@pytest.mark.parametrize("seed", range(40))
def test_both_orchestrators_agree_on_random_flag_sets(hip, seed):
    """main() drives the device-resident pass from SupergridPlan's restatement of the reference's size logic (OGG:969-1313);
    main(path="functions") walks the reference's own sequence and takes every size from the arrays it has just computed, as the
    reference does.  Two restatements of the same logic: on 40 seeded random flag sets (the generator of tests/test_plan_cpu.py, which
    holds the plan to the ORACLE on the CPU: poles, doughnut fractions, cuts by row and by angle, match_dy, latitude overrides, --grids
    subsets, parity) they must produce the same arrays bit for bit -- or raise the same text."""
    import contextlib
    import io
    import ocean_model_grid_generator_amd.ocean_grid_generator as ogg
    from test_plan_cpu import _flags
    flags = _flags(seed)
    out = []
    for path in ("pass", "functions"):
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                out.append(ogg.main(gridfilename=None, no_changing_meta=True, return_arrays=True, path=path, **flags))
        except SystemExit as exc:
            out.append(("exit", str(exc.code)))
        except Exception as exc:   # noqa: BLE001 -- the text is what is compared
            out.append(("raise", str(exc)))
    got, ref = out
    if isinstance(got, tuple) or isinstance(ref, tuple):
        if isinstance(got, tuple) and isinstance(ref, tuple) and got[1] != ref[1]:
            # both refuse, for different reasons: the plan refuses up front what the walk only meets later (a cut that needs a cap
            # which is not there) -- the kind must still be the same
            assert got[0] == ref[0], (flags, got, ref)
        else:
            assert got == ref, (flags, got if isinstance(got, tuple) else "arrays", ref if isinstance(ref, tuple) else "arrays")
        return
    ref = dict(ref, sub={k: dict(zip(FIELDS, v)) for k, v in ref["sub"].items()})
    _same_as_function_level(got, ref, str(flags))


@pytest.mark.parametrize("flags", [
    dict(inverse_resolution=1.0, ensure_nj_even=True, south_cutoff_row=2),      # the reference's own guard rejects the even one of
    dict(inverse_resolution=1.0, ensure_nj_even=True, south_cutoff_row=3),      # these two (OGG:1434): both paths must agree
    dict(inverse_resolution=1.0, ensure_nj_even=True, south_cutoff_row=30),     # consumes the whole southern cap and part of SO
    dict(inverse_resolution=0.5, ensure_nj_even=True, r_dp=0.2, south_cutoff_row=4),
    dict(inverse_resolution=0.5, ensure_nj_even=True, r_dp=0.2, south_cutoff_ang=-80.5),
    dict(inverse_resolution=1.0, grids=["bipolar", "mercator", "so"]),
    dict(inverse_resolution=0.5, bipolar_lower_lat=66.0, mercator_upper_lat=66.0, mercator_lower_lat=-65.0),
])
def test_plan_cuts_and_parity_bumps_follow_the_reference(sg, flags):
    """SupergridPlan.south_cut / stitch against the oracle's restatement of OGG:1268-1313 incl. the --ensure_nj_even parity bumps
    (SC: (n_SC - jcut) % 2 == 0 -> jcut += 1; SO: (n_cell_SO - jcut_SO - 1) % 2 == 0 -> jcut_SO += 1) and the final guards."""
    import ocean_model_grid_generator_amd.ocean_grid_generator as ogg
    flags = dict(flags)
    r = flags.pop("inverse_resolution")
    try:
        want = orc.make_supergrid(r, skip_doughnut_rows=True, **flags)
    except Exception as exc:   # the reference's own guards reject this flag set: ours must raise the same text
        with pytest.raises(Exception) as ei:
            ogg.main(r, gridfilename=None, no_changing_meta=True, return_arrays=True, **flags)
        assert str(ei.value) == str(exc)
        return
    got = ogg.main(r, gridfilename=None, no_changing_meta=True, return_arrays=True, **flags)
    plan = sg.SupergridPlan(r, **flags)
    if flags.get("south_cutoff_ang", -90.0) <= -90.0:
        assert plan.nyp == want["y"].shape[0]
    _check_supergrid(got, want, "cuts_%s" % "_".join("%s" % v for v in flags.values()))


def test_full_size_properties_r8(sg):
    """BASELINE config 2 (1/8 degree) at full size, checked through size-independent properties: the analytic-sphere
    self-check of the reference (OGG:732-770) per sub-grid, continuity across the stitch joints, monotone latitude
    along the symmetry meridian, exact special values."""
    plan = sg.SupergridPlan(8.0)
    assert (plan.nyp, plan.Ni + 1) == (4393, 5761) and plan.cells == 4392 * 5760
    g = run(sg, plan, latlon="fused")
    g2 = run(sg, plan, latlon="stencil")
    for f in FIELDS:
        assert np.array_equal(g[f], g2[f]), f
    Ni = plan.Ni
    sub = g["sub"]
    m = sub["Merc"]
    err = orc.metrics_error(m["dx"], m["dy"], m["area"], Ni, m["y"][0, 0], m["y"][-1, 0])
    assert max(abs(e) for e in err) < 1e-10, err
    b = sub["BP"]
    lat0_bp = m["y"][-1, Ni // 4]
    err = orc.metrics_error(b["dx"], b["dy"], b["area"], Ni, lat0_bp, 90.0, bipolar=True)
    assert max(abs(e) for e in err) < 1e-9, err
    s = sub["SO"]
    err = orc.metrics_error(s["dx"], s["dy"], s["area"], Ni, s["y"][0, 0], s["y"][-1, 0])
    assert max(abs(e) for e in err) < 1e-10, err
    c = sub["SC"]
    err = orc.metrics_error(c["dx"], c["dy"], c["area"], Ni, c["y"][-1, 0], c["y"][0, 0])
    assert max(abs(e) for e in err) < 1e-10, err
    # total area of the sphere
    total = g["area"].sum()
    assert abs(total / (4 * np.pi * 6371.0e3 ** 2) - 1) < 1e-12
    ycol = g["y"][:, Ni // 4]
    assert np.all(np.diff(ycol) > 0) and ycol[-1] == 90.0 and ycol[0] == -90.0
    assert np.searchsorted(ycol, 0.0) % 2 == 1
    # the bipolar h_j vanishes identically on the meridians i = 0, Ni/2, Ni (alpha2 == 1, OGG:81-84) -- in the
    # reference too -- so dy is exactly 0 there and positive everywhere else
    zc = np.unique(np.argwhere(g["dy"] == 0)[:, 1])
    assert np.all(g["dx"] > 0) and np.all(g["dy"] >= 0) and set(zc.tolist()) <= {0, Ni // 2, Ni} and np.all(g["area"] > 0)
    assert np.all(np.abs(g["angle_dx"]) <= 180.0)
    # dy is continuous across the Mercator / bipolar joint along the symmetry meridian (OGG:1027-1030)
    jM = sub["SC"]["y"].shape[0] + sub["SO"]["y"].shape[0] + sub["Merc"]["y"].shape[0] - 3
    col = g["dy"][:, Ni // 4]
    assert abs(col[jM] / col[jM - 1] - 1) < 0.02


def test_reference_own_r8_test_configuration_through_main_vs_oracle(hip):
    """The reference's own 1/8 degree test configuration (t/test_ocean_grid_gen.py:176-185, extras/Makefile:40-41):
    -r 8 --r_dp 0.2 --south_cutoff_row 5 --match_dy bp so p125sc --ensure_nj_even, 4481 x 5761, through the drop-in main() (plan, one
    device-resident pass, stitching) against the oracle, every element; shapes and sub-grid sizes as recorded from the unmodified reference
    (tests/golden/ref_hashes.json: r8_p125; tests/golden/make_golden.py asserted the oracle bit-identical to the reference on it, and
    where this host's numpy / libm fingerprint equals the recorded one the oracle's sha256 per field is checked again here)."""
    import hashlib
    import ocean_model_grid_generator_amd.ocean_grid_generator as ogg
    from test_oracle_golden import SAME_PLATFORM
    rec = json.load(open(os.path.join(GOLD, "ref_hashes.json")))
    cfg = rec["configs"]["r8_p125"]
    flags = dict(cfg["flags"])
    got = ogg.main(gridfilename=None, no_changing_meta=True, return_arrays=True, **flags)
    for f, shp in cfg["shapes"].items():
        assert list(got[f].shape) == shp, (f, got[f].shape, shp)
    assert got["x"].shape == (4481, 5761)
    r = flags.pop("inverse_resolution")
    want = orc.make_supergrid(r, skip_doughnut_rows=True, **flags)
    if SAME_PLATFORM:
        for f in FIELDS:
            assert hashlib.sha256(np.ascontiguousarray(want[f], dtype=np.float64).tobytes()).hexdigest() == cfg["sha256"][f], f
    _check_supergrid(got, want, "r8_p125")


@pytest.mark.parametrize("r,world,rank", [(16, 1, 0), (16, 8, 7), (32, 1, 0)])
def test_full_size_properties_r16_on_device(sg, r, world, rank):
    """BASELINE config 5 (1/16 degree, 101 M cells, 4.9 GB of fields) through the fused pass, checked where the fields are:
    the reference's analytic-sphere self-check from device sums, signs, monotone latitude, the joints.  And once at 1/32 degree --
    beyond BASELINE.json: 405 M cells, 19.4 GB of fields in one GPU's HBM -- for the index arithmetic (1464 look-back strips per row, 3.5e8
    cells in the cap) and the sizing."""
    import torch
    plan = sg.SupergridPlan(float(r))
    k = r // 16
    assert (plan.nyp, plan.Ni + 1) == (8784 * k + 1, 11520 * k + 1) and plan.cells == 8784 * k * 11520 * k
    g = sg.Supergrid(plan, rank=rank, world=world, device="cuda:0", halo="recompute")
    assert g.launch == "pass"
    g.step()
    torch.cuda.synchronize()
    q = plan.Ni // 4
    for s in plan.subs:
        b = g.buf[s.name]
        if b["n"] == 0:
            continue
        assert bool(torch.all(b["dx"] > 0)) and bool(torch.all(b["dy"] >= 0)) and bool(torch.all(b["area"] > 0)), s.name
        assert bool(torch.all(torch.abs(b["angle_dx"]) <= 180.0)), s.name
        ycol = b["y"][: b["n"], q]
        assert bool(torch.all(ycol[1:] > ycol[:-1])), s.name
    if world == 1:
        err = g.metrics_error()
        for name, e in err.items():
            assert max(abs(v) for v in e) < 1e-9, (name, e)
        sums = g.metrics_sums().cpu().numpy()
        assert abs(sums[:, 0].sum() / (4 * np.pi * 6371.0e3 ** 2) - 1) < 1e-12
        ys = {s.name: (float(g.buf[s.name]["y"][0, q]), float(g.buf[s.name]["y"][g.buf[s.name]["n"] - 1, q])) for s in plan.subs}
        assert ys["SC"][0] == -90.0 and ys["BP"][1] == 90.0
        assert ys["SC"][1] == ys["SO"][0] and ys["SO"][1] == ys["Merc"][0] and ys["Merc"][1] == ys["BP"][0]


def test_full_size_r8_pass_vs_oracle(sg):
    """The headline workload (1/8 degree, 25.3 M cells) through the fused pass against the numpy oracle, every field, every
    element.  Bounds: test_gpu_parity.py's (module docstring there)."""
    import torch
    from test_gpu_parity import record
    plan = sg.SupergridPlan(8.0)
    g = sg.Supergrid(plan, device="cuda:0")
    assert g.launch == "pass"
    g.step()
    torch.cuda.synchronize()
    got = sg.stitch(plan, [g.bands_to_host()])
    del g
    want = orc.make_supergrid(8.0, skip_doughnut_rows=True)
    rep = {}
    for f in FIELDS:
        assert got[f].shape == want[f].shape, f
    d = np.abs(got["y"] - want["y"])
    rep["y_max"] = float(d.max())
    rep["y_max_excl_bp_poles"] = float(np.sort(d.ravel())[-5])          # the two pole points (and their duplicate columns)
    assert rep["y_max_excl_bp_poles"] < 1e-12 and rep["y_max"] < 1e-6
    d = np.abs(got["x"] - want["x"])
    rep["x_max"] = float(d.max())
    rep["x_frac_gt_1e12"] = float(np.mean(d > 1e-12))
    assert rep["x_max"] < 2e-11 and rep["x_frac_gt_1e12"] < 1e-5                # next to the cap's symmetry meridians only
    # dx, dy of the lat-lon sub-grids are differences of coordinates that themselves differ by ~1e-14 degrees (ocml vs the host
    # libm in atan(sinh)): 1e-14 deg x 111 km/deg = 1e-9 m on a 7 km cell; area likewise (one ulp of sin moves it by 3e-5 m^2)
    for f, (a, r) in (("dx", (2e-8, 5e-14)), ("dy", (2e-8, 5e-14)), ("area", (0.0, 1.2e-11))):
        d = np.abs(got[f] - want[f])
        rep[f + "_max_abs"] = float(d.max())
        rep[f + "_max_rel"] = float((d / np.maximum(np.abs(want[f]), 1e-300))[want[f] != 0].max())
    record("full_r8_pass_vs_oracle", **rep)
    for f, (a, r) in (("dx", (2e-8, 5e-14)), ("dy", (2e-8, 5e-14)), ("area", (0.0, 1.2e-11))):   # area: 10 x the measured 1.2e-12 / 9.8e-6 m^2
        assert np.all(np.abs(got[f] - want[f]) <= a + r * np.abs(want[f])), (f, rep)
    assert rep["area_max_abs"] < 1e-4, rep
    d = np.abs(got["angle_dx"] - want["angle_dx"])
    d = np.minimum(d, np.abs(d - 360.0))
    rep["angle_p999"] = float(np.quantile(d, 0.999))
    rep["angle_frac_gt_1e9"] = float(np.mean(d > 1e-9))
    assert rep["angle_p999"] < 1e-10 and rep["angle_frac_gt_1e9"] < 1e-5
    record("full_r8_pass_vs_oracle", **rep)


def test_full_size_r16_pass_vs_oracle(sg):
    """BASELINE config 5 (1/16 degree, 101 M cells, 4.9 GB of fields) through the fused pass against the numpy oracle, every field,
    every element (the oracle needs about a minute and ~15 GB on the host).  The bounds of the 1/8 degree test, except where the
    resolution enters: the MIDAS area of a lat-lon cell is 4x smaller and its one-ulp-of-sin noise with it (measured 7.4e-6 m^2), the
    cells at the cap's two pole points reach 6e-14 relative in area with every cell literal (test_bipolar_quad_full_size_r16_by_zone)."""
    import gc
    import torch
    from test_gpu_parity import record
    plan = sg.SupergridPlan(16.0)
    g = sg.Supergrid(plan, device="cuda:0")
    assert g.launch == "pass"
    g.step()
    torch.cuda.synchronize()
    got = sg.stitch(plan, [g.bands_to_host()])
    got.pop("sub", None)
    del g
    gc.collect()
    want = orc.make_supergrid(16.0, skip_doughnut_rows=True)
    want.pop("sub", None)
    rep = {}
    for f in FIELDS:
        assert got[f].shape == want[f].shape == ((8785 if f in ("x", "y", "dx", "angle_dx") else 8784), (11521 if f in ("x", "y", "dy", "angle_dx") else 11520)), f
    d = np.abs(got["y"] - want["y"])
    rep["y_max"] = float(d.max())
    rep["y_max_excl_bp_poles"] = float(np.sort(d[-3:].ravel())[-5]) if d[:-3].max() < 1e-12 else float(d[:-3].max())
    assert d[:-3].max() < 1e-12 and rep["y_max_excl_bp_poles"] < 1e-12 and rep["y_max"] < 1e-6
    d = np.abs(got["x"] - want["x"])
    rep["x_max"] = float(d.max())
    rep["x_frac_gt_1e12"] = float(np.mean(d > 1e-12))
    assert rep["x_max"] < 2e-11 and rep["x_frac_gt_1e12"] < 1e-5
    for f, (a, r) in (("dx", (2e-8, 5e-14)), ("dy", (2e-8, 5e-14)), ("area", (0.0, 1.2e-11))):
        d = np.abs(got[f] - want[f])
        rep[f + "_max_abs"] = float(d.max())
        nz = want[f] != 0
        rep[f + "_max_rel"] = float((d[nz] / np.abs(want[f][nz])).max())
        assert np.all(d <= a + r * np.abs(want[f])), (f, rep)
        del d, nz
    assert rep["area_max_abs"] < 1e-4, rep
    d = np.abs(got["angle_dx"] - want["angle_dx"])
    d = np.minimum(d, np.abs(d - 360.0))
    rep["angle_p999"] = float(np.quantile(d, 0.999))
    rep["angle_frac_gt_1e9"] = float(np.mean(d > 1e-9))
    assert rep["angle_p999"] < 1e-10 and rep["angle_frac_gt_1e9"] < 1e-5
    record("full_r16_pass_vs_oracle", **rep)


def test_full_size_r8_latdp_pass_vs_oracle(sg):
    """BASELINE config 4 (1/8 degree with the displaced south pole, --lat_dp -85.85) at full size against the oracle (the
    oracle needs ~40 s for the cap's finite-difference quadrature), in BOTH arc forms of the cap's quadrature.  The differences
    of the two forms from the oracle, and from each other, go to the parity report (bench.py quotes them from
    profiles/dp_parity.json); the chord form must stay within 5e-9 of the oracle and of the literal form."""
    import torch
    from test_gpu_parity import record, dp_quad_rel_tol
    flags = dict(lon_dp=80.0, lat_dp=-85.85)
    want = orc.make_supergrid(8.0, skip_doughnut_rows=True, **flags)
    rep, caps = {}, {}
    for form in ("literal", "chord"):
        plan = sg.SupergridPlan(8.0, dp_arc=form, **flags)
        g = sg.Supergrid(plan, device="cuda:0")
        assert g.launch == "pass"
        g.step()
        torch.cuda.synchronize()
        got = sg.stitch(plan, [g.bands_to_host()])
        del g
        nsc = plan.subs[0].nj1 - 1      # cell rows of the displaced-pole cap
        assert plan.subs[0].kind == "dpole"
        for f in FIELDS:
            assert got[f].shape == want[f].shape, f
        rep["y_max"] = float(np.abs(got["y"] - want["y"]).max())
        d = np.abs(got["x"] - want["x"])
        rep["x_max"] = float(np.minimum(d, np.abs(d - 360.0)).max())
        assert rep["y_max"] < 1e-12 and rep["x_max"] < 2e-11
        caps[form] = {f: got[f][:nsc] for f in ("dx", "dy", "area")}
        for f in ("dx", "dy", "area"):
            d = np.abs(got[f] - want[f])
            rel = d / np.maximum(np.abs(want[f]), 1e-300)
            rep["%s_cap_max_rel_%s" % (f, form)] = float(rel[:nsc][want[f][:nsc] != 0].max())
            rep["%s_cap_max_abs_%s" % (f, form)] = float(d[:nsc].max())
            rep[f + "_rest_max_rel"] = float(rel[nsc:][want[f][nsc:] != 0].max())
        del got
    for f in ("dx", "dy", "area"):
        rep[f + "_cap_chord_vs_literal_max_rel"] = float((np.abs(caps["chord"][f] - caps["literal"][f]) / np.abs(caps["literal"][f])).max())
    record("full_r8_latdp_pass_vs_oracle", **rep)
    tol = dp_quad_rel_tol(plan.Ni)   # 1.15e-8: 10x the measured difference of either form
    for f in ("dx", "dy", "area"):
        for form in ("literal", "chord"):
            assert rep["%s_cap_max_rel_%s" % (f, form)] < min(tol, 5e-9 if form == "chord" else tol), (f, form, rep)
        assert rep[f + "_cap_chord_vs_literal_max_rel"] < 5e-9, (f, rep)
        assert rep[f + "_rest_max_rel"] < 5e-12, (f, rep)
