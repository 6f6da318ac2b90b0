"""World-size-2 and -3 tests of the multi-GPU plumbing on the CPU (gloo): the latitude-band partition and the one-row
halo exchange of supergrid.Supergrid run unchanged on host tensors; the oracle stands in for the kernels INSIDE THIS
TEST ONLY to show that band + halo reproduce the unsharded result bit for bit."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import ogg_oracle as orc


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _plan(sg_mod, r, **kw):
    Ni = int(r * 2 * 360)
    phi_s, phi_n = (-68.0, 65.0) if r == 2 else (-66.85954725, 64.05895973)
    y0, y1 = orc.mercator_y_star(Ni, phi_s, phi_n, True, kw.get("ensure_nj_even", False))
    return sg_mod.SupergridPlan(r, mercator_axis=(y0, orc.phi_mercator(Ni, np.arange(y0, y1 + 1))), **kw)


def _worker(rank, world, port, r, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import ocean_model_grid_generator_amd.supergrid as sg_mod
        plan = _plan(sg_mod, r, ensure_nj_even=True)
        g = sg_mod.Supergrid(plan, rank=rank, world=world, device="cpu", halo="rccl", latlon="stencil")
        full = orc.make_supergrid(r, ensure_nj_even=True)["sub"]
        ok = True
        for s in plan.subs:
            b = g.buf[s.name]
            x_full, y_full = full[s.name][0], full[s.name][1]
            assert x_full.shape[0] == s.nj1
            b["x"][: b["n"]] = torch.from_numpy(np.ascontiguousarray(x_full[b["lo"]:b["hi"]]))
            b["y"][: b["n"]] = torch.from_numpy(np.ascontiguousarray(y_full[b["lo"]:b["hi"]]))
            if b["needs_halo"]:
                b["x"][b["n"]] = float("nan")
                b["y"][b["n"]] = float("nan")
        g.exchange_halo()
        for s in plan.subs:
            b = g.buf[s.name]
            if s.kind not in ("mercator", "latlon") or b["n"] == 0:
                assert not b["needs_halo"]
                continue
            x, y = b["x"].numpy(), b["y"].numpy()
            if b["needs_halo"]:
                ok &= np.array_equal(x[b["n"]], full[s.name][0][b["hi"]]) and np.array_equal(y[b["n"]], full[s.name][1][b["hi"]])
            dx, dy, area = orc.generate_grid_metrics_MIDAS(x, y) if x.shape[0] > 1 else (None, None, None)
            if dx is not None:
                ok &= np.array_equal(dx[: b["n"]], full[s.name][2][b["lo"]:b["hi"]])
                ok &= np.array_equal(dy[: b["n_cell"]], full[s.name][3][b["lo"]:b["lo"] + b["n_cell"]])
                ok &= np.array_equal(area[: b["n_cell"]], full[s.name][4][b["lo"]:b["lo"] + b["n_cell"]])
        # metrics_error of a band-sharded run: per-band sums (formed here on the host from the oracle's fields, on a GPU by
        # ogg_metrics_sums_dev), one all-reduce, the reference's formulas -- against the reference's function on whole sub-grids
        import ocean_model_grid_generator_amd.ocean_grid_generator as ogg
        sums = torch.zeros((len(plan.subs), 7), dtype=torch.float64)
        for k, s in enumerate(plan.subs):
            b = g.buf[s.name]
            if b["n"] == 0:
                continue
            y_full, dx_f, dy_f, ar_f = (full[s.name][i] for i in (1, 2, 3, 4))
            ca, cb = ogg.metrics_error_columns(plan.Ni, g._pole_column(s))
            cells = slice(b["lo"], b["lo"] + b["n_cell"])
            sums[k, 0] = ar_f[cells].sum()
            sums[k, 1] = dy_f[cells, ca].sum()
            sums[k, 2] = dy_f[cells, cb].sum() if cb >= 0 else 0.0
            if b["lo"] == 0:
                sums[k, 3], sums[k, 5] = dx_f[0].sum(), y_full[0, 0]
            if b["hi"] == s.nj1:
                sums[k, 4], sums[k, 6] = dx_f[-1].sum(), y_full[-1, 0]
        got = g.metrics_error(sums=sums)
        for s in plan.subs:
            y_full, dx_f, dy_f, ar_f = (full[s.name][i] for i in (1, 2, 3, 4))
            if s.kind == "bipolar":
                want = orc.metrics_error(dx_f, dy_f, ar_f, plan.Ni, s.lat0_bp, 90.0, bipolar=True)
            elif s.name == "SC":
                want = orc.metrics_error(dx_f, dy_f, ar_f, plan.Ni, y_full[-1, 0], y_full[0, 0])
            else:
                want = orc.metrics_error(dx_f, dy_f, ar_f, plan.Ni, y_full[0, 0], y_full[-1, 0])
            ok &= len(got[s.name]) == len(want) and all(abs(a - b) < 1e-9 for a, b in zip(got[s.name], want))
        q.put((rank, bool(ok)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_halo_exchange_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, 0.25, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    results = dict(q.get(timeout=10) for _ in range(world))
    assert results == {r: True for r in range(world)}


def test_band_partition_covers_every_row_once():
    import ocean_model_grid_generator_amd.supergrid as sg_mod
    for n in (1, 2, 7, 193, 2801):
        for world in (1, 2, 3, 8, 16):
            rows = []
            for r in range(world):
                lo, hi = sg_mod.band(n, r, world)
                assert 0 <= lo <= hi <= n
                rows += list(range(lo, hi))
            assert rows == list(range(n))


def test_plan_shapes_match_reference_runs():
    """Size logic of the pipeline plan against the shapes recorded from the reference (no GPU needed)."""
    import json
    import ocean_model_grid_generator_amd.supergrid as sg_mod
    cfgs = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_hashes.json")))["configs"]
    for name in ("r0.25_even", "r1_cut2", "r2", "r4_om4", "r0.5_dp", "r0.5_latdp", "r8", "r8_latdp"):
        flags = dict(cfgs[name]["flags"])
        plan = _plan(sg_mod, flags.pop("inverse_resolution"), **flags)
        assert [plan.nyp, plan.Ni + 1] == cfgs[name]["shapes"]["x"], name
        assert plan.cells == cfgs[name]["shapes"]["area"][0] * cfgs[name]["shapes"]["area"][1]


def test_split_times_give_every_rank_the_same_edges_and_cover_every_row():
    """The band split with a MEASURED pair (tail_us, pass_us) (SupergridPlan.calibrate_split -> set_split_times): whatever the two
    numbers, every sub-grid's rows are covered exactly once, the last rank's share shrinks as the fix-up launch grows, and a tail
    that is more than half of a rank's share switches the correction off (equal shares) instead of starving the last rank."""
    import ocean_model_grid_generator_amd.supergrid as sg_mod
    rng = np.random.default_rng(4)
    for r in (0.5, 2, 8):
        plan = _plan(sg_mod, r)
        assert plan.split_times and "UNVALIDATED" in plan.split_times["source"]
        for world in (2, 4, 8):
            for _ in range(6):
                tail, whole = float(rng.uniform(1, 40)), float(rng.uniform(20, 5000))
                plan.set_split_times(tail, whole, "test")
                for s in plan.subs:
                    rows = []
                    for k in range(world):
                        lo, hi = sg_mod.Supergrid.rows_of(s, k, world)
                        assert 0 <= lo <= hi <= s.nj1
                        rows += list(range(lo, hi))
                    assert rows == list(range(s.nj1)), (r, world, s.name)
                    last = sg_mod.Supergrid.rows_of(s, world - 1, world)
                    if world * tail / whole > 0.5 and getattr(s, "row_cost", None) is None:
                        assert last == sg_mod.band(s.nj1, world - 1, world)
    plan = _plan(sg_mod, 8)
    merc = next(s for s in plan.subs if s.name == "Merc")
    shares = []
    for tail in (0.0001, 5.0, 10.0, 15.0):
        plan.set_split_times(tail, 250.0, "test")
        lo, hi = sg_mod.Supergrid.rows_of(merc, 7, 8)
        shares.append(hi - lo)
    assert shares == sorted(shares, reverse=True) and shares[0] > shares[-1]


def _calib_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import ocean_model_grid_generator_amd.supergrid as sg_mod
        plan = _plan(sg_mod, 2)
        if rank == 0:
            # no GPU here: rank 0's measurement is replaced by fixed numbers; the broadcast and what follows are the real code
            t = torch.tensor([7.25, 123.5, 0.875], dtype=torch.float64)
            dist.broadcast(t, src=0)
            plan.set_split_times(float(t[0]), float(t[1]), "test", top_capacity=(world, float(t[2])))
        else:
            plan.calibrate_split("cpu", rank=rank, world=world)
        q.put((rank, plan.split_times["tail_us"], plan.split_times["pass_us"],
               [sg_mod.Supergrid.rows_of(s, k, world) for s in plan.subs for k in range(world)]))
    finally:
        dist.destroy_process_group()


def test_calibrated_split_is_broadcast_from_rank_0():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    world = 2
    procs = [ctx.Process(target=_calib_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=10) for _ in range(world))
    assert res[0][1:] == res[1][1:] and res[0][1] == 7.25 and res[0][2] == 123.5
    # the measured share took precedence over the model's 1 - 2 * 7.25 / 123.5: the last rank of two holds 0.875 / 1.875 of the Mercator rows
    import ocean_model_grid_generator_amd.supergrid as sg_mod
    plan = _plan(sg_mod, 2)
    plan.set_split_times(7.25, 123.5, "test", top_capacity=(2, 0.875))
    merc = next(s for s in plan.subs if s.name == "Merc")
    lo, hi = sg_mod.Supergrid.rows_of(merc, 1, 2)
    assert abs((hi - lo) / merc.nj1 - 0.875 / 1.875) < 2.0 / merc.nj1
    assert sg_mod.Supergrid.rows_of(merc, 2, 3) != sg_mod.Supergrid.rows_of(merc, 1, 2)    # another world size: the model again


def _refine_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import ocean_model_grid_generator_amd.supergrid as sg_mod
        plan = _plan(sg_mod, 2)
        plan.set_split_times(6.0, 120.0, "test", top_capacity=(world, 0.9))
        before = [sg_mod.Supergrid.rows_of(s, k, world) for s in plan.subs for k in range(world)]
        # every rank "measures" its own share (no GPU here: a made-up time, the last rank 20 % slower) and the list is all-gathered
        mine = 40.0 + rank if rank < world - 1 else 50.0
        gathered = [None] * world
        dist.all_gather_object(gathered, mine)
        changed = plan.refine_split(gathered, world)
        after = [sg_mod.Supergrid.rows_of(s, k, world) for s in plan.subs for k in range(world)]
        q.put((rank, changed, before, after, plan.split_times["self_calibration"][-1]))
    finally:
        dist.destroy_process_group()


def test_every_rank_rebalances_the_split_from_the_gathered_times():
    """Per-rank self-calibration (bench.py at N > 1): the ranks all-gather their own times per pass and each applies
    SupergridPlan.refine_split to the same list -- the same new edges on every rank, every rank's share scaled by mean(T) / T_rank
    (relative to rank 0's), every row still covered once; inside the dead band nothing moves."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    world = 3
    procs = [ctx.Process(target=_refine_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    res = sorted(q.get(timeout=10) for _ in range(world))
    assert all(r[1] for r in res) and res[0][2:] == res[1][2:] == res[2][2:]
    rec = res[0][4]
    assert rec["per_rank_us"] == [40.0, 41.0, 50.0] and rec["share_of_last_rank_before"] == 0.9
    assert abs(rec["share_of_last_rank_after"] - 0.9 * 40.0 / 50.0) < 1e-12
    assert np.allclose(rec["shares_after"], [1.0, 40.0 / 41.0, 0.9 * 40.0 / 50.0], atol=1e-5) and rec["shares_before"] == [1.0, 1.0, 0.9]
    import ocean_model_grid_generator_amd.supergrid as sg_mod
    plan = _plan(sg_mod, 2)
    plan.set_split_times(6.0, 120.0, "test", top_capacity=(3, 0.9))
    merc = next(s for s in plan.subs if s.name == "Merc")
    n_before = np.diff(sg_mod.Supergrid.rows_of(merc, 2, 3))[0]
    assert plan.refine_split([40.0, 41.0, 50.0], 3)
    n_after = np.diff(sg_mod.Supergrid.rows_of(merc, 2, 3))[0]
    assert n_after < n_before
    for s in plan.subs:
        rows = []
        for k in range(3):
            lo, hi = sg_mod.Supergrid.rows_of(s, k, 3)
            rows += list(range(lo, hi))
        assert rows == list(range(s.nj1))
    # the same at the world size of the node the bench is meant for: eight ranks, every row of every sub-grid of the 1/8 degree grid once
    plan8 = _plan(sg_mod, 8)
    plan8.set_split_times(6.5, 213.0, "test")
    assert plan8.refine_split([31.0, 31.4, 31.2, 31.1, 31.3, 31.0, 31.2, 34.0], 8)
    assert plan8.split_times["top_capacity"]["world"] == 8
    for s in plan8.subs:
        rows = []
        for k in range(8):
            lo, hi = sg_mod.Supergrid.rows_of(s, k, 8)
            rows += list(range(lo, hi))
        assert rows == list(range(s.nj1)), s.name
    # a rank in the MIDDLE that is slower than the cost model says (the rows under the guarded ones: 34.3 us against 31.6 in the
    # one-GPU rehearsal of the 1/8 degree grid) hands rows to the others; a second step from balanced times changes nothing
    plan8 = _plan(sg_mod, 8)
    plan8.set_split_times(6.5, 192.0, "test", top_capacity=(8, 0.8))
    merc8 = next(s for s in plan8.subs if s.name == "Merc")
    n6 = np.diff(sg_mod.Supergrid.rows_of(merc8, 6, 8))[0]
    n7 = np.diff(sg_mod.Supergrid.rows_of(merc8, 7, 8))[0]
    assert plan8.refine_split([30.4, 31.7, 31.7, 31.6, 31.7, 31.6, 34.3, 31.6], 8)
    assert np.diff(sg_mod.Supergrid.rows_of(merc8, 6, 8))[0] < n6 and abs(np.diff(sg_mod.Supergrid.rows_of(merc8, 7, 8))[0] - n7) <= 0.02 * n7
    assert plan8.split_times["rank_capacity"]["world"] == 8 and len(plan8.split_times["rank_capacity"]["shares"]) == 8
    for s in plan8.subs:
        rows = []
        for k in range(8):
            lo, hi = sg_mod.Supergrid.rows_of(s, k, 8)
            rows += list(range(lo, hi))
        assert rows == list(range(s.nj1)), s.name
    shares = list(plan8.split_times["rank_capacity"]["shares"])
    assert not plan8.refine_split([31.8, 31.9, 31.7, 31.8, 31.9, 31.8, 32.0, 31.7], 8)
    assert plan8.split_times["rank_capacity"]["shares"] == shares
    # within 3 %: the split stays (and the record says so)
    assert not plan.refine_split([40.0, 41.0, 41.2], 3)
    assert plan.split_times["self_calibration"][-1]["share_of_last_rank_after"] == plan.split_times["self_calibration"][-1]["share_of_last_rank_before"]
    assert len(plan.split_times["self_calibration"]) == 2

