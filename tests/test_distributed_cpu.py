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
