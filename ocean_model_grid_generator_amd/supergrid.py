"""Device-resident, latitude-band-sharded supergrid pass (the benchmark / multi-GPU path).

One process per GPU.  Every rank owns a contiguous band of rows of EACH sub-grid (southern cap, Southern Ocean,
Mercator, bipolar cap): cost per row differs by ~100x between the lat-lon sub-grids and the caps, so cutting the
stitched grid into contiguous slabs would not balance.  One pass (default pipeline, ``latlon="fused"``, ``launch="pass"``) is

  ogg_supergrid_pass_dev  three launches on the caller's stream (through a plan handle, as Supergrid calls it, the first of them rides
                          in the previous pass's second): the quadrature tables; then ONE launch that carries the
                          lat-lon row strips (x, y, dx, dy, area, angle_dx from the axis formulas: 48 B written per cell,
                          nothing read), the mesh + angle workgroups of both caps and their quadrature strips side by side; then
                          the literal fix-up of the guarded bipolar cells (a displaced-pole quadrature in the reference's literal
                          arc form, ``dp_arc="literal"``, is a fourth launch: it needs more registers than launch B should have)

with no exchange between ranks: the caps are analytic in (i, j) and the lat-lon kernel needs only the axis formulas.
``launch="kernels"`` runs one launch per sub-grid and phase instead (the caps on side streams when ``overlap`` is set) and gives
the same bits.  ``latlon="stencil"`` is the pipeline of BASELINE.json's north_star taken literally: tile x, y (K1); send the first
x/y row of every band to the rank below (neighbour send/recv over RCCL, torch.distributed backend "nccl"); generic 2x3-stencil
kernel (K2) that reads x, y back.  Same bits again.  ``metrics_error()`` is the one collective of the default pipeline.

All six fields of every band stay in HBM (torch tensors); ``bands_to_host()`` copies them out and ``stitch()`` assembles the
sub-grids on the host exactly as the reference does (OGG:1315-1365).

torch is used for device memory, streams, graphs and torch.distributed only; every number is produced by libogg_hip.so.
"""
import ctypes
import math
import os

import numpy as np

from . import _lib as L
from . import ocean_grid_generator as ogg

FIELDS = ("x", "y", "dx", "dy", "area", "angle_dx")


def band(n_rows, rank, world):
    """Rows [lo, hi) of an n_rows-row sub-grid owned by `rank` (contiguous, balanced to within one row)."""
    return (n_rows * rank) // world, (n_rows * (rank + 1)) // world


def band_weighted(cost, rank, world, top_capacity=1.0, capacities=None):
    """Contiguous rows [lo, hi) of `rank` such that every rank gets about the same total cost; `cost` is the per-row
    cost array.  ``top_capacity``: the share of the LAST rank relative to the others' (it carries a fixed extra launch, the
    literal fix-up of the bipolar quadrature, and so takes a smaller share of EVERY sub-grid: the lat-lon strips and the cap
    workgroups of a rank finish together, so shortening one of them alone shortens nothing).  ``capacities``: one relative share
    per rank instead (refine_split: what every rank measured on its own GPU).  The boundaries are a function of (cost, world,
    top_capacity or capacities) only, identical on every rank."""
    c = np.concatenate(([0.0], np.cumsum(np.asarray(cost, dtype=np.float64))))
    total = c[-1]
    caps = [1.0] * (world - 1) + [float(top_capacity)] if capacities is None else [float(v) for v in capacities]
    if len(caps) != world or min(caps) <= 0.0:
        raise ValueError("band_weighted: %d capacities for %d ranks (all must be positive)" % (len(caps), world))
    before = np.concatenate(([0.0], np.cumsum(caps)))

    def edge(r):
        if r <= 0:
            return 0
        if r >= world:
            return len(cost)
        return int(np.searchsorted(c, total * before[r] / before[-1], side="left"))

    return edge(rank), edge(rank + 1)


class SubGridPlan(object):
    """Size and scalars of one sub-grid; `row0` is the first KEPT point row in the sub-grid's native numbering
    (e.g. the first row after the displaced-pole doughnut) and `nj1` the number of kept point rows."""

    def __init__(self, name, kind, nj1, row0=0, **scalars):
        self.name, self.kind, self.nj1, self.row0 = name, kind, int(nj1), int(row0)
        self.__dict__.update(scalars)


def all_reduce(t, op=None):
    """torch.distributed.all_reduce of a device tensor: over RCCL where the process group is one (backend "nccl"); through the host
    where it is gloo -- the CPU tests and the one-GPU rehearsal of bench.py, whose ranks share a card that RCCL will not share."""
    import torch.distributed as dist
    kw = {} if op is None else {"op": op}
    if t.is_cuda and dist.get_backend() == "gloo":
        c = t.cpu()
        dist.all_reduce(c, **kw)
        t.copy_(c)
    else:
        dist.all_reduce(t, **kw)
    return t


class SupergridPlan(object):
    """Host-side size logic of main() (OGG:969-1197, 1268-1313) for its whole flag surface: sub-grid selection (--grids), latitude
    overrides, --enhanced_equatorial (the spliced 1-D axis is built on the host, OGG:349-428, and handed to the lat-lon kernel as an
    explicit axis), --match_dy, --ensure_nj_even, the displaced pole, the doughnut and the south cuts (rows a cut removes are not
    generated when the cut follows from the sizes alone; a cut by angle of a displaced-pole cap is applied at stitch time).
    Needs the GPU unless ``mercator_axis`` is given: y* and the Mercator axis come from the device kernels, exactly as main() reads
    the joint latitudes off phiMerc."""

    def __init__(self, inverse_resolution, r_dp=0.0, lon_dp=80.0, lat_dp=-99.0, exfracdp=0.49, south_cutoff_row=0, south_cutoff_ang=-90.0,
                 skip_metrics=False, ensure_nj_even=False, no_south_cap=False, enhanced_equatorial=0, match_dy=(), grids="all",
                 shift_equator_to_u_point=True, bipolar_lower_lat=-99.0, mercator_lower_lat=-99.0, mercator_upper_lat=-99.0,
                 south_ocean_lower_lat=-99.0, Re=ogg._default_Re, mercator_axis=None, dp_arc=None, cap_symmetry=None):
        """``mercator_axis`` = (y0, phi_M) lets a caller that already holds the Mercator ordinate range and axis skip the
        two device calls (the CPU tests of the band / halo logic pass values computed elsewhere).  ``dp_arc``: arc form of
        the displaced-pole quadrature, "chord" (default, or OGG_DP_ARC: the form closer to the exact value of the reference's formula,
        DESIGN.md section 2) or "literal" (the reference's operation sequence; a fourth launch of the pass).  ``exfracdp=None``:
        main()'s own default 0.28*7/4 (OGG:891-892).  ``cap_symmetry``: None (the library's default: mirrored columns unless
        OGG_CAP_SYMMETRY=0), True / "mirror", False / "none" (every column of both caps evaluated, as the reference does)."""
        import contextlib
        import io

        refineS, refineR = 2, inverse_resolution
        self.Re = Re
        self.dp_arc = {"literal": L.DP_ARC_LITERAL, "chord": L.DP_ARC_CHORD}[dp_arc or ogg.default_dp_arc()]
        self.cap_symmetry = ogg._sym(cap_symmetry)   # OGG_SYM_* of include/ogg_hip.h
        self.skip_metrics = skip_metrics
        self.ensure_nj_even = ensure_nj_even
        self.south_cutoff_row, self.south_cutoff_ang = south_cutoff_row, south_cutoff_ang
        self.lon0, self.lenlon = -300.0, 360.0
        self.Ni = Ni = int(refineR * refineS * 360)
        doughnut = exfracdp if exfracdp is not None else 0.28 * 7 / 4

        def want(tok):   # OGG:1003,1035,1100,1140
            return (tok in grids) or ("all" in grids)

        # ---- Mercator (OGG:987-1030)
        phi_s, phi_n = -66.85954725, 64.05895973
        if mercator_upper_lat > -90:
            phi_n = mercator_upper_lat
        if mercator_lower_lat > -90:
            phi_s = mercator_lower_lat
        if refineR == 2:
            phi_s, phi_n = -68.0, 65.0
        if refineR == 1 and enhanced_equatorial:
            phi_s, phi_n = -77.8, 60.0
        merc = bp = so = sc = None
        phi_kept = None   # the Mercator latitudes main() sees as phiMerc[:, Ni//4]
        if want("mercator"):
            if mercator_axis is None:
                with contextlib.redirect_stdout(io.StringIO()):
                    phi_M, y_star = ogg.mercator_axis(Ni, phi_s, phi_n, refineR, shift_equator_to_u_point, ensure_nj_even, enhanced_equatorial,
                                                      return_y_star=True)
                y0 = int(y_star[0])
            else:
                y0, phi_M = int(mercator_axis[0]), np.asarray(mercator_axis[1], dtype=np.float64)
                if np.searchsorted(phi_M, 0.0) == 0:
                    raise Exception("   Ooops: Equator is not in the grid")
            skipM = 1 if (phi_M.size % 2 == 0 and ensure_nj_even) else 0                     # OGG:434-437
            merc = SubGridPlan("Merc", "mercator", phi_M.size - skipM, row0=skipM, y0=y0, n_axis=phi_M.size,
                               explicit_axis=(np.ascontiguousarray(phi_M) if enhanced_equatorial else None))
            phi_kept = phi_M[skipM:]
            dphi_so, dphi_no = phi_kept[1] - phi_kept[0], phi_kept[-1] - phi_kept[-2]        # OGG:1024-1025
            lat0_bp = float(phi_kept[-1])
        # ---- bipolar cap (OGG:1035-1062)
        if want("bipolar"):
            if bipolar_lower_lat > -90:
                lat0_bp = bipolar_lower_lat
            elif merc is None:
                raise Exception("the bipolar cap takes its lower latitude from the Mercator sub-grid (OGG:1030): add mercator to --grids "
                                "or give --bipolar_lower_lat")
            Nj_ncap = int(60 * refineR * refineS)
            if refineR == 2:
                Nj_ncap = 119 * refineS
            if refineR == 1 and enhanced_equatorial:
                Nj_ncap = 154
            if "bp" in match_dy:
                Nj_ncap = int(0.5 + (90.0 - lat0_bp) / dphi_no)
            if Nj_ncap % 2 != 0 and ensure_nj_even:
                Nj_ncap -= 1
            bp = SubGridPlan("BP", "bipolar", Nj_ncap + 1, Nj=Nj_ncap, lat0_bp=float(lat0_bp), lon_bp=self.lon0,
                             rp=float(np.tan(0.5 * (90 - lat0_bp) * ogg.PI_180)))
            # Row cost for the band split of the cap, in units of a plain cell row.  The quadrature guards its algebraic per-point form
            # near the two pole points (csrc/ogg_bipolar_dev.h, bp_point_fast): cell rows whose top edge lies above acos(2/sqrt(K))
            # carry the guard (1.3x a plain row).  The band that holds the last row also runs the tail launch (literal fix-up of the
            # guarded cells), a fixed cost of ~6.5 us whatever the resolution: that is taken off the last rank's share of EVERY sub-grid
            # (rows_of: tail_us / pass_us below), not off its cap rows alone -- a rank's lat-lon strips and cap workgroups end together
            # (timeline of launch B at 1/8 of the 1/8 degree grid: strips 35 us, quadrature 39 us), so a smaller cap share alone left the
            # last rank as slow as before.  (scripts/rank_sweep.py; OGG_BP_ROW_COST="fix,guard,lump" overrides the row weights: rows with
            # fix-up cells, guarded rows, a lump on the last row; OGG_TOP_RANK_TAIL_US the tail time.)
            import os
            K = float(os.environ.get("OGG_BP_GUARD_K", "4000"))
            if K > 4.0:
                lat_rows = lat0_bp + (np.arange(Nj_ncap + 1) + 1.0) * (90.0 - lat0_bp) / Nj_ncap   # top edge of each cell row
                guard_lat = math.degrees(math.acos(2.0 / math.sqrt(K)))
                fix_lat = math.degrees(math.acos(1.0 / math.sqrt(K)))
                # (With the cap's columns mirrored a guarded row costs 1.3 / 0.33 = 4 plain ones in arithmetic; weights of 3, 4 and 5 were
                # swept on the 8-way split of the 1/8 degree grid, one GPU: slowest rank 31.6 / 33.6 / 32.7 us against 31.7 with 1.3 and
                # the per-rank rebalancing step -- more ranks get guarded rows and each pays the fix-up launch.  1.3 stays.)
                w = [float(v) for v in os.environ.get("OGG_BP_ROW_COST", "1.3,1.3,0").split(",")]
                w_fix, w_guard, lump = w[0], w[1], (w[2] if len(w) > 2 else 0.0)
                bp.row_cost = np.where(lat_rows >= fix_lat, w_fix, np.where(lat_rows >= guard_lat, w_guard, 1.0))
                bp.row_cost[-1] += lump   # the tail launch (fix-up + j = ny row) only the top band runs, in plain-row equivalents
        # ---- Southern Ocean (OGG:1080-1103); like the reference this needs the Mercator sub-grid
        lat0_SO = -78.0
        if south_ocean_lower_lat > -90:
            lat0_SO = south_ocean_lower_lat
        if phi_kept is None:
            raise Exception("the Southern Ocean and southern cap sections read the Mercator sub-grid (OGG:1083): add mercator to --grids")
        latUp_SO = float(phi_kept[0])
        lenlat_SO = latUp_SO - lat0_SO
        Nj_SO = int(refineR * 55)
        if refineR == 2 and enhanced_equatorial:
            Nj_SO = 109
        if refineR == 1 and enhanced_equatorial:
            Nj_SO = 0
        if "so" in match_dy:
            Nj_SO = int(0.5 + lenlat_SO / dphi_so)
        skipS = 0
        if Nj_SO != 0 and want("so"):
            skipS = 1 if ((Nj_SO + 1) % 2 == 0 and ensure_nj_even) else 0
            so = SubGridPlan("SO", "latlon", Nj_SO + 1 - skipS, row0=skipS, lnj=Nj_SO, lat0=lat0_SO, lenlat=lenlat_SO)
        # ---- southern cap (OGG:1122-1197)
        if so is None:
            raise Exception("the southern cap section reads the Southern Ocean sub-grid (OGG:1122): it cannot be left out")
        lat0_SC = lat0_SO + (skipS * lenlat_SO) / float(Nj_SO) if skipS else lat0_SO        # phiSO[0, Ni//4]
        if "p125sc" in match_dy:
            lat0_SC = lat0_SO
        Nj_scap = int(refineR * 40) * 7 // 4
        if no_south_cap or (enhanced_equatorial and refineR in (1, 2)):
            Nj_scap = 0
        if Nj_scap != 0 and want("sc"):
            if r_dp == 0.0 and lat_dp <= -90.0:
                Nj_scap = int((lat0_SC + 90.0) / (1.0 / refineR / refineS))
                skipC = 1 if ((Nj_scap + 1) % 2 == 0 and ensure_nj_even) else 0
                sc = SubGridPlan("SC", "latlon", Nj_scap + 1 - skipC, row0=skipC, lnj=Nj_scap, lat0=-90.0, lenlat=90 + lat0_SO)
            else:
                if lat_dp > -90:
                    r_dp = float(np.tan((90 + lat_dp) * ogg.PI_180) / np.tan((90 + lat0_SC) * ogg.PI_180))
                jmint = 0
                if doughnut != 0.0:
                    jmin = np.ceil(doughnut * Nj_scap)
                    jmint = int(jmin + np.mod(jmin, 2))
                if (Nj_scap + 1 - jmint) % 2 == 0 and ensure_nj_even:
                    jmint += 1
                sc = SubGridPlan("SC", "dpole", Nj_scap + 1 - jmint, row0=jmint, Nj=Nj_scap, lat0=lat0_SC, lon_dp=lon_dp, r_dp=r_dp)
        self.Nj_scap = Nj_scap
        self.lat0_SO, self.latUp_SO = lat0_SO, latUp_SO
        self.subs = [s for s in (sc, so, merc, bp) if s is not None]  # south -> north
        if bp is not None and not skip_metrics:
            # the last rank's fix-up launch against a single-GPU pass of this grid (1.08e11 cells/s measured at 1/8 degree): see rows_of
            import os
            # measured (events, top rank of 8): 6.5 us at 1/8 degree, 17 us at 1/16 degree -- a launch plus the literal re-evaluation of the
            # guarded cells, whose number grows with the square of the resolution
            tail_us = float(os.environ.get("OGG_TOP_RANK_TAIL_US", "%.3f" % (2.9 + 0.0563 * refineR * refineR)))
            cells = sum(s.nj1 - 1 for s in self.subs) * Ni
            pass_us = cells / 1.08e5
            if sc is not None and sc.kind == "dpole":   # + the displaced-pole quadrature: 0.66 ns per cell literal (launch D), 0.10 ns chord
                pass_us += (sc.nj1 - 1) * Ni * (6.6e-4 if self.dp_arc == L.DP_ARC_LITERAL else 1.0e-4)
            self.set_split_times(tail_us, pass_us, "fitted constants (builder boxes, rounds 2-3): UNVALIDATED on a multi-GPU node; "
                                 "calibrate_split() replaces them by this box's own timings")
        # Rows that --south_cutoff_row / _ang remove from the stitched grid (OGG:1268-1313) are not generated when the cut is known from the
        # sizes alone -- by row always; by angle on a regular cap, whose latitudes are an axis formula -- just as the doughnut rows are not
        # (OM4: 82 of the displaced-pole cap's 143 kept rows).  The cut of a displaced-pole cap by ANGLE needs the cap's latitudes and is
        # applied after the pass (south_cut(sc_y0) / Supergrid.south_cut()).
        self.cut_applied = None
        by_angle_later = south_cutoff_row <= 0 and south_cutoff_ang > -90 and sc is not None and sc.kind != "latlon"
        if (south_cutoff_row > 0 or south_cutoff_ang > -90) and not by_angle_later:
            c_sc, c_so, gone = self.south_cut()
            if gone:
                self.subs = [s for s in self.subs if s.name != "SC"]
                so.row0, so.nj1 = so.row0 + min(c_so, so.nj1), max(so.nj1 - c_so, 0)
                if so.nj1 == 0:      # a cut beyond the Southern Ocean piece: nothing of it is stitched (its [jcut:] slice is empty)
                    self.subs = [s for s in self.subs if s.name != "SO"]
            elif c_sc > 0:
                sc.row0, sc.nj1 = sc.row0 + c_sc, sc.nj1 - c_sc
            self.cut_applied = (c_sc, c_so, gone)

    split_times = None

    def set_split_times(self, tail_us, pass_us, source, top_capacity=None, rank_capacity=None):
        """What the band split takes the last rank's share from (rows_of).  The MODEL: `tail_us`, what the bipolar quadrature's fix-up
        launch -- which only the rank with the top rows runs -- adds to that rank's pass, and `pass_us`, one pass of the whole grid on
        one GPU: share of the last rank = 1 - world * tail_us / pass_us of the others'.  A MEASURED share (`top_capacity`, from
        calibrate_split) takes precedence for the world size it was measured at.  The same values on every rank give the same edges."""
        for s in self.subs:
            s.tail_us, s.pass_us, s.top_capacity, s.rank_capacity = float(tail_us), float(pass_us), top_capacity, rank_capacity
        self.split_times = {"tail_us": float(tail_us), "pass_us": float(pass_us), "source": source}
        if top_capacity is not None:
            self.split_times["top_capacity"] = {"world": int(top_capacity[0]), "share_of_last_rank": float(top_capacity[1])}
        if rank_capacity is not None:   # (refine_split: one relative share per rank; takes precedence for its world size)
            self.split_times["rank_capacity"] = {"world": int(rank_capacity[0]), "shares": [float(v) for v in rank_capacity[1]]}

    def calibrate_split(self, device, rank=0, world=1, passes=40, broadcast=True, force=False, rounds=5):
        """Replace the fitted split constants by a measurement on THIS box, before any band buffer exists (rank 0 alone works, about a
        thousand passes; the others wait in the broadcast).  Rank 0 (i) runs the whole grid as one rank -- `passes` timed passes after
        100 warm ones: pass_us; eight more with the library's launch events: the fix-up launch less the cost of an event record, a first
        guess of tail_us and hence of the last rank's share c = 1 - world * tail_us / pass_us --, then (ii) up to `rounds` times (until the two agree to 1 %): times the share
        of a middle rank and of the last rank under the current split (on its one GPU, one after the other) and scales
        c <- c * T_middle / T_last, which equalises the two whatever makes the last rank slower (a fixed extra launch, or rows that cost
        more than the row weights say on this box); the share of the round whose slower rank was fastest is kept; and (iii) broadcasts (tail_us, pass_us, c).  Every rank then derives the same edges.
        Needs a GPU on rank 0 and, for world > 1 with ``broadcast``, an initialised process group.  Returns ``split_times`` (None when
        the split has no such term: one rank, no bipolar cap, --skip_metrics).  OGG_TOP_RANK_TAIL_US in the environment pins the
        model's tail time and skips (ii).  ``force``: measure and broadcast at world size 1 too (bench.py under OGG_FORCE_DIST=1: the
        RCCL broadcast on a one-GPU box)."""
        import os
        import time
        if (world <= 1 and not force) or self.split_times is None:
            return self.split_times
        vals = (0.0, 0.0, 0.0)
        if rank == 0:
            import torch

            def timed_us(g, warm, n):
                for _ in range(warm):
                    g.run_pass()
                torch.cuda.synchronize(g.device)
                t0 = time.perf_counter()
                for _ in range(n):
                    g.run_pass()
                torch.cuda.synchronize(g.device)
                return (time.perf_counter() - t0) / n * 1e6

            g = Supergrid(self, rank=0, world=1, device=device, halo="recompute")
            g.launch, g.overlap = "pass", False
            pass_us = timed_us(g, 100, passes)          # (the clocks of a fresh process take ~30 ms to ramp)
            g.reserve_pass_events(8)
            g.pass_events = []
            for _ in range(8):
                g.run_pass()
            tail_us = g.pass_launch_times_ms()["pass_tail"]["ms"] * 1e3
            g.pass_events = None
            # an interval between two event records holds the records' own cost besides the launch: two records back to back measure it
            st = g._stream()
            ev = g._new_events()
            gap = []
            for _ in range(8):
                L.call("ogg_event_record", ev[0], st)
                L.call("ogg_event_record", ev[1], st)
                L.call("ogg_stream_synchronize", st)
                ms = ctypes.c_float()
                L.call("ogg_event_elapsed_ms", ev[0], ev[1], ctypes.byref(ms))
                gap.append(ms.value * 1e3)
            for k in range(5):
                L.call("ogg_event_destroy", ev[k])
            g.close()
            del g
            tail_us = max(tail_us - min(gap), 1.0)
            cap = 0.0                                    # 0: no measured share (the model stands)
            if os.environ.get("OGG_TOP_RANK_TAIL_US"):
                tail_us = float(os.environ["OGG_TOP_RANK_TAIL_US"])
            elif world > 1:
                cap = 1.0 - world * tail_us / pass_us if world * tail_us / pass_us <= 0.5 else 1.0
                tried = []                               # (slower of the two shares, share of the last rank) of every round
                for _ in range(rounds):
                    self.set_split_times(tail_us, pass_us, "calibrating", top_capacity=(world, cap))
                    t = []
                    for r in ((world - 1) // 2, world - 1):
                        h = Supergrid(self, rank=r, world=world, device=device, halo="recompute")
                        h.launch, h.overlap = "pass", False
                        t.append(timed_us(h, 60, 150))
                        h.close()
                        del h
                    tried.append((max(t), cap))
                    if abs(t[0] / t[1] - 1.0) < 0.01:      # the two shares within 1 %: balanced as far as a timing can tell
                        break
                    cap = min(max(cap * t[0] / t[1], 0.3), 1.3)
                # the split whose slower share was fastest: a share's time is not always monotone in its size (tiling thresholds of the
                # pass, the write path of some boxes), and then the last step of the iteration need not be its best
                cap = min(tried)[1]
            torch.cuda.empty_cache()
            vals = (tail_us, pass_us, cap)
        if broadcast:   # three doubles from rank 0: a device tensor over RCCL, through the host where the group is gloo (as all_reduce does)
            import torch
            import torch.distributed as dist
            on_gpu = dist.get_backend() != "gloo"
            t = torch.tensor(list(vals), dtype=torch.float64, device=(device if on_gpu else "cpu"))
            dist.broadcast(t, src=0)
            vals = tuple(float(v) for v in t.tolist())
        tail_us, pass_us, cap = vals
        self.set_split_times(tail_us, pass_us, "measured by rank 0 at plan build (whole grid: %d passes + launch events of 8 more; then the shares "
                             "of a middle and of the last rank under the split, until they agree to 1 %% or %d times)" % (passes, rounds),
                             top_capacity=((world, cap) if cap > 0.0 else None))
        return self.split_times

    def refine_split(self, per_rank_us, world, deadband=0.03):
        """One rebalancing step from what every rank measured on its OWN GPU under the current split, all ranks running at once (the
        conditions of the job: calibrate_split times the shares one after the other on rank 0's idle chip).  ``per_rank_us``: every
        rank's time per pass in rank order -- the same list on every rank (bench.py all-gathers it), so every rank derives the same new
        edges and nothing is broadcast.  Every rank's share is scaled by mean(T) / T_rank (a rank whose rows cost more than the cost
        model says -- the rows next to the guarded ones, a GPU of the slower write-path class -- gets fewer); inside the ``deadband``
        (every rank within 3 % of the mean: a timing cannot tell) the split stays.  Returns True when the split changed (band buffers
        must be rebuilt).  The measured times and the shares before and after are recorded in ``split_times["self_calibration"]``."""
        if world <= 1 or self.split_times is None or len(per_rank_us) != world:
            return False
        t = [float(v) for v in per_rank_us]
        if min(t) <= 0.0:
            return False
        caps_old = self.rank_capacities(world)
        mean = sum(t) / world
        worst = max(abs(v / mean - 1.0) for v in t)
        rec = {"world": world, "per_rank_us": [round(v, 3) for v in t], "shares_before": [round(c, 5) for c in caps_old],
               "share_of_last_rank_before": caps_old[-1] / caps_old[0], "others_over_last": (sum(t[:-1]) / (world - 1)) / t[-1],
               "largest_deviation_from_mean": worst}
        changed = worst > deadband
        caps_new = caps_old
        if changed:
            caps_new = [min(max(c * mean / v, 0.3 * c), 1.5 * c) for c, v in zip(caps_old, t)]
            norm = caps_new[0]
            caps_new = [c / norm for c in caps_new]     # (relative shares: rank 0 = 1, as before)
        rec["shares_after"] = [round(c, 5) for c in caps_new]
        rec["share_of_last_rank_after"] = caps_new[-1] / caps_new[0]
        hist = list(self.split_times.get("self_calibration", []))
        self.set_split_times(self.split_times["tail_us"], self.split_times["pass_us"], self.split_times["source"],
                             top_capacity=(world, caps_new[-1] / caps_new[0]), rank_capacity=((world, caps_new) if changed else self._rank_capacity()))
        self.split_times["self_calibration"] = hist + [rec]
        return changed

    def _rank_capacity(self):
        return getattr(self.subs[0], "rank_capacity", None)

    def rank_capacities(self, world):
        """Relative shares of the ranks under the current split (rank 0 = 1 unless refine_split has set them)."""
        s0 = self.subs[0]
        rc = getattr(s0, "rank_capacity", None)
        if rc is not None and int(rc[0]) == world:
            return [float(v) for v in rc[1]]
        old = getattr(s0, "top_capacity", None)
        if old is not None and int(old[0]) == world:
            cap = float(old[1])
        else:
            tail, whole = self.split_times["tail_us"], self.split_times["pass_us"]
            cap = min(1.0, 1.0 - world * tail / whole) if (whole > 0 and world * tail / whole <= 0.5) else 1.0
        return [1.0] * (world - 1) + [cap]

    def south_cut(self, sc_y0=None):
        """(rows cut from the southern cap, rows cut from the Southern Ocean piece, cap removed) by --south_cutoff_row / _ang,
        OGG:1268-1313, including the parity bumps of --ensure_nj_even.  ``sc_y0``: column 0 of the cap's latitudes, needed for
        --south_cutoff_ang only.  (0, 0, False) once the plan has applied the cut to its own sub-grid sizes (``cut_applied``)."""
        if getattr(self, "cut_applied", None) is not None:
            return 0, 0, False
        first = self.subs[0]
        has_sc = first.name == "SC"
        n_sc = first.nj1 if has_sc else 0
        if self.south_cutoff_row > 0:
            jcut = self.south_cutoff_row - 1
        elif self.south_cutoff_ang > -90:
            if not has_sc:
                raise Exception("--south_cutoff_ang reads the southern cap's latitudes (OGG:1278): there is no southern cap")
            if sc_y0 is None:
                if first.kind != "latlon":
                    raise ValueError("south_cut: the cap's latitudes are needed for --south_cutoff_ang")
                sc_y0 = first.lat0 + (np.arange(first.row0, first.row0 + first.nj1) * first.lenlat) / float(first.lnj)
            jcut = 1 + int(np.nonzero(np.asarray(sc_y0) < self.south_cutoff_ang)[0][-1])
        else:
            return 0, 0, False
        if has_sc and jcut < n_sc:
            if (n_sc - jcut) % 2 == 0 and self.ensure_nj_even:
                jcut += 1
            return jcut, 0, False
        so = next((s for s in self.subs if s.name == "SO"), None)
        if so is None:
            return 0, 0, False
        if not has_sc:
            raise Exception("--south_cutoff_row without a southern cap: the reference reads lamSC here (OGG:1299)")
        jcut_so = jcut - n_sc
        if (so.nj1 - 1 - jcut_so - 1) % 2 == 0 and self.ensure_nj_even:
            jcut_so += 1
        return n_sc, jcut_so, True

    def rows_cut(self, sc_y0=None):
        """Rows removed at the south end of the stitched grid (stitching drops the cap's last row, so a cap that is cut away
        altogether takes nj1 - 1 stitched rows with it)."""
        c_sc, c_so, gone = self.south_cut(sc_y0)
        if not gone:
            return c_sc
        so = next(s for s in self.subs if s.name == "SO")
        return self.subs[0].nj1 - 1 + min(c_so, so.nj1 - 1)   # a cut beyond the Southern Ocean piece leaves none of its rows (OGG:1306-1311)

    @property
    def nyp(self):
        return sum(s.nj1 - 1 for s in self.subs) + 1 - self.rows_cut()

    @property
    def cells(self):
        """cells of the final stitched grid (the unit of BASELINE.json's metric): (nyp-1)*Ni"""
        return (self.nyp - 1) * self.Ni


class Supergrid(object):
    """Band-sharded device pass.  `halo` is "rccl" (neighbour send/recv through torch.distributed; needs an initialised
    process group when world > 1), "local" (world virtual ranks inside ONE process: the neighbour's row is copied
    device-to-device; used to test the band logic on a single GPU) or "recompute" (the halo row of a lat-lon sub-grid is
    re-tiled from the 1-D axis every rank already holds; bitwise identical)."""

    def __init__(self, plan, rank=0, world=1, device="cuda:0", halo="rccl", peers=None, latlon="fused"):
        """``latlon``: how the sub-grids that are lat-lon by construction (Mercator, Southern Ocean, regular southern
        cap) are produced.  "fused" (default): one kernel writes all six fields from the two 1-D axes -- 48 B written per
        cell, nothing read, and no halo (every rank holds the whole 1-D axis).  "stencil": tile x, y; exchange the halo
        row; run the generic 2x3-stencil kernel that reads x, y back -- the kernel the drop-in functions use for arbitrary
        meshes.  Both give the same bits."""
        import torch

        self.torch = torch
        self.plan, self.rank, self.world, self.halo, self.latlon = plan, rank, world, halo, latlon
        self.device = torch.device(device)
        self.peers = peers  # halo="local": list of all virtual ranks' Supergrid objects
        self.overlap = True
        # "pass": lat-lon sub-grids + bipolar cap through ogg_tripolar_pass_dev (three launches on one stream, the two kinds
        # of work sharing each launch); "kernels": one call per sub-grid and phase, on side streams when `overlap` is set
        self.launch = "pass" if latlon == "fused" else "kernels"
        self.pass_events = None  # a list: tripolar_pass() times its launches into it
        self._event_pool = []
        self._pass_args = None
        self._rows_ws = None
        self._side = None
        self.buf = {}
        self.timings = {}
        ni1 = plan.Ni + 1
        # device "cpu" allocates the same band buffers in host memory: only the partition and the halo exchange can
        # run there (tests over gloo); the phases need a GPU
        import contextlib
        with (torch.cuda.device(self.device) if self.device.type == "cuda" else contextlib.nullcontext()):
            self.lon1d = torch.empty(ni1, dtype=torch.float64, device=self.device)
            # Where the output arrays lie moves a write-bound pass by +-6 %: allocated one by one, the fields of a band land wherever the
            # allocator puts them, and the 1/8 degree pass runs 0.185 or 0.21 ms by the draw (four grids in one process: 0.185 / 0.207 /
            # 0.187 / 0.210; 1/16 degree 0.91 / 1.07).  OGG_FIELD_SLAB=<bytes> (experiment) takes the fields of all bands from ONE
            # allocation, every field at a multiple of that many bytes: 2 GiB apart all four grids ran 0.184-0.186 on one box and one of
            # two on another; 256 B ... 1 GiB apart no better than the draw (scripts/split_pass_probe.py --copies 4,
            # profiles/r05_split_pass_probe.md).  Not understood, so not a default: one allocation per field.
            def field_shapes():
                for s in plan.subs:
                    lo, hi = self.rows_of(s, rank, world)
                    n = hi - lo
                    n_cell = max(min(hi, s.nj1 - 1) - lo, 0)
                    halo = 1 if (latlon == "stencil" and s.kind in ("mercator", "latlon") and n_cell > 0 and hi < s.nj1) else 0
                    for r, c in ((n + halo, ni1), (n + halo, ni1), (n, ni1 - 1), (n_cell, ni1), (n_cell, ni1 - 1), (n, ni1)):
                        yield max(r, 0) * c * 8

            slab, slab_off, slab_align = None, 0, 0
            if self.device.type == "cuda":
                forced = os.environ.get("OGG_FIELD_SLAB")
                sizes = list(field_shapes())
                if forced is not None:
                    slab_align = int(forced)
                if slab_align > 0:
                    total = sum((b + slab_align - 1) // slab_align * slab_align for b in sizes) + slab_align
                    try:
                        free_bytes = torch.cuda.mem_get_info(self.device)[0]
                    except Exception:  # noqa: BLE001 -- no memory information: no slab
                        free_bytes = 0
                    if total > free_bytes * 9 // 10:   # (several grids in one process: the later ones fall back)
                        slab_align = 0
                    else:
                        slab = torch.empty(total, dtype=torch.uint8, device=self.device)
                        slab_off = (-slab.data_ptr()) % slab_align
                        self._slab = slab
            self.field_slab = {"spacing_bytes": slab_align, "bytes": int(slab.numel()) if slab is not None else 0}

            def field_buffer(r, c):
                nonlocal slab_off
                if slab is None:
                    return torch.empty((max(r, 0), c), dtype=torch.float64, device=self.device)
                nbytes = max(r, 0) * c * 8
                t = slab[slab_off:slab_off + nbytes].view(torch.float64).view(max(r, 0), c)
                slab_off += (nbytes + slab_align - 1) // slab_align * slab_align
                return t

            for s in plan.subs:
                lo, hi = self.rows_of(s, rank, world)
                n = hi - lo
                n_cell = min(hi, s.nj1 - 1) - lo  # cell rows owned (a cell row j belongs to the owner of point row j)
                needs_halo = latlon == "stencil" and s.kind in ("mercator", "latlon") and n_cell > 0 and hi < s.nj1
                b = {"lo": lo, "hi": hi, "n": n, "n_cell": max(n_cell, 0), "needs_halo": needs_halo}
                rows_xy = n + (1 if needs_halo else 0)
                for f, (r, c) in (("x", (rows_xy, ni1)), ("y", (rows_xy, ni1)), ("dx", (n, ni1 - 1)), ("dy", (b["n_cell"], ni1)),
                                  ("area", (b["n_cell"], ni1 - 1)), ("angle_dx", (n, ni1))):
                    b[f] = field_buffer(r, c)
                if s.kind == "mercator":
                    if getattr(s, "explicit_axis", None) is not None:   # enhanced-equator axis: spliced on the host (OGG:349-428)
                        b["axis"] = torch.from_numpy(s.explicit_axis).to(self.device)
                    else:
                        b["axis"] = torch.empty(s.n_axis, dtype=torch.float64, device=self.device)
                elif s.kind == "latlon":
                    b["axis"] = torch.empty(s.lnj + 1, dtype=torch.float64, device=self.device)
                elif s.kind == "bipolar":
                    b["ws_bytes"] = int(L.load().ogg_bipolar_quad_workspace_bytes(5, plan.Ni, s.Nj))
                    b["ws"] = torch.empty(b["ws_bytes"], dtype=torch.uint8, device=self.device)
                elif s.kind == "dpole":   # one workspace for the pass (mesh words + quadrature tables and words); the stand-alone
                    b["ws_bytes"] = int(L.load().ogg_dpole_band_workspace_bytes(4, plan.Ni, n))   # kernels use its two parts
                    b["ws"] = torch.zeros(max(b["ws_bytes"], 16), dtype=torch.uint8, device=self.device)
                    b["ws_mesh_bytes"] = (int(L.load().ogg_displaced_pole_grid_workspace_bytes(plan.Ni, n)) + 255) // 256 * 256
                self.buf[s.name] = b

    @staticmethod
    def rows_of(s, rank, world):
        """Point rows [lo, hi) of sub-grid `s` owned by `rank`: equal row counts, or equal cost where rows differ in cost; the
        last rank's share of every sub-grid is smaller by what the fix-up launch costs it (SubGridPlan.tail_us / pass_us, set by
        SupergridPlan when the grid has a bipolar cap with metrics)."""
        cost = getattr(s, "row_cost", None)
        tail_us, pass_us = getattr(s, "tail_us", 0.0), getattr(s, "pass_us", 0.0)
        measured = getattr(s, "top_capacity", None)
        per_rank = getattr(s, "rank_capacity", None)
        if world > 1 and per_rank is not None and int(per_rank[0]) == world:   # every rank's share as refine_split rebalanced them
            return band_weighted(np.ones(s.nj1) if cost is None else cost, rank, world, capacities=per_rank[1])
        cap = 1.0
        if world > 1 and measured is not None and int(measured[0]) == world:
            cap = float(measured[1])        # the last rank's share as calibrate_split measured it for this world size
        # (a grid so small, or ranks so many, that the fix-up launch is more than half of a rank's share: the linear correction no longer
        # describes anything -- equal shares then, rather than a last rank clamped to a sliver or to no rows at all)
        elif world > 1 and tail_us > 0.0 and pass_us > 0.0 and world * tail_us / pass_us <= 0.5:
            cap = min(1.0, 1.0 - world * tail_us / pass_us)
        if cost is None and cap == 1.0:
            return band(s.nj1, rank, world)
        return band_weighted(np.ones(s.nj1) if cost is None else cost, rank, world, cap)

    # -- helpers ---------------------------------------------------------------------------------------------
    def _stream(self):
        if self.device.type != "cuda":
            raise RuntimeError("the supergrid phases run on a GPU only (there is no CPU compute path)")
        return self.torch.cuda.current_stream(self.device).cuda_stream

    @staticmethod
    def _p(t, row=0):
        return t.data_ptr() + row * t.stride(0) * 8 if t.dim() == 2 else t.data_ptr() + row * 8

    def _timed(self, key, fn):
        """Run fn() bracketed by events on the launch stream when per-kernel timing is on."""
        if self._events is None:
            fn()
            return
        e0 = self.torch.cuda.Event(enable_timing=True)
        e1 = self.torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        self._events.setdefault(key, []).append((e0, e1))

    _events = None

    # -- phases ----------------------------------------------------------------------------------------------
    @staticmethod
    def _selected(s, only, kinds):
        return (only is None or s.name == only) and (kinds is None or s.kind in kinds)

    def _latlon_bands(self, only=None, kinds=None):
        """ogg_latlon_band descriptors of this rank's lat-lon sub-grid bands."""
        bands = []
        for s in self.plan.subs:
            b = self.buf[s.name]
            if b["n"] == 0 or not self._selected(s, only, kinds) or s.kind not in ("mercator", "latlon"):
                continue
            band = L.LatlonBand()
            if s.kind == "mercator" and getattr(s, "explicit_axis", None) is not None:
                band.axis_kind, band.lat1d = 2, b["axis"].data_ptr()
            elif s.kind == "mercator":
                band.axis_kind, band.y0 = 1, s.y0
            else:
                band.axis_kind, band.a0, band.len, band.denom = 0, s.lat0, s.lenlat, float(s.lnj)
            band.k0, band.n_pt_rows, band.n_cell_rows = s.row0 + b["lo"], b["n"], b["n_cell"]
            band.x, band.y, band.angle = b["x"].data_ptr(), b["y"].data_ptr(), b["angle_dx"].data_ptr()
            band.dx = b["dx"].data_ptr()
            band.dy = b["dy"].data_ptr() if b["n_cell"] else None
            band.area = b["area"].data_ptr() if b["n_cell"] else None
            bands.append(band)
        return bands

    def tripolar_pass(self):
        """Lat-lon sub-grids and both caps of this rank through ogg_supergrid_pass_dev."""
        p, st = self.plan, self._stream()
        if self._pass_args is None:  # the descriptors only hold scalars and buffer addresses, which never change: build once
            bands = self._latlon_bands()
            arr = (L.LatlonBand * max(len(bands), 1))(*bands)
            cap = scap = None
            for s in p.subs:
                b = self.buf[s.name]
                if s.kind == "bipolar" and b["n"] > 0:
                    cap = L.BipolarBand()
                    cap.Ni, cap.Nj, cap.lat0_bp, cap.lon_bp, cap.rp, cap.Re, cap.order = p.Ni, s.Nj, s.lat0_bp, s.lon_bp, s.rp, p.Re, 5
                    cap.symmetry = p.cap_symmetry
                    cap.j0, cap.n_pt_rows, cap.n_cell_rows = b["lo"], b["n"], b["n_cell"]
                    cap.x, cap.y, cap.angle = b["x"].data_ptr(), b["y"].data_ptr(), b["angle_dx"].data_ptr()
                    cap.dx = b["dx"].data_ptr()
                    cap.dy = b["dy"].data_ptr() if b["n_cell"] else None
                    cap.area = b["area"].data_ptr() if b["n_cell"] else None
                    cap.workspace, cap.workspace_bytes = b["ws"].data_ptr(), b["ws_bytes"]
                elif s.kind == "dpole" and b["n"] > 0:
                    scap = L.DpoleBand()
                    scap.Ni, scap.Nj, scap.lon0, scap.lat0, scap.lon_dp, scap.r_dp, scap.Re = p.Ni, s.Nj, p.lon0, s.lat0, s.lon_dp, s.r_dp, p.Re
                    scap.order, scap.arc_form, scap.symmetry = 4, p.dp_arc, p.cap_symmetry
                    scap.j0, scap.n_pt_rows, scap.n_cell_rows = s.row0 + b["lo"], b["n"], b["n_cell"]
                    scap.x, scap.y, scap.angle = b["x"].data_ptr(), b["y"].data_ptr(), b["angle_dx"].data_ptr()
                    scap.dx = b["dx"].data_ptr()
                    scap.dy = b["dy"].data_ptr() if b["n_cell"] else None
                    scap.area = b["area"].data_ptr() if b["n_cell"] else None
                    scap.workspace, scap.workspace_bytes = b["ws"].data_ptr(), b["ws_bytes"]
            # ... and so is the plan of the launches (kernel parameters, grid sizes, tiling knobs read from the environment NOW):
            # a pass then costs the host one ctypes call and the launches
            capref = ctypes.byref(cap) if cap is not None else None
            scapref = ctypes.byref(scap) if scap is not None else None
            handle = ctypes.c_void_p()
            L.call("ogg_supergrid_pass_plan_dev", len(bands), arr, p.Ni + 1, p.lon0, p.lenlon, p.Re, 0 if p.skip_metrics else 1, capref, scapref,
                   ctypes.byref(handle))
            self._pass_args = (bands, arr, cap, scap, handle)
            self._pass_run = L.load().ogg_supergrid_pass_run_dev
        handle = self._pass_args[4]
        if self.pass_events is None:
            rc = self._pass_run(handle, None, None, st)
            if rc:
                L.check(rc)
            return
        # per-launch timing: five HIP events recorded by the library around its (up to) four launches
        evs = self._event_pool.pop() if self._event_pool else self._new_events()
        self.pass_bytes = (ctypes.c_double * 4)()
        L.call("ogg_supergrid_pass_run_dev", handle, evs, self.pass_bytes, st)
        self.pass_events.append(evs)

    def pass_plan_info(self):
        """(workspace slots of the fused pass's plan, passes so far that started with launch B because the previous pass's launch B had
        built their tables) -- (0, 0) before the first pass."""
        if self._pass_args is None:
            return 0, 0
        lib = L.load()
        return int(lib.ogg_supergrid_pass_plan_slots(self._pass_args[4])), int(lib.ogg_supergrid_pass_plan_carried_runs(self._pass_args[4]))

    def replan(self):
        """Drop the cached plan of the pass (rebuilt by the next pass: after a change of an OGG_* tiling knob in the environment)."""
        if self._pass_args is not None:
            L.call("ogg_supergrid_pass_plan_destroy", self._pass_args[4])
            self._pass_args = None

    def close(self):
        """Release the plan of the pass (its workspace slots are device allocations of the library) at a point of the caller's choosing
        rather than at garbage-collection time."""
        self.replan()

    def __del__(self):
        try:
            self.replan()
        except Exception:   # noqa: BLE001 -- interpreter shutdown
            pass

    @staticmethod
    def _new_events():
        evs = (ctypes.c_void_p * 5)()
        for k in range(5):
            e = ctypes.c_void_p()
            L.call("ogg_event_create", ctypes.byref(e))
            evs[k] = e
        return evs

    def reserve_pass_events(self, n):
        """Create the events for n timed passes ahead of a timed region."""
        self._event_pool = [self._new_events() for _ in range(n)]

    def pass_launch_times_ms(self):
        """Mean duration and algorithmic bytes of the four launches of the fused pass (the fourth exists only with a displaced-pole
        cap in the literal arc form) over the passes that ran while `pass_events` was a list; empties the list."""
        L.call("ogg_stream_synchronize", self._stream())
        tot = [0.0, 0.0, 0.0, 0.0]
        n = len(self.pass_events)
        for evs in self.pass_events:
            for k in range(4):
                ms = ctypes.c_float()
                L.call("ogg_event_elapsed_ms", evs[k], evs[k + 1], ctypes.byref(ms))
                tot[k] += ms.value
            for k in range(5):
                L.call("ogg_event_destroy", evs[k])
        self.pass_events = []
        by = list(self.pass_bytes) if n else [0.0, 0.0, 0.0, 0.0]
        out = {name: {"ms": tot[k] / max(n, 1), "alg_bytes": by[k]} for k, name in enumerate(("pass_a", "pass_b", "pass_tail", "pass_dpquad"))}
        out["sampled_passes"] = n
        return out

    def phase_a(self, only=None, kinds=None):
        """Coordinates of this rank's bands (optionally of one sub-grid / of some kinds of sub-grid only)."""
        p, st = self.plan, self._stream()
        ni1 = p.Ni + 1
        if only is None and self.latlon != "fused":
            L.call("ogg_linear_axis_dev", ni1, p.lon0, p.lenlon, float(p.Ni), self.lon1d.data_ptr(), st)
        if self.latlon == "fused":
            bands = self._latlon_bands(only, kinds)
            if bands:
                import os
                arr = (L.LatlonBand * len(bands))(*bands)
                if os.environ.get("OGG_LATLON_ROWS", "0") != "0":
                    # a launch with nothing but lat-lon sub-grids: (field, row)-ordered workgroups fed from row / column tables (an
                    # option: same bits, and within 4 % of the column-tile kernel either way on the same box)
                    need = int(L.load().ogg_latlon_rows_workspace_bytes(len(bands), arr, ni1))
                    if self._rows_ws is None or self._rows_ws.numel() < need:
                        self._rows_ws = self.torch.empty(need, dtype=self.torch.uint8, device=self.device)
                    ws = self._rows_ws
                    self._timed("latlon_fused", lambda: L.call("ogg_latlon_supergrid_rows_ws_dev", len(bands), arr, ni1, p.lon0, p.lenlon,
                                                               p.Re, 0 if p.skip_metrics else 1, ws.data_ptr(), ws.numel(), st))
                else:
                    self._timed("latlon_fused", lambda: L.call("ogg_latlon_supergrid_multi_dev", len(bands), arr, ni1, p.lon0, p.lenlon,
                                                               p.Re, 0 if p.skip_metrics else 1, st))
        for s in p.subs:
            b = self.buf[s.name]
            if b["n"] == 0 or not self._selected(s, only, kinds):
                continue
            if s.kind in ("mercator", "latlon"):
                if self.latlon == "fused":
                    continue
                if s.kind == "mercator":
                    if getattr(s, "explicit_axis", None) is None:
                        L.call("ogg_mercator_axis_dev", p.Ni, s.y0, s.n_axis, b["axis"].data_ptr(), st)
                else:
                    L.call("ogg_linear_axis_dev", s.lnj + 1, s.lat0, s.lenlat, float(s.lnj), b["axis"].data_ptr(), st)
                rows = b["n"] + (1 if (b["needs_halo"] and self.halo == "recompute") else 0)
                self._timed("tile_latlon", lambda: L.call("ogg_tile_latlon_dev", rows, ni1, self._p(b["axis"], s.row0 + b["lo"]),
                                                          self.lon1d.data_ptr(), b["x"].data_ptr(), b["y"].data_ptr(), st))
            elif s.kind == "bipolar":
                if hasattr(L.load(), "ogg_bipolar_cap_mesh_angle_sym_dev"):
                    self._timed("bipolar_mesh", lambda: L.call("ogg_bipolar_cap_mesh_angle_sym_dev", p.Ni, s.Nj, s.lat0_bp, s.lon_bp, b["lo"], b["n"],
                                                               p.cap_symmetry, b["x"].data_ptr(), b["y"].data_ptr(), None, None,
                                                               b["angle_dx"].data_ptr(), st))
                else:   # an older build under A/B timing (OGG_LIB_PATH, scripts/ab_time.py): every column
                    self._timed("bipolar_mesh", lambda: L.call("ogg_bipolar_cap_mesh_angle_dev", p.Ni, s.Nj, s.lat0_bp, s.lon_bp, b["lo"], b["n"],
                                                               b["x"].data_ptr(), b["y"].data_ptr(), None, None, b["angle_dx"].data_ptr(), st))
            elif s.kind == "dpole":   # mesh, unwrap and angle_dx in one launch (its look-back words: the head of the band's workspace)
                self._timed("dpole_mesh", lambda: L.call("ogg_displaced_pole_grid_angle_ws_dev", p.Ni, s.Nj, p.lon0, s.lat0, s.lon_dp, s.r_dp,
                                                         s.row0 + b["lo"], b["n"], b["x"].data_ptr(), b["y"].data_ptr(),
                                                         b["angle_dx"].data_ptr(), b["ws"].data_ptr(), b["ws_mesh_bytes"], st))

    def exchange_halo(self):
        """First x/y row of the band above -> halo row of this band (MIDAS sub-grids only)."""
        if self.world == 1 or self.halo == "recompute" or self.latlon == "fused":
            return
        torch = self.torch
        if self.halo == "local":
            for s in self.plan.subs:
                b = self.buf[s.name]
                if b["needs_halo"]:
                    up = self.peers[self.rank + 1].buf[s.name]
                    b["x"][b["n"]].copy_(up["x"][0])
                    b["y"][b["n"]].copy_(up["y"][0])
            return
        import torch.distributed as dist

        ops = []
        for s in self.plan.subs:
            if s.kind not in ("mercator", "latlon"):
                continue
            b = self.buf[s.name]
            if b["needs_halo"]:  # receive from the rank above
                ops.append(dist.P2POp(dist.irecv, b["x"][b["n"]], self.rank + 1))
                ops.append(dist.P2POp(dist.irecv, b["y"][b["n"]], self.rank + 1))
            # send my first row to the rank below if IT needs a halo: it does iff it owns cell rows and is not the top band
            if self.rank > 0 and b["n"] > 0:
                lo_b, hi_b = self.rows_of(s, self.rank - 1, self.world)
                if min(hi_b, s.nj1 - 1) - lo_b > 0 and hi_b < s.nj1:
                    ops.append(dist.P2POp(dist.isend, b["x"][0], self.rank - 1))
                    ops.append(dist.P2POp(dist.isend, b["y"][0], self.rank - 1))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()

    def phase_b(self, only=None, kinds=None):
        """Metrics and angle of this rank's bands."""
        p, st = self.plan, self._stream()
        ni1 = p.Ni + 1
        for s in p.subs:
            b = self.buf[s.name]
            if b["n"] == 0 or not self._selected(s, only, kinds):
                continue
            if p.skip_metrics:
                for f in ("dx", "dy", "area"):
                    L.call("ogg_fill_dev", b[f].numel(), -1.0, b[f].data_ptr(), st)
            if s.kind in ("mercator", "latlon") and self.latlon == "fused":
                continue  # all six fields were written in phase A
            if s.kind in ("mercator", "latlon"):
                rows_xy = b["x"].shape[0]
                if p.skip_metrics:
                    self._timed("angle_x", lambda: L.call("ogg_grid_metrics_midas_dev", rows_xy, ni1, b["x"].data_ptr(), b["y"].data_ptr(),
                                                          b["n"], 0, p.Re, 1, None, None, None, b["angle_dx"].data_ptr(), st))
                else:
                    self._timed("midas_angle", lambda: L.call("ogg_grid_metrics_midas_dev", rows_xy, ni1, b["x"].data_ptr(),
                                                              b["y"].data_ptr(), b["n"], b["n_cell"], p.Re, 1, b["dx"].data_ptr(),
                                                              b["dy"].data_ptr() if b["n_cell"] else None,
                                                              b["area"].data_ptr() if b["n_cell"] else None,
                                                              b["angle_dx"].data_ptr(), st))
            else:
                if not p.skip_metrics:
                    if s.kind == "bipolar":
                        if hasattr(L.load(), "ogg_bipolar_cap_metrics_quad_sym_ws_dev"):
                            self._timed("bipolar_quad", lambda: L.call("ogg_bipolar_cap_metrics_quad_sym_ws_dev", 5, p.Ni, s.Nj, s.lat0_bp,
                                                                       s.lon_bp, s.rp, p.Re, b["lo"], b["n"], b["n_cell"], p.cap_symmetry,
                                                                       b["dx"].data_ptr(), b["dy"].data_ptr(), b["area"].data_ptr(),
                                                                       b["ws"].data_ptr(), b["ws_bytes"], st))
                        else:   # an older build under A/B timing
                            self._timed("bipolar_quad", lambda: L.call("ogg_bipolar_cap_metrics_quad_ws_dev", 5, p.Ni, s.Nj, s.lat0_bp, s.lon_bp,
                                                                       s.rp, p.Re, b["lo"], b["n"], b["n_cell"], b["dx"].data_ptr(),
                                                                       b["dy"].data_ptr(), b["area"].data_ptr(), b["ws"].data_ptr(),
                                                                       b["ws_bytes"], st))
                    else:
                        j0 = s.row0 + b["lo"]
                        has_sym = hasattr(L.load(), "ogg_displaced_pole_metrics_quad_form_sym_ws_dev")   # (not in an older build under A/B timing)
                        self._timed("dpole_quad", lambda: L.call(*(("ogg_displaced_pole_metrics_quad_form_sym_ws_dev", p.dp_arc, p.cap_symmetry) if has_sym
                                                                   else ("ogg_displaced_pole_metrics_quad_form_ws_dev", p.dp_arc)), 4, p.Ni, s.Nj, p.lon0,
                                                                 s.lat0, s.lon_dp, s.r_dp, p.Re, j0, b["n"], b["n_cell"], b["dx"].data_ptr(),
                                                                 b["dy"].data_ptr() if b["n_cell"] else None,
                                                                 b["area"].data_ptr() if b["n_cell"] else None,
                                                                 b["ws"].data_ptr() + b["ws_mesh_bytes"],
                                                                 b["ws_bytes"] - b["ws_mesh_bytes"], st))

    def step(self, time_kernels=False):
        """One full pass of the hot path for this rank's bands; outputs stay in HBM."""
        self._events = {} if time_kernels else None
        self.run_pass()

    def run_pass(self):
        """Phases A and B.  launch == "pass": everything through ogg_supergrid_pass_dev, on the caller's stream.  launch == "kernels":
        one call per sub-grid and phase; with `overlap` the caps run on side streams next to the lat-lon sub-grids (the HBM-bound
        lat-lon kernel overlaps the VALU-bound quadratures) and everything joins back on the caller's stream before returning
        (under graph capture this becomes a forked graph)."""
        torch = self.torch
        if self.latlon == "fused" and self.launch == "pass":
            self.tripolar_pass()
            if self.plan.skip_metrics:
                self.phase_b()  # the -1 fill of OGG:1327-1329
            return
        if self.latlon != "fused" or self.device.type != "cuda" or not self.overlap:
            self.phase_a()
            self.exchange_halo()
            self.phase_b()
            return
        main = torch.cuda.current_stream(self.device)
        caps = [s for s in self.plan.subs if s.kind in ("bipolar", "dpole")]
        # independent pieces of work: mesh (+ angle) and quadrature of every cap, one stream each
        tasks = []
        for s in caps:  # the quadratures are the longest chains: enqueue them first
            tasks.append(lambda s=s: self.phase_b(only=s.name))
        for s in caps:
            tasks.append(lambda s=s: self.phase_a(only=s.name))
        if self._side is None or len(self._side) < len(tasks):
            self._side = [torch.cuda.Stream(self.device) for _ in tasks]
        fork = torch.cuda.Event()
        fork.record(main)
        joins = []
        for st, task in zip(self._side, tasks):
            st.wait_event(fork)
            with torch.cuda.stream(st):
                task()
                e = torch.cuda.Event()
                e.record(st)
                joins.append(e)
        self.phase_a(kinds=("mercator", "latlon"))
        self.phase_b(kinds=("mercator", "latlon"))
        for e in joins:
            main.wait_event(e)

    def capture(self):
        """Capture one pass (phases A and B) into a HIP graph; `replay()` then costs one graph launch instead of ~10 kernel
        launches from Python.  Not available when the pass contains an RCCL halo exchange (stencil mode, world > 1)."""
        torch = self.torch
        if self.world > 1 and self.latlon == "stencil" and self.halo == "rccl":
            raise RuntimeError("the RCCL halo exchange is not captured; use eager steps")
        self._events = None
        self.step()
        torch.cuda.synchronize(self.device)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            self.run_pass()
        self.graph = g
        return g

    def replay(self):
        self.graph.replay()

    def kernel_times_ms(self):
        """Mean per-launch duration of each kernel over the events recorded since the last reset."""
        self.torch.cuda.synchronize(self.device)
        out = {}
        for k, evs in (self._events or {}).items():
            ts = [a.elapsed_time(b) for a, b in evs]
            out[k] = {"launches": len(ts), "total_ms": float(sum(ts)), "mean_ms": float(sum(ts) / len(ts))}
        return out

    # -- self-check --------------------------------------------------------------------------------------------
    def metrics_sums(self):
        """Device tensor (n_subs, 7): the five sums of ogg_metrics_sums_dev over this rank's band of every sub-grid, then the
        latitude of the sub-grid's first and of its last point row (each contributed by the one rank that owns that row)."""
        torch, p, st = self.torch, self.plan, self._stream()
        out = torch.zeros((len(p.subs), 7), dtype=torch.float64, device=self.device)
        for k, s in enumerate(p.subs):
            b = self.buf[s.name]
            if b["n"] == 0:
                continue
            first, last = b["lo"] == 0, b["hi"] == s.nj1
            col_a, col_b = ogg.metrics_error_columns(p.Ni, self._pole_column(s))
            L.call("ogg_metrics_sums_dev", b["n"], b["n_cell"], p.Ni, b["dx"].data_ptr(), b["dy"].data_ptr() if b["n_cell"] else None,
                   b["area"].data_ptr() if b["n_cell"] else None, col_a, col_b, int(first), int(last), out[k].data_ptr(), st)
            if first:
                out[k, 5].copy_(b["y"][0, 0])
            if last:
                out[k, 6].copy_(b["y"][b["n"] - 1, 0])
        return out

    def _pole_column(self, s):
        if s.kind != "dpole":
            return -999
        return int(self.plan.Ni * np.mod(s.lon_dp - self.plan.lon0, 360) / 360.0)   # OGG:1160 poles_i

    def metrics_error(self, sums=None):
        """The reference's CHECK_metrics lines (OGG:732-770) for every sub-grid of a band-sharded run: % errors of sum(area), of a
        meridian arc and of the parallel arc(s) against the sphere, from five sums per band taken on the device and ONE all-reduce
        (RCCL) of n_subs x 7 doubles over the ranks -- the only collective of the default pipeline.  Returns {sub-grid name:
        tuple as returned by the reference's metrics_error}.  ``sums``: precomputed metrics_sums() (tests)."""
        torch, p = self.torch, self.plan
        t = self.metrics_sums() if sums is None else sums
        if self.world > 1:
            if self.halo == "local":  # virtual ranks in one process
                if sums is None:
                    t = sum(g.metrics_sums() for g in self.peers if g is not self) + t
            else:
                all_reduce(t)
        v = t.cpu().numpy()
        out = {}
        for k, s in enumerate(p.subs):
            area, dy_a, dy_b, dx_first, dx_last, lat_first, lat_last = (float(x) for x in v[k])
            if s.kind == "bipolar":
                out[s.name] = ogg.metrics_error_from_sums(area, dy_a, dy_b, dx_first, dx_last, s.lat0_bp, 90.0, p.Re, bipolar=True)
            elif s.kind == "dpole":  # OGG:1161; with the doughnut rows skipped the area and dy sums are partial, as the reference warns
                e = ogg.metrics_error_from_sums(area, dy_a, dy_b, dx_first, dx_last, s.lat0, -90.0, p.Re,
                                                displaced_pole=self._pole_column(s))
                if s.row0 > 0:  # doughnut rows are not generated: area and meridian arc cannot be estimated (OGG:762-763 says so)
                    e = (float("nan"), float("nan"), e[2])
                out[s.name] = e
            elif s.name == "SC":     # regular southern cap, OGG:1146: (phiSC[-1, 0], phiSC[0, 0])
                out[s.name] = ogg.metrics_error_from_sums(area, dy_a, dy_b, dx_first, dx_last, lat_last, lat_first, p.Re)
            else:                    # Mercator, Southern Ocean: (phi[0, 0], phi[-1, 0])
                out[s.name] = ogg.metrics_error_from_sums(area, dy_a, dy_b, dx_first, dx_last, lat_first, lat_last, p.Re)
        return out

    # -- results ---------------------------------------------------------------------------------------------
    def check_lookback_flags(self):
        """Raise if a displaced-pole kernel of the last pass gave up waiting for a strip map (its results would be invalid; the wait
        is bounded so that a launch always drains -- never observed).  Synchronises the stream."""
        if self._pass_args is not None:   # the plan of the fused pass: the look-back words of BOTH its workspace slots
            flags = ctypes.c_int(0)
            L.call("ogg_supergrid_pass_plan_flags_dev", self._pass_args[4], ctypes.byref(flags), self._stream())
            if flags.value & 6:
                raise L.OggHipError(L.OGG_EHIP, "displaced-pole %s: a look-back wait timed out" % ("mesh" if flags.value & 2 else "quadrature"))
        for s in self.plan.subs:
            b = self.buf[s.name]
            if s.kind != "dpole" or b["n"] == 0:
                continue
            for off, what in ((0, "mesh"),) + (() if self.plan.skip_metrics else ((b["ws_mesh_bytes"], "quadrature"),)):
                flag = ctypes.c_int(0)
                L.call("ogg_workspace_error_flag_dev", b["ws"].data_ptr() + off, ctypes.byref(flag), self._stream())
                if flag.value != 0:
                    raise L.OggHipError(L.OGG_EHIP, "displaced-pole %s: a look-back wait timed out (flag %d)" % (what, flag.value))

    def south_cut(self):
        """plan.south_cut() with the cap's latitudes read from the device when --south_cutoff_ang needs them (world 1)."""
        first = self.plan.subs[0]
        y0 = None
        if first.name == "SC" and self.plan.south_cutoff_ang > -90 and self.plan.south_cutoff_row <= 0:
            assert self.world == 1
            y0 = self.buf["SC"]["y"][: self.buf["SC"]["n"], 0].cpu().numpy()
        return self.plan.south_cut(y0)

    def _pieces(self, cut, point_rows):
        """This rank's pieces of a stitched field, south -> north, as [(sub-grid, first band row, end band row, row of the stitched
        field where the piece starts)], and the field's total row count (OGG:1315-1365): fields on point rows (x, y, dx, angle_dx)
        drop the last row of every sub-grid but the northernmost; dy and area (cell rows) are concatenated whole.  cut: the triple
        of plan.south_cut().  Every band of every rank maps to ONE contiguous range of stitched rows."""
        c_sc, c_so, gone = cut
        subs = [s for s in self.plan.subs if not (s.name == "SC" and gone)]
        out, off = [], 0
        for k, s in enumerate(subs):
            b = self.buf[s.name]
            first = c_sc if s.name == "SC" else (c_so if (s.name == "SO" and gone) else 0)     # first kept row of the sub-grid
            last = (s.nj1 - (0 if k == len(subs) - 1 else 1)) if point_rows else s.nj1 - 1     # end of its kept rows
            kept = max(last - first, 0)
            r_lo, r_hi = b["lo"], (b["hi"] if point_rows else b["lo"] + b["n_cell"])             # this rank's rows of it
            lo, hi = max(r_lo, first), min(r_hi, last)
            if hi > lo:
                out.append((s, lo - b["lo"], hi - b["lo"], off + (lo - first)))
            off += kept
        return out, off

    def stitched_rows(self, cut):
        """nyp of the stitched grid"""
        return self._pieces(cut, True)[1]

    def stitched_column(self, field, col, cut, only=None):
        """One column of a stitched field (or of one sub-grid) as a host array: what main()'s guards and description need.  World 1."""
        assert self.world == 1
        torch = self.torch
        if only is not None:
            b = self.buf[only]
            return b[field][: b["n"], col].cpu().numpy()
        parts = [self.buf[s.name][field][lo:hi, col] for s, lo, hi, _ in self._pieces(cut, field in ("x", "y", "dx", "angle_dx"))[0]]
        return torch.cat(parts).cpu().numpy() if parts else np.zeros(0)

    def write_nc(self, fnam, cut, description=None, history=None, source=None, no_changing_meta=None, create=None, barrier=None):
        """write_nc of OGG:773-829 for the fields in HBM: the reference's layout (dimensions nyp, nxp, ny, nx, string(255); variables
        tile, y, x, dy, dx, area, angle_dx in that order; NetCDF-3 64-bit offset), every band streamed from the device into its byte
        range of the file (nc_stream).  Band-sharded runs write ONE file together: the rank with ``create`` (default: rank 0) writes
        the header and extends the file to its final size, then -- after ``barrier()`` (default: torch.distributed.barrier when a
        process group exists and world > 1) -- every rank streams its own bands to their offsets; no gather.  Returns (bytes this
        rank streamed, seconds)."""
        import os
        import time
        from . import nc_stream, netcdf3
        t0 = time.perf_counter()
        nyp, nx = self.stitched_rows(cut), self.plan.Ni
        ny = nyp - 1
        create = (self.rank == 0) if create is None else create
        if barrier is None:
            barrier = lambda: None   # noqa: E731
            if self.world > 1 and self.halo == "rccl":
                import torch.distributed as dist
                if dist.is_available() and dist.is_initialized():
                    barrier = dist.barrier
        gatts = []
        if not no_changing_meta:
            gatts = [("history", history or ""), ("description", description or ""), ("source", source or "")]
        ds = netcdf3.Dataset(fnam, [("nyp", nyp), ("nxp", nx + 1), ("ny", ny), ("nx", nx), ("string", 255)], gatts)
        ds.decl_var("tile", netcdf3.NC_CHAR, ("string",), [])
        spec = (("y", "y", ("nyp", "nxp"), "degrees"), ("x", "x", ("nyp", "nxp"), "degrees"), ("dy", "dy", ("ny", "nxp"), "meters"),
                ("dx", "dx", ("nyp", "nx"), "meters"), ("area", "area", ("ny", "nx"), "m2"), ("angle_dx", "angle_dx", ("nyp", "nxp"), "degrees"))
        for name, _, dims, units in spec:
            ds.decl_var(name, netcdf3.NC_DOUBLE, dims, [("units", units)])
        if create:
            print("   Writing netcdf file with ny,nx= ", ny, nx)
            fd = os.open(fnam, os.O_RDWR | os.O_CREAT | os.O_TRUNC, 0o644)
            try:
                ds.write_header(fd)
                os.pwrite(fd, b"tile1".ljust(255, b"\0"), ds.var_begin("tile"))
            finally:
                os.close(fd)
        barrier()   # the file exists at its final size
        fd = os.open(fnam, os.O_RDWR)
        try:
            stream = nc_stream.DeviceToFile(fd, self.device)
            try:
                for name, f, _, _ in spec:
                    row_bytes = (nx + 1 if f in ("x", "y", "dy", "angle_dx") else nx) * 8
                    for s, lo, hi, row0 in self._pieces(cut, f in ("x", "y", "dx", "angle_dx"))[0]:
                        stream.put(self.buf[s.name][f][lo:hi], ds.var_begin(name) + row0 * row_bytes)
            finally:
                stream.finish()   # also after an error: no writer thread may outlive the file descriptor
        finally:
            os.close(fd)
        barrier()   # every rank's bytes are in the file
        return stream.bytes, time.perf_counter() - t0

    def bands_to_host(self):
        """This rank's bands as numpy arrays (halo rows dropped): {sub: {field: array}}."""
        out = {}
        for s in self.plan.subs:
            b = self.buf[s.name]
            out[s.name] = {f: (b[f][: b["n"]] if f in ("x", "y") else b[f]).cpu().numpy() for f in FIELDS}
        return out


def stitch(plan, per_rank_bands, guards=False):
    """Concatenate the bands of all ranks per sub-grid, apply --south_cutoff_row / _ang to the southern pieces (OGG:1268-1313), then
    stitch the sub-grids south -> north as OGG:1315-1365 (x, y, dx, angle_dx drop the southern piece's last row; dy, area are
    concatenated whole).  ``guards``: raise like the reference on a grid it rejects (OGG:1371-1375, 1425-1436)."""
    subs = {}
    for s in plan.subs:
        subs[s.name] = {f: np.concatenate([bands[s.name][f] for bands in per_rank_bands], axis=0) for f in FIELDS}
    c_sc, c_so, gone = plan.south_cut(subs["SC"]["y"][:, 0] if "SC" in subs else None)
    if gone:
        del subs["SC"]
        subs["SO"] = {f: v[c_so:, :] for f, v in subs["SO"].items()}
    elif c_sc > 0:
        subs["SC"] = {f: v[c_sc:, :] for f, v in subs["SC"].items()}
    g = None
    for s in plan.subs:
        if s.name not in subs:
            continue
        piece = subs[s.name]
        if g is None:
            g = dict(piece)
        else:
            g = {f: np.concatenate((g[f] if f in ("dy", "area") else g[f][:-1, :], piece[f]), axis=0) for f in FIELDS}
    if guards:
        check_guards(g["y"][:, plan.Ni // 4], "BP" in subs)
    g["sub"] = subs
    return g


def check_guards(ycol, has_bp):
    """The reference's final checks on column Ni//4 of the stitched latitudes (OGG:1371-1375, 1425-1436)."""
    if has_bp and np.any((np.roll(ycol, shift=-1, axis=0) - ycol) == 0):
        raise Exception("lattitude array has repeated values along symmetry meridian!")
    equator_index = np.searchsorted(ycol, 0.0)
    if equator_index == 0:
        raise Exception("   Ooops: Equator is not in the grid")
    print("   Equator is at j=", equator_index)
    if equator_index % 2 == 0:
        raise Exception("Ooops: Equator is not going to be a u-point. Use option --south_cutoff_row to one more or on less row from south.")
    if ycol.shape[0] % 2 == 0:
        raise Exception("Ooops: The number of j's in the supergrid is not even. Use option --south_cutoff_row to one more or on less row from south.")
