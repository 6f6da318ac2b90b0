#!/usr/bin/env python3
"""Headline benchmark: supergrid cells/s (coordinates + metrics) of the 1/8 degree tripolar grid on N MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload r8|r8_latdp|r4_om4|r16|r2]

A step is one full pass of the hot path over the whole supergrid (all sub-grids: axes, coordinate tiles / meshes, halo
exchange, MIDAS metrics + angle, cap quadratures), inputs being ~10 scalars, outputs (six fp64 fields) left in HBM.
For N > 1 it runs one rank per GPU over RCCL: either launched under torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in
the environment), or by itself -- a plain `python bench.py --gpus N` starts the N ranks as child processes and relays rank 0's line.
Every sub-grid is split into N latitude bands and the total work is fixed, so scaling is "strong".  ONE JSON line on stdout.

Extra objects on that line:
  roofline      the kernel with the largest share of the step, priced as HBM traffic: algorithmic bytes per launch /
                mean launch duration (HIP events on the launch stream, inside the timed region) against 8 TB/s
  kernels       the same for every kernel of the step (the cap quadratures are fp64-VALU bound: their HBM fraction is
                small by construction)
  cpu_baseline  the numpy oracle (a port of the reference's algorithm, 1 core like the reference) on a bounded sample
                of the same workload: the southernmost 1/sample_div of the rows of every sub-grid
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
    "r8": dict(inverse_resolution=8.0),                                              # BASELINE.json configs[2]
    "r8_latdp": dict(inverse_resolution=8.0, lon_dp=80.0, lat_dp=-85.85),            # configs[3] (valid spelling)
    "r4_om4": dict(inverse_resolution=4.0, r_dp=0.2, south_cutoff_row=83),           # configs[1]
    "r16": dict(inverse_resolution=16.0),                                            # configs[4]
    "r2": dict(inverse_resolution=2.0),
    "r32": dict(inverse_resolution=32.0),   # beyond BASELINE.json: 1/32 degree, 405 M cells, 19.4 GB of fields (a size check of the 288 GB part)
}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec (6.3 TB/s achievable)

# algorithmic HBM bytes per output unit of each kernel: reads + writes that any implementation must do
ALG_BYTES = {
    "latlon_fused": ("point", 48),    # all six fields written once, nothing read
    "tile_latlon": ("point", 16),     # x, y written once
    "midas_angle": ("point", 48),     # x, y read (16 B) + dx, dy, area, angle_dx written (32 B)
    "angle_x": ("point", 24),         # x, y read + angle_dx written
    "bipolar_mesh": ("point", 24),    # x, y, angle_dx written; no reads
    "bipolar_quad": ("point", 24),    # dx, dy, area written; no reads
    "dpole_mesh": ("point", 24),      # x, y, angle_dx written (mesh, unwrap and angle in one launch); no reads
    "dpole_quad": ("point", 24),
}


def band_points(plan, rank, world, sg_mod):
    """points handled per launch by each kernel on this rank (for pricing launches in bytes)."""
    ni1 = plan.Ni + 1
    pts = {}
    for s in plan.subs:
        lo, hi = sg_mod.Supergrid.rows_of(s, rank, world)
        n = (hi - lo) * ni1
        if s.kind in ("mercator", "latlon"):
            pts["latlon_fused"] = [pts.get("latlon_fused", [0])[0] + n]   # all lat-lon bands go out in one launch
            pts.setdefault("tile_latlon", []).append(n)
            pts.setdefault("midas_angle", []).append(n)
        elif s.kind == "bipolar":
            pts.setdefault("bipolar_mesh", []).append(n)
            pts.setdefault("bipolar_quad", []).append(n)
        else:
            pts.setdefault("dpole_mesh", []).append(n)
            pts.setdefault("dpole_quad", []).append(n)
    return pts


def cpu_baseline(plan, sample_div, per_cell_loop=False):
    """Oracle timed on the southernmost 1/sample_div of the kept rows of every sub-grid of `plan` (1 thread, like the reference).  Every
    size comes from the plan's host-only attributes (Ni, axes, lat0_bp, Nj, the cap's kept rows after the doughnut and the south cuts),
    so the cells counted are the cells of the benched grid: with sample_div == 1 exactly plan.cells.
    ``per_cell_loop``: the quadratures average cell by cell in Python like the reference (OGG:176-187, 585-599) instead of
    vectorised over the chunk -- the "reference-shaped" row of SURVEY 8(d)."""
    from oracle import ogg_oracle as orc

    Ni = plan.Ni
    t0 = time.perf_counter()
    cells = 0
    lam = plan.lon0 + np.arange(Ni + 1) * plan.lenlon / float(Ni)

    def sample_rows(s):   # cell rows of the sample: the southernmost 1/div of the sub-grid's kept cell rows, at least one
        return max((s.nj1 - 1) // sample_div, 1) if s.nj1 > 1 else 0

    for s in plan.subs:
        n = sample_rows(s)
        if n == 0:
            continue
        if s.kind in ("mercator", "latlon"):
            if s.kind == "mercator":
                axis = s.explicit_axis if getattr(s, "explicit_axis", None) is not None else orc.phi_mercator(Ni, np.arange(s.y0, s.y0 + s.n_axis))
            else:
                axis = s.lat0 + np.arange(s.lnj + 1) * s.lenlat / float(s.lnj)
            axis = np.asarray(axis)[s.row0:s.row0 + n + 1]
            x = np.tile(lam, (n + 1, 1))
            y = np.tile(axis.reshape(-1, 1), (1, Ni + 1))
            if not plan.skip_metrics:
                orc.generate_grid_metrics_MIDAS(x, y)
            orc.angle_x(x, y)
        elif s.kind == "bipolar":
            latg = s.lat0_bp + np.arange(s.row0, s.row0 + n + 1) * (90 - s.lat0_bp) / float(s.Nj)
            lams, phis, _, _ = orc.bipolar_projection(np.tile(lam, (n + 1, 1)), np.tile(latg.reshape(-1, 1), (1, Ni + 1)), s.lon_bp, s.rp)
            orc.angle_x(lams, phis)
            if not plan.skip_metrics:
                orc.bipolar_cap_metrics_quad_fast(5, Ni, s.Nj, s.lat0_bp, s.lon_bp, s.rp, rows_per_chunk=16, j_first=s.row0, j_last=s.row0 + n,
                                                  per_cell_loop=per_cell_loop)
        else:   # displaced-pole cap: the rows main() keeps (behind the doughnut, OGG:1177-1186, and the south cut)
            x, y, _, _ = orc.displacedPoleCap_mesh(np.arange(Ni + 1), np.arange(s.row0, s.row0 + n + 1), Ni, s.Nj, plan.lon0, s.lat0, s.lon_dp, s.r_dp)
            orc.angle_x(x, y)
            if not plan.skip_metrics:
                orc.displacedPoleCap_metrics_quad(4, Ni, s.Nj, plan.lon0, s.lat0, s.lon_dp, s.r_dp, rows_per_chunk=8, j_first=s.row0,
                                                  j_last=s.row0 + n, per_cell_loop=per_cell_loop)
        cells += n * Ni
    dt = time.perf_counter() - t0
    if sample_div == 1:
        assert cells == plan.cells, (cells, plan.cells)
    return cells, dt


def _baseline_metric():
    """BASELINE.json's metric string, verbatim (the file ships with the repo)."""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:
        return "supergrid cells/sec (coords+metrics), 1/8\u00b0 tripolar at 1/2/4/8 MI355X"


METRIC = _baseline_metric()


def _dp_parity(arc):
    """Committed parity record of the displaced-pole quadrature (profiles/dp_parity.json, written from a GPU test run by
    scripts/make_dp_parity.py): how far each arc form is from the CPU oracle at full 1/8 degree size.  A static record of the test
    run, not a measurement of this process."""
    try:
        rec = json.load(open(os.path.join(ROOT, "profiles", "dp_parity.json")))
    except Exception:
        return None
    out = {"arc_form_timed": arc, "source": "profiles/dp_parity.json (tests/test_gpu_pipeline.py::test_full_size_r8_latdp_pass_vs_oracle on MI355X)",
           "max_rel_vs_oracle_r8_latdp": {form: {f: v["max_rel"] for f, v in rec["full_size_r8_latdp"][form].items()}
                                          for form in ("literal", "chord")},
           "chord_vs_literal_max_rel": rec["full_size_chord_vs_literal_max_rel"]}
    try:   # ... and how far the fp64 reference (oracle) and both forms are from a 50-digit evaluation of the reference's own formula
        tt = json.load(open(os.path.join(ROOT, "profiles", "r04_truth_table.json")))["dp_quadrature_OGG522_601"]
        out["max_rel_vs_exact_value_of_the_reference_formula_r8_latdp"] = {
            who: {f: tt["kept_rows/" + f][key]["max_rel"] for f in ("dx", "dy", "area")}
            for who, key in (("fp64_reference_numpy", "oracle_vs_truthA"), ("hip_literal", "hip_literal_vs_truthA"), ("hip_chord", "hip_chord_vs_truthA"))}
        out["truth_source"] = ("profiles/r04_truth_table.json: tests/test_gpu_truth.py on MI355X against tests/golden/truth_table.npz "
                               "(scripts/truth_table.py, mpmath 50 digits, 10 500 cells of the kept rows)")
    except Exception:  # noqa: BLE001 -- the record is optional
        pass
    return out


VALU_PEAK_GINSTR = 1024 * 2.4 / 4   # wave64 fp64 VALU instructions per ns over the chip: 1024 SIMDs, 2.4 GHz, 4 cycles per instruction


def valu_bound_roofline(hbm_roof, valu, launch_ms):
    """The roofline object of a launch that is fp64-VALU bound (its HBM fraction says nothing: a cap quadrature writes 24 B per cell behind
    thousands of fp64 instructions): priced in wave64 VALU instructions per second against 1024 SIMDs x 2.4 GHz / 4 cycles.  The instruction
    count per launch comes from the committed counters of the same kernel sources (`valu`); without them the HBM pricing stays, labelled."""
    if not valu or not valu.get("wave64_valu_instr_per_launch"):
        hbm_roof["limited_by"] = "fp64 VALU issue (no counters of this library's kernels under profiles/ to price it: HBM pricing kept)"
        return hbm_roof
    ach = valu["wave64_valu_instr_per_launch"] / (launch_ms * 1e-3) / 1e9
    return {"kernel": hbm_roof["kernel"], "bound": "fp64 VALU", "achieved": round(ach, 1), "peak": VALU_PEAK_GINSTR, "unit": "G wave64-instr/s",
            "frac": round(ach / VALU_PEAK_GINSTR, 4), "valu_busy_frac": valu.get("valu_busy_frac"), "traffic": hbm_roof.get("traffic"),
            "hbm": {"achieved": hbm_roof["achieved"], "peak": hbm_roof["peak"], "unit": "GB/s", "frac": hbm_roof["frac"]},
            "note": "the launch that takes the largest share of the step is fp64-VALU bound: wave64 VALU instructions per launch (committed "
                    "counters of the same kernel sources) / this run's launch duration, against 1024 SIMDs x 2.4 GHz / 4 cycles per fp64 "
                    "instruction; `valu_busy_frac` = SQ_ACTIVE_INST_VALU x 4 / SIMD cycles of the counter run; `hbm` = the same launch "
                    "priced in algorithmic bytes (24 B per cell: meaningless as a bound here, kept because north_star asks for it)",
            "source": valu.get("source")}


def time_other_arc(supergrid, plan_flags, arc, device, steps, torch, cap_symmetry=None):
    """ms per pass of the same workload with the OTHER arc form of the displaced-pole quadrature (a second set of band buffers; after the
    timed region)."""
    plan = supergrid.SupergridPlan(dp_arc=arc, cap_symmetry=cap_symmetry, **plan_flags)
    sg = supergrid.Supergrid(plan, rank=0, world=1, device=device, halo="recompute")
    sg.launch, sg.overlap = "pass", False
    for _ in range(30):
        sg.run_pass()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        sg.run_pass()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    return {"dp_arc": arc, "ms_per_step": ms, "value": plan.cells / (ms * 1e-3), "unit": "cells/s", "launch": "pass, eager", "steps": steps,
            "note": "same workload, the other arc form (opt-in for chord: OGG_DP_ARC / dp_arc / arc_form); not part of `value`"}


def power_probe(sg, torch, seconds, card_index=0):
    """Socket power and clocks (rocm-smi) while (a) the fused pass, (b) the stand-alone lat-lon kernel, (c) the stand-alone cap kernels run
    in a loop for `seconds` each, after the timed region.  The fused pass of the headline workload draws the socket's power limit
    (1.4 kW) and the shader clock drops ~13 % below what either kernel class holds alone: the third roofline."""
    import subprocess
    import threading

    def smi():
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--json"], stdout=subprocess.PIPE,
                                 stderr=subprocess.DEVNULL, text=True, timeout=10).stdout
            d = json.loads(out)
            cards = sorted(k for k in d if k.startswith("card"))
            c = d["card%d" % card_index] if ("card%d" % card_index) in d else d[cards[0]]
            num = lambda v: float(str(v).strip("()").lower().replace("mhz", ""))
            return {"watts": num(next(v for k, v in c.items() if "power" in k.lower())),
                    "sclk_mhz": num(next(v for k, v in c.items() if k.lower().startswith("sclk clock speed"))),
                    "mclk_mhz": num(next(v for k, v in c.items() if k.lower().startswith("mclk clock speed"))),
                    "junction_c": num(next(v for k, v in c.items() if "junction" in k.lower()))}
        except Exception:  # noqa: BLE001 -- no rocm-smi, no permission, another JSON layout: no probe
            return None

    if smi() is None:
        return None
    caps = [x.name for x in sg.plan.subs if x.kind in ("bipolar", "dpole")]
    saved = (sg.launch, sg.overlap, sg._events)
    sg._events = None

    def latlon_alone():
        sg.phase_a(kinds=("mercator", "latlon"))
        sg.phase_b(kinds=("mercator", "latlon"))

    def caps_alone():
        for c in caps:
            sg.phase_a(only=c)
            sg.phase_b(only=c)

    out = {}
    for name, launch, fn in (("fused_pass", "pass", sg.run_pass), ("latlon_kernel_alone", "kernels", latlon_alone),
                             ("cap_kernels_alone", "kernels", caps_alone)):
        if name == "cap_kernels_alone" and not caps:
            continue
        sg.launch, sg.overlap = launch, False
        samples, stop = [], threading.Event()

        def sampler():
            while not stop.is_set():
                v = smi()
                if v:
                    samples.append(v)
                stop.wait(0.3)

        th = threading.Thread(target=sampler)
        th.start()
        t0, n = time.perf_counter(), 0
        while time.perf_counter() - t0 < seconds:
            for _ in range(50):
                fn()
            torch.cuda.synchronize()
            n += 50
        dt = time.perf_counter() - t0
        stop.set()
        th.join()
        use = samples[1:] if len(samples) > 2 else samples     # the first sample may precede the ramp
        if use:
            out[name] = {k: round(sum(v[k] for v in use) / len(use), 1) for k in use[0]}
            out[name].update({"samples": len(use), "ms_per_call": round(dt / n * 1e3, 5),
                              "joules_per_call": round(out[name]["watts"] * dt / n, 5)})
    sg.launch, sg.overlap, sg._events = saved
    return out or None


def self_launch_command(n, argv, port):
    """The command `python bench.py --gpus N` runs for N > 1 when it was not started under torch.distributed.run: one rank per GPU
    of this node, rendezvous on 127.0.0.1 (the container's hostname may not resolve)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(n, argv, timeout=None):
    """Run the N ranks as children of this process (which has not touched the GPU and never does), pass their stderr through, print
    rank 0's ONE JSON line on stdout and return the exit code for this process: the children's if they failed, 1 if no line came."""
    import socket
    import subprocess
    import threading

    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    for k in ("RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    for attempt in (0, 1):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        # (the port is free NOW; somebody else may take it before torch.distributed.run binds it: one retry with another port)
        cmd = self_launch_command(n, argv, port)
        # the ranks' stderr is passed through AS IT COMES (a reader thread: a hang at N ranks leaves its diagnostics on the terminal, and a
        # timeout loses nothing) and kept for the one check below; stdout (rank 0's JSON line) is collected
        p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
        err_lines, out_lines = [], []

        def pump(stream, keep, echo):
            for line in stream:
                keep.append(line)
                if echo:
                    sys.stderr.write(line)
                    sys.stderr.flush()

        readers = [threading.Thread(target=pump, args=(p.stderr, err_lines, True), daemon=True),
                   threading.Thread(target=pump, args=(p.stdout, out_lines, False), daemon=True)]
        for th in readers:
            th.start()
        try:
            p.wait(timeout=timeout)
        except subprocess.TimeoutExpired:
            p.kill()
            for th in readers:
                th.join(5)
            sys.stderr.write("bench.py: the %d ranks did not finish within %s s (their stderr is above)\n" % (n, timeout))
            return 124
        for th in readers:
            th.join(5)
        stdout = "".join(out_lines)
        stderr = "".join(err_lines)
        if p.returncode == 0 or attempt == 1 or not any(t in stderr for t in ("EADDRINUSE", "Address already in use", "address already in use")):
            break
        sys.stderr.write("bench.py: the rendezvous port %d was taken between choosing and binding it; once more with another port\n" % port)
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    for l in stdout.splitlines():
        if not l.startswith("{"):
            sys.stderr.write(l + "\n")
    if p.returncode != 0:
        sys.stderr.write("bench.py: torch.distributed.run with %d ranks failed (exit code %d): %s\n" % (n, p.returncode, " ".join(cmd)))
        return p.returncode
    if not lines:
        sys.stderr.write("bench.py: rank 0 printed no result line\n")
        return 1
    print(lines[-1])
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="r8", choices=sorted(WORKLOADS))
    ap.add_argument("--halo", default="rccl", choices=["rccl", "recompute"], help="halo source of the stencil pipeline")
    ap.add_argument("--latlon", default="fused", choices=["fused", "stencil"],
                    help="fused: lat-lon sub-grids from their 1-D axes in one kernel (no reads, no halo); stencil: tile, RCCL halo, "
                         "generic 2x3-stencil kernel")
    ap.add_argument("--cpu-sample-div", type=int, default=1, help="CPU baseline runs 1/div of the rows of every sub-grid (0: skip)")
    ap.add_argument("--as-rank", type=int, default=None, help="experiment: run only the bands of this rank of --as-world on one GPU")
    ap.add_argument("--as-world", type=int, default=1)
    ap.add_argument("--overlap", type=int, default=1, help="run the caps on side streams next to the lat-lon sub-grids")
    ap.add_argument("--graph", type=int, default=1, help="replay the step from a captured HIP graph in the timed region (0: eager launches)")
    ap.add_argument("--d2h", type=int, default=1, help="after the timed region: time the copy of this rank's bands to pinned host memory (information)")
    ap.add_argument("--checksum", type=int, default=1, help="after the timed region: 64-bit sums of the bit patterns of every field of every "
                    "sub-grid, added over the ranks (one all-reduce): a band-sharded run must print the single-GPU values")
    ap.add_argument("--self-check", type=int, default=1, help="after the timed region: % errors of area / arcs per sub-grid (device sums + all-reduce)")
    ap.add_argument("--power-probe", type=float, default=1.5, help="after the timed region (1 GPU): seconds per phase of the rocm-smi power / "
                    "clock probe (fused pass, lat-lon kernel alone, cap kernels alone); 0: skip")
    ap.add_argument("--dp-arc", default="chord", choices=["chord", "literal"],
                    help="arc form of the displaced-pole quadrature that `value` / `ms_per_step` are measured on (workloads with a displaced pole "
                         "only): chord (default since round 4, what main() and the Python entry points run unless told otherwise: the same "
                         "finite-difference stencil, distances from the probes' positions on the sphere) or literal (the reference's haversine "
                         "arithmetic, a fourth launch).  Against a 50-digit evaluation of the reference's own formula the chord form is CLOSER "
                         "than the fp64 reference itself (profiles/r04_truth_table.json; `parity`).  The other form is timed after the timed "
                         "region and reported as `dp_arc_other` (--dp-arc-other 0: skip)")
    ap.add_argument("--dp-arc-other", type=int, default=1)
    ap.add_argument("--cap-symmetry", default="mirror", choices=["mirror", "none"],
                    help="mirror (default, what main() runs): the bipolar cap from a quarter of its columns, the displaced-pole quadrature from "
                         "half of them, written to their mirror images (DESIGN.md 2: as far from the exact value of the reference's formula as "
                         "the reference's own columns); none: every column evaluated, as the reference does (rounds 1-4)")
    ap.add_argument("--tune-strips", type=int, default=1,
                    help="set-up, untimed, one GPU, fused pass: time the pass with a few numbers of resident lat-lon strip workgroups "
                         "(OGG_PASS_LL_WG, a bit-neutral tiling knob) and keep the fastest -- which count suits the write path differs "
                         "between boxes and between the states of one box (DESIGN.md 4.1); 0: the library's default")
    ap.add_argument("--launch", default="auto", choices=["auto", "pass", "kernels"],
                    help="pass: ogg_tripolar_pass_dev (three launches, lat-lon and cap workgroups share them); kernels: one launch per "
                         "sub-grid and phase (--overlap: caps on side streams); auto: time both during set-up and keep the faster")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: start the N ranks as CHILD processes, before anything here touches the GPU (no exec)
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d does not match WORLD_SIZE %d" % (args.gpus, world))
    # OGG_BENCH_ONE_GPU=1: a REHEARSAL of the N-rank run on a box with one GPU -- every rank on cuda:0, the collectives over gloo (RCCL
    # does not put two ranks on one device).  Everything else is the N-GPU run: the launcher, the bands of each rank, the barriers,
    # the reductions, the gathered per-rank lines.  The output says so ("rehearsal"); its times are those of N processes sharing a card.
    rehearsal = world > 1 and bool(os.environ.get("OGG_BENCH_ONE_GPU"))
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = "cuda:%d" % local_rank
    use_dist = world > 1 or bool(os.environ.get("OGG_FORCE_DIST"))   # OGG_FORCE_DIST: exercise the RCCL calls at world size 1
    if use_dist:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device(device))

    from ocean_model_grid_generator_amd import _lib, supergrid

    flags = WORKLOADS[args.workload]
    plan = supergrid.SupergridPlan(dp_arc=args.dp_arc, cap_symmetry=(args.cap_symmetry == "mirror"), **flags)
    # band split: the last rank's share follows from two timings (the fix-up launch only it runs, a whole pass) that rank 0 takes on
    # THIS box before any band exists and broadcasts (OGG_SPLIT_CALIBRATE=0: the fitted constants of rounds 2-3)
    if os.environ.get("OGG_SPLIT_CALIBRATE", "1") != "0":
        if args.as_rank is not None and args.as_world > 1:
            plan.calibrate_split(device, rank=0, world=args.as_world, broadcast=False)
        elif world > 1 or use_dist:   # (use_dist at world 1 = OGG_FORCE_DIST: the same calls over RCCL on a one-GPU box)
            plan.calibrate_split(device, rank=rank, world=world, force=True)
    if args.as_rank is not None:  # single-GPU rehearsal of one rank's share
        sg = supergrid.Supergrid(plan, rank=args.as_rank, world=args.as_world, device=device, halo="recompute", latlon=args.latlon)
    else:
        sg = supergrid.Supergrid(plan, rank=rank, world=world, device=device, halo=args.halo, latlon=args.latlon)

    sg.overlap = bool(args.overlap)

    def sync():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    # set-up, untimed: 60 passes so that the clocks have ramped before anything is compared or timed
    if args.latlon == "fused":
        sg.launch = "pass"
    for _ in range(60):  # a fixed count: in the stencil pipeline every pass holds a neighbour exchange
        sg.run_pass()
    torch.cuda.synchronize()
    if args.latlon == "fused":
        # ... and for at least 60 ms: the clocks of a fresh process take ~30 ms of WORK to ramp, and one rank of eight does 60 passes in
        # 2 ms (no collective in a fused pass: every rank may run its own count)
        t_warm = time.perf_counter()
        while time.perf_counter() - t_warm < 0.06:
            for _ in range(20):
                sg.run_pass()
            torch.cuda.synchronize()
    # Several ranks: every rank times its OWN share on its own GPU, all at once (the job's conditions; calibrate_split timed the shares one
    # after the other on rank 0's idle chip), the times are all-gathered, and up to two rebalancing steps scale every rank's share by
    # mean(T) / T_rank -- the same arithmetic on the same list on every rank, hence the same edges.  Every step is measured again; one that
    # made the slowest rank slower is taken back.
    if args.latlon == "fused" and world > 1 and args.as_rank is None and os.environ.get("OGG_SPLIT_SELF_CALIBRATE", "1") != "0":
        def own_us(n=200):
            sync()
            ta = time.perf_counter()
            for _ in range(n):
                sg.run_pass()
            torch.cuda.synchronize()
            mine = (time.perf_counter() - ta) / n * 1e6
            gathered = [None] * world
            dist.all_gather_object(gathered, mine)
            return gathered

        def rebuild():
            g = supergrid.Supergrid(plan, rank=rank, world=world, device=device, halo=args.halo, latlon=args.latlon)
            g.overlap, g.launch = bool(args.overlap), "pass"
            for _ in range(100):
                g.run_pass()
            torch.cuda.synchronize()
            return g

        before = own_us()
        for _ in range(2):
            kept = (getattr(plan.subs[0], "top_capacity", None), getattr(plan.subs[0], "rank_capacity", None))
            if not plan.refine_split(before, world):
                break
            sg.close()
            sg = rebuild()
            after = own_us()
            rec = plan.split_times["self_calibration"][-1]
            rec["per_rank_us_after"] = [round(v, 3) for v in after]
            if max(after) > 1.01 * max(before):
                hist = plan.split_times["self_calibration"]
                plan.set_split_times(plan.split_times["tail_us"], plan.split_times["pass_us"], plan.split_times["source"],
                                     top_capacity=kept[0], rank_capacity=kept[1])
                rec["taken_back"] = True
                plan.split_times["self_calibration"] = hist
                sg.close()
                sg = rebuild()
                break
            before = after
    sg.launch = "kernels"
    can_graph = not (world > 1 and args.latlon == "stencil" and args.halo == "rccl")
    use_graph = bool(args.graph) and can_graph
    tuned = None
    has_dp = any(s.kind == "dpole" for s in plan.subs)
    if args.latlon == "fused" and args.launch == "auto" and world > 1:
        # several ranks: the fused pass, which exists for small shares (a side stream's dependency costs 10-20 us, as much as a rank's
        # kernels at 8 ranks); a 30-pass timing of a 40 us share is too noisy to choose by (a rehearsal picked "kernels" for one rank of
        # four and made it the slowest)
        sg.launch, sg.overlap, use_graph = "pass", False, False
    elif args.latlon == "fused" and args.launch == "auto":
        # set-up, untimed: every rank keeps the launch scheme that is fastest for ITS share (there is no collective in a pass)
        tuned = {}
        # (the pass is launched eagerly: a graph of its three launches gains nothing and would hide them from the events)
        cands = [("pass", 0, 0), ("kernels", 1, 1), ("kernels", 0, 0)]
        for launch, overlap, graph in cands:
            sg.launch, sg.overlap = launch, bool(overlap)
            if graph and not can_graph:
                continue
            if graph:
                sg.capture()
            for _ in range(10):
                sg.replay() if graph else sg.run_pass()
            torch.cuda.synchronize()
            ta = time.perf_counter()
            for _ in range(30):
                sg.replay() if graph else sg.run_pass()
            torch.cuda.synchronize()
            tuned["%s overlap=%d graph=%d" % (launch, overlap, graph)] = (time.perf_counter() - ta) / 30 * 1e3
        best = min(tuned, key=tuned.get)
        launch, overlap, graph = best.split()
        sg.launch, sg.overlap, use_graph = launch, overlap.endswith("1"), graph.endswith("1")
    elif args.latlon == "fused":
        sg.launch = args.launch
        if args.launch == "pass":
            use_graph = False
    # set-up, untimed: the number of resident strip workgroups of the fused pass.  The library's defaults are the counts that won on the
    # boxes they were swept on; the write path of another box, or of the same box ten minutes later, may prefer a neighbour (1/16 degree:
    # 161 workgroups are 5 % faster than 72 on a box in its fast state and 8 % slower on one in its slow state).  Round-robin, two rounds
    # of 30 passes per candidate; the knob is read when the plan is (re)built and changes no bit of the result.
    strips_tuned = None
    if (args.tune_strips and world == 1 and args.as_rank is None and args.latlon == "fused" and sg.launch == "pass" and not use_graph
            and "OGG_PASS_LL_WG" not in os.environ):
        try:
            cands = [None, "72", "96", "108", "120", "138", "161"]
            times = {c: [] for c in cands}
            for rnd in range(2):
                for c in cands:
                    if c is None:
                        os.environ.pop("OGG_PASS_LL_WG", None)
                    else:
                        os.environ["OGG_PASS_LL_WG"] = c
                    sg.replan()
                    for _ in range(10):
                        sg.run_pass()
                    torch.cuda.synchronize()
                    ta = time.perf_counter()
                    for _ in range(30):
                        sg.run_pass()
                    torch.cuda.synchronize()
                    times[c].append((time.perf_counter() - ta) / 30 * 1e3)
            best = min(cands, key=lambda c: min(times[c]))
            # a neighbour must beat the default by more than a timing can be off by (1 %) to replace it
            if best is not None and min(times[best]) > 0.99 * min(times[None]):
                best = None
            if best is None:
                os.environ.pop("OGG_PASS_LL_WG", None)
            else:
                os.environ["OGG_PASS_LL_WG"] = best
            sg.replan()
            strips_tuned = {"candidates_ms": {("default" if c is None else c): [round(v, 5) for v in times[c]] for c in cands},
                            "kept": "default" if best is None else best}
        except Exception as exc:  # noqa: BLE001 -- never lose the bench line over the tuning: the library's default stands
            os.environ.pop("OGG_PASS_LL_WG", None)
            sg.replan()
            strips_tuned = {"error": repr(exc)}
    for _ in range(args.warmup):
        sg.step()
    if use_graph:
        sg.capture()
        sg.replay()  # one untimed replay
    sync()
    # timed region: exactly K passes.  In pass mode 2-3 of the passes also record HIP events around their three launches
    # (on the launch stream, by the library itself): the per-launch durations of the roofline object come from the timed region.
    sample = sg.launch == "pass" and not use_graph
    # 2-3 sampled passes: the event records cost ~7 us per sampled pass, and a sampled pass runs its table launch itself (~4 us) instead of
    # finding its tables built by the previous pass's launch B.  They run right AFTER the timed region, which holds K plain passes -- at
    # every N (round 4 kept them inside at N = 1 only: two definitions of the region on one scaling curve)
    sample_inside = False
    stride = max(1, args.steps // 2)
    if sample:
        sg.reserve_pass_events(args.steps // stride + 1)
        events = []
    carried0 = sg.pass_plan_info()[1]
    # the region on three host clocks: scripts/trace_window.py finds the K timed launches in a rocprofv3 kernel trace of this command by them
    clocks = {"monotonic": time.CLOCK_MONOTONIC, "boottime": getattr(time, "CLOCK_BOOTTIME", time.CLOCK_MONOTONIC), "realtime": time.CLOCK_REALTIME}
    region_ns = {k: [time.clock_gettime_ns(c)] for k, c in clocks.items()}
    t0 = time.perf_counter()
    for k in range(args.steps):
        if use_graph:
            sg.replay()
        else:
            if sample_inside:
                sg.pass_events = events if k % stride == 0 else None
            sg.run_pass()
    torch.cuda.synchronize()
    # This rank's own K passes, from the opening barrier + synchronize to ITS synchronize.  A pass has no inter-rank dependency (no
    # collective, no halo in the default pipeline), so the job's wall time is the MAX of this over the ranks: that is `value`.
    t_end = time.perf_counter()
    dt_local = t_end - t0
    for k, c in clocks.items():
        region_ns[k].append(time.clock_gettime_ns(c))
    if use_dist:
        dist.barrier()
        torch.cuda.synchronize()
    # ... and the same through a closing barrier + synchronize (an all-reduce and two host wake-ups: tens of microseconds, which is
    # noise at one rank's 5 ms and a tenth of the region at eight ranks' 0.7 ms): reported beside it, `ms_per_step_with_closing_barrier`
    dt_barrier = time.perf_counter() - t0
    plan_slots, carried1 = sg.pass_plan_info()
    launches = None
    if sample:
        if not sample_inside:
            for k in range(args.steps // stride + 1):
                sg.pass_events = events
                sg.run_pass()
        sg.pass_events = events
        launches = sg.pass_launch_times_ms()
        sg.pass_events = None
    # `value`: the JOB's wall time for its K passes = latest end over the ranks - earliest start over the ranks, on the node's common
    # monotonic clock (one node: every rank reads the same CLOCK_MONOTONIC) -- the ranks leave the opening barrier a few microseconds
    # apart, so this is >= every rank's own elapsed time; the same expression at N = 1 is that rank's elapsed time
    dt = dt_slowest = dt_local
    if use_dist:
        t = torch.tensor([dt_barrier, dt_local, -t0, t_end], dtype=torch.float64, device=device)
        supergrid.all_reduce(t, op=dist.ReduceOp.MAX)
        dt_barrier, dt_slowest = float(t[0].item()), float(t[1].item())
        dt = float(t[3].item()) + float(t[2].item())   # max(end) - min(start)
    # self-check, untimed: the reference's CHECK_metrics numbers (OGG:732-770) for the bands now in HBM -- five sums per band on the
    # device and one all-reduce (RCCL) of n_subs x 7 doubles, the only collective of the default pipeline
    self_check = None
    if not plan.skip_metrics and args.self_check:
        try:
            if args.as_rank is None:
                self_check = {k: [None if e != e else float(e) for e in v] for k, v in sg.metrics_error().items()}  # NaN (not estimable) -> null
        except Exception as exc:  # never lose the bench line over the self-check
            self_check = {"error": repr(exc)}
    # bit-level fingerprint of the result: per sub-grid and field the wrapping 64-bit sum of the values' bit patterns over this rank's
    # band, added over the ranks -- independent of the band decomposition, so N ranks must reproduce the 1-GPU numbers exactly
    checksums = None
    if args.checksum and args.as_rank is None:
        try:
            cs = torch.zeros((len(plan.subs), len(supergrid.FIELDS)), dtype=torch.int64, device=device)
            for a, s_ in enumerate(plan.subs):
                b = sg.buf[s_.name]
                for c, f in enumerate(supergrid.FIELDS):
                    t = b[f][: b["n"]] if f in ("x", "y") else b[f]
                    if t.numel():
                        cs[a, c] = t.contiguous().view(torch.int64).sum()
            if use_dist:
                supergrid.all_reduce(cs)
            checksums = {s_.name: {f: "%016x" % (int(cs[a, c].item()) & 0xFFFFFFFFFFFFFFFF) for c, f in enumerate(supergrid.FIELDS)}
                         for a, s_ in enumerate(plan.subs)}
        except Exception as exc:
            checksums = {"error": repr(exc)}
    per_rank = None
    if use_dist:   # every rank's own time and launch scheme (rank 0 prints them)
        mine = {"rank": rank, "ms_per_step": dt_local / args.steps * 1e3, "launch": "%s graph=%d" % (sg.launch, int(use_graph)), "device": local_rank}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    # for information (SURVEY 8d ii): this rank's bands copied to pinned host memory, after the pass
    d2h = None
    if args.d2h:
        try:
            pinned = [(b[f], torch.empty(b[f].shape, dtype=b[f].dtype, pin_memory=True)) for b in sg.buf.values() for f in supergrid.FIELDS]
            for _ in range(2):
                torch.cuda.synchronize()
                td = time.perf_counter()
                for dev, host in pinned:
                    host.copy_(dev, non_blocking=True)
                torch.cuda.synchronize()
                td = time.perf_counter() - td
            nbytes = sum(dev.numel() * 8 for dev, _ in pinned)
            d2h = {"ms": td * 1e3, "bytes": nbytes, "GBps": nbytes / td / 1e9}
            del pinned
        except Exception as exc:
            d2h = {"error": repr(exc)}
    dp_other = None
    if has_dp and args.dp_arc_other and world == 1 and args.as_rank is None and args.latlon == "fused":
        try:
            dp_other = time_other_arc(supergrid, flags, "chord" if args.dp_arc == "literal" else "literal", device, args.steps, torch,
                                      cap_symmetry=(args.cap_symmetry == "mirror"))
        except Exception as exc:  # never lose the bench line over the secondary number
            dp_other = {"error": repr(exc)}
    # what this box's write path gives a plain fill right now, in this process (after the timed region): the boxes of the pool differ by 20 %
    # in what they sustain and one box moves by as much within minutes, so a launch's GB/s is best read beside it
    fill_ref = fill_ref_sized = None
    if rank == 0 and args.as_rank is None:
        def time_fill(n_doubles, reps):
            buf = torch.empty(n_doubles, dtype=torch.float64, device=device)
            for _ in range(3):
                buf.fill_(1.0)
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ev0.record()
            for _ in range(reps):
                buf.fill_(1.0)
            ev1.record()
            torch.cuda.synchronize()
            rate = round(reps * n_doubles * 8 / (ev0.elapsed_time(ev1) * 1e-3) / 1e9, 1)
            del buf
            return rate
        try:
            fill_ref = time_fill(1 << 27, 20)   # 1 GiB
            # ... and a fill of as many bytes as one pass of this workload writes (48 B per cell), repeated like the passes are: a 1 GiB
            # buffer written over and over keeps a quarter of its lines in the 256 MB Infinity Cache between two fills, a 4.9 GB one
            # (1/16 degree) next to nothing -- the pass of a large grid is best read beside a fill of its own size
            n_pass = plan.cells * 6 // world
            if n_pass > (3 << 26):
                fill_ref_sized = time_fill(n_pass, 8)
        except Exception as exc:  # never lose the bench line over the reference
            pass
    power = None
    if args.power_probe > 0 and world == 1 and args.latlon == "fused" and args.as_rank is None:
        try:
            power = power_probe(sg, torch, args.power_probe, local_rank)
        except Exception as exc:  # never lose the bench line over the probe
            power = {"error": repr(exc)}
    # per-kernel durations: the same K passes again, launched eagerly with HIP events around every kernel on the launch
    # stream (events cannot be read back from inside a replayed graph)
    timed_launch = "%s%s%s" % (sg.launch, ", caps on side streams" if (sg.launch == "kernels" and sg.overlap) else "",
                               ", hip graph replay" if use_graph else ", eager")
    sg._events = {}
    sg.launch = "kernels"
    sg.overlap = False  # kernels one after the other on one stream: undisturbed per-kernel durations
    t1 = time.perf_counter()
    for _ in range(args.steps):
        sg.run_pass()
    torch.cuda.synchronize()
    dt_eager = time.perf_counter() - t1
    ktimes = sg.kernel_times_ms()

    if rank == 0:
        pts = band_points(plan, rank, world, supergrid)
        kernels = {}
        for k, v in ktimes.items():
            unit, b = ALG_BYTES[k]
            launches_per_step = len(pts[k])
            alg_bytes = b * float(sum(pts[k])) / launches_per_step      # mean algorithmic bytes per launch
            gbs = alg_bytes / (v["mean_ms"] * 1e-3) / 1e9
            kernels[k] = {"launches_per_step": launches_per_step, "mean_ms": round(v["mean_ms"], 5),
                          "ms_per_step": round(v["total_ms"] / args.steps, 5), "alg_bytes_per_launch": int(alg_bytes),
                          "alg_GBps": round(gbs, 1), "hbm_frac": round(gbs / HBM_PEAK_GBS, 4)}
        # counters kept under profiles/ (builder-side rocprofv3 --pmc runs of this command): quoted only when they were taken with a
        # library built from the SAME kernel sources as the one loaded now (ogg_version() carries the source hash)
        lib_hash = _lib.source_hash()
        counters_note = {"lib_src_hash": lib_hash}

        def committed(fname, key):
            path = os.path.join(ROOT, "profiles", fname)
            rec = json.load(open(path)).get(key, {}) if os.path.exists(path) else {}
            h = rec.get("_lib_src_hash") if isinstance(rec, dict) else None
            counters_note[fname] = {"key": key, "collected_with_src_hash": h, "quoted": bool(rec) and h == lib_hash}
            return rec if (rec and h == lib_hash) else {}

        ckey = args.workload + (" --dp-arc literal" if (has_dp and args.dp_arc == "literal") else "") + (" --cap-symmetry none" if args.cap_symmetry == "none" else "")
        pmc = committed("hbm_traffic.json", ckey)
        valu = committed("valu_counters.json", ckey)
        roof_valu = None
        if launches:  # the launches of the fused pass, timed inside the timed region
            n_sampled = launches.pop("sampled_passes")
            if launches.get("pass_dpquad", {}).get("alg_bytes", 0) == 0:
                launches.pop("pass_dpquad", None)   # no fourth launch in this pass (two events back to back)
            for k, v in launches.items():
                gbs = v["alg_bytes"] / (v["ms"] * 1e-3) / 1e9 if v["ms"] > 0 else 0.0
                v.update({"ms": round(v["ms"], 5), "alg_bytes": int(v["alg_bytes"]), "alg_GBps": round(gbs, 1),
                          "hbm_frac": round(gbs / HBM_PEAK_GBS, 4)})
            dom = max(launches, key=lambda k: launches[k]["ms"])
            # launch B of a plan handle reads the strips' row scalars from a table (pass_b_kernel<5, true>: 128 VGPRs, four waves per
            # SIMD); the one-shot entry points and one-slot plans run pass_b_kernel<5, false>
            kname = {"pass_a": "pass_a_kernel<5>", "pass_b": "pass_b_kernel<5, true>", "pass_tail": "bipolar_quad_tail_kernel<5>",
                     "pass_dpquad": "pass_d_kernel<4>"}[dom]
            if dom == "pass_b" and kname not in valu:
                kname = next((k for k in valu if k.startswith("pass_b_kernel<5")), kname)
            vc = valu.get(kname)
            roof = {"kernel": kname, "bound": "hbm", "achieved": launches[dom]["alg_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": launches[dom]["hbm_frac"], "traffic": pmc.get(dom),
                    "traffic_source": "profiles/hbm_traffic.json: FETCH_SIZE / WRITE_SIZE passes of a builder-side rocprofv3 run of this command "
                                      "(read side x2), committed -- not counters of this process; null when those counters were taken with "
                                      "other kernel sources than the loaded library's (`counters`)",
                    "limited_by": "fp64 VALU issue, not HBM (see roofline_valu)" if (vc and vc["valu_busy_frac"] > 0.6) else None,
                    "note": "longest of the launches of the fused pass; HIP events recorded by the library on the launch stream in "
                            "%d passes right after the timed region.  It carries lat-lon row strips (HBM-write bound) AND cap mesh / "
                            "quadrature workgroups (fp64 VALU; with mirrored cap columns a third of round 4's arithmetic): the launch is "
                            "bound by its writes (DESIGN.md 4.1) unless `limited_by` says otherwise.  `plain_fill_GBps`: a 1 GiB "
                            "torch fill_ timed in this process after the region -- what this box's write path gives ONE contiguous stream "
                            "right now (`plain_fill_of_the_pass_size_GBps`: the same for a buffer of 48 B x cells, when that is more than "
                            "1.5 GiB: a buffer that size leaves nothing in the 256 MB Infinity Cache between two fills).  `kernels` lists the stand-alone kernels, one after the other." % n_sampled}
            if fill_ref:
                roof["plain_fill_GBps"] = fill_ref
                roof["achieved_over_plain_fill"] = round(roof["achieved"] / fill_ref, 4)
            if fill_ref_sized:
                roof["plain_fill_of_the_pass_size_GBps"] = fill_ref_sized
                roof["achieved_over_plain_fill_of_the_pass_size"] = round(roof["achieved"] / fill_ref_sized, 4)
            if vc:
                # issue-time floor of the launch: every wave64 VALU instruction occupies its SIMD for >= 4 cycles (quarter-rate fp64
                # rcp/rsq/sqrt: 16), 1024 SIMDs, at the clock the counter run held
                clock_ghz = vc["gui_cycles_per_xcd"] / (launches[dom]["ms"] * 1e-3) / 1e9
                roof_valu = {"kernel": kname, "bound": "fp64 VALU issue", "valu_busy_frac": round(vc["valu_busy_frac"], 4),
                             "wave64_valu_instr_per_launch": vc["wave64_valu_instr"], "fp64_transcendental_instr": vc["fp64_transcendental_instr"],
                             "issue_floor_ms": round(vc["valu_busy_frac"] * vc["gui_cycles_per_xcd"] / (clock_ghz * 1e9) * 1e3, 5) if clock_ghz > 0 else None,
                             "launch_ms": launches[dom]["ms"],
                             "source": "profiles/valu_counters.json: SQ_INSTS_VALU, SQ_ACTIVE_INST_VALU, GRBM_GUI_ACTIVE of a builder-side "
                                       "rocprofv3 --pmc run of this command, committed -- not counters of this process; launch_ms is this "
                                       "run's own event time"}
            if roof["frac"] < 0.25:
                roof = valu_bound_roofline(roof, roof_valu, launches[dom]["ms"])
        else:
            dom = max(kernels, key=lambda k: kernels[k]["ms_per_step"])
            per_step = pmc.get(dom)  # PMC bytes per step of that kernel
            traffic = int(per_step / kernels[dom]["launches_per_step"]) if per_step else None
            roof = {"kernel": dom, "bound": "hbm", "achieved": kernels[dom]["alg_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": kernels[dom]["hbm_frac"], "traffic": traffic,
                    "traffic_source": "profiles/hbm_traffic.json (builder-side rocprofv3 run, committed)",
                    "note": "per-kernel durations from the sequential events pass; latlon_fused (78 % of the cells) is the HBM-bound "
                            "kernel, the cap kernels are fp64-VALU bound (DESIGN.md 4)"}
            if roof["frac"] < 0.25:   # e.g. the literal displaced-pole quadrature: 24 B per cell behind ~35 k fp64 instructions
                kn = {"dpole_quad": "dpole_quad_kernel<4, 0>", "bipolar_quad": "bipolar_quad_kernel<5, 0>", "bipolar_mesh": "bipolar_mesh_kernel<false>",
                      "dpole_mesh": "dpole_mesh_kernel"}.get(dom)
                vc = valu.get(kn) if kn else None
                rv = None
                if vc:
                    rv = {"kernel": kn, "valu_busy_frac": round(vc["valu_busy_frac"], 4), "wave64_valu_instr_per_launch": vc["wave64_valu_instr"],
                          "source": "profiles/valu_counters.json (builder-side rocprofv3 --pmc run, committed; same kernel sources as the loaded library)"}
                roof = valu_bound_roofline(roof, rv, kernels[dom]["mean_ms"])
        # the number nobody can dispute: every field of the step written once (48 B per cell) over the step's own wall time, per GPU
        roof["frac_step"] = round(48.0 * plan.cells * args.steps / dt / 1e9 / (HBM_PEAK_GBS * world), 4)
        roof["frac_step_note"] = "48 B x cells / ms_per_step / (n_gpus x 8 TB/s): the whole step, launch gaps and the fix-up launch included"
        out = {
            "metric": METRIC, "value": plan.cells * args.steps / dt, "unit": "cells/s",
            "n_gpus": world, "world_size": (dist.get_world_size() if use_dist else 1), "per_rank": per_rank,
            **({"rehearsal": "%d ranks share cuda:0, collectives over gloo (OGG_BENCH_ONE_GPU): not a multi-GPU measurement" % world} if rehearsal else {}), "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3,
            # `value` / `ms_per_step`: K passes of the slowest rank, opening barrier + synchronize to its own synchronize (max over ranks)
            "ms_per_step_slowest_rank": dt_slowest / args.steps * 1e3,
            # the same region through a closing barrier + synchronize (max over ranks): what round 1-3 reported as `ms_per_step`
            "ms_per_step_with_closing_barrier": dt_barrier / args.steps * 1e3,
            "timing": "K plain passes between an opening barrier + synchronize and each rank's own synchronize; `value` from the latest end "
                      "minus the earliest start over the ranks on the node's monotonic clock (a pass holds no inter-rank dependency; the same "
                      "definition at every N); launch events of 2-3 extra passes right after the region",
            "timed_region_clock_ns": region_ns,
            # fused pass: of the K timed passes, how many started with launch B because the previous pass's launch B had built their
            # tables (every pass builds one set of tables; the passes that record events run launch A themselves) -- DESIGN.md 4.1
            "pass_plan": {"workspace_slots": plan_slots, "timed_passes_whose_tables_rode_in_the_previous_launch_b": carried1 - carried0},
            "band_split": plan.split_times,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "1/8 deg tripolar supergrid with metrics (-r 8)" if args.workload == "r8" else args.workload,
                       "flags": flags, "dp_arc": (args.dp_arc if has_dp else None), "cap_symmetry": args.cap_symmetry, "supergrid": [plan.nyp, plan.Ni + 1], "cells": plan.cells,
                       "field_slab": getattr(sg, "field_slab", None),
                       "parallelism": "latitude bands x%d per sub-grid, latlon=%s, halo=%s" % (
                           world, args.latlon, (args.halo if (world > 1 and args.latlon == "stencil") else "none"))},
            "device": _lib.device_name(), "launch": timed_launch, "autotune_ms": tuned, "autotune_strip_workgroups": strips_tuned,
            "ms_per_step_eager_with_events": dt_eager / args.steps * 1e3,
            "aggregate_GBps_at_48B_per_cell": round(48.0 * plan.cells * args.steps / dt / 1e9, 1),
            "roofline": roof, "roofline_valu": roof_valu, "pass_launches": launches,
            "counters": counters_note,
            "parity": _dp_parity(args.dp_arc) if has_dp else None,
            "dp_arc_other": dp_other,
            "self_check_metrics_error_percent": self_check, "field_checksums": checksums,
            "d2h_pinned_after_pass": d2h,
            "kernels": kernels,
            "power": power,
        }
        fp = (power or {}).get("fused_pass")
        if fp and "watts" in fp:
            alone = [v["sclk_mhz"] for k, v in power.items() if k != "fused_pass" and isinstance(v, dict) and "sclk_mhz" in v]
            if alone and fp["sclk_mhz"] < 0.95 * min(alone):
                roof["limited_by"] = ("socket power: the fused pass holds %.0f MHz at %.0f W where either kernel class alone holds >= %.0f MHz "
                                      "(`power`: rocm-smi samples after the timed region; DVFS give-back, DESIGN.md 4)%s"
                                      % (fp["sclk_mhz"], fp["watts"], min(alone),
                                         ("; below that, " + roof["limited_by"]) if roof.get("limited_by") else ""))
        out["cpu_baseline"] = None   # timed on rank 0 of a one-GPU run only (and not with --cpu-sample-div 0)
        if world == 1 and args.cpu_sample_div > 0:
            cells, cdt = cpu_baseline(plan, args.cpu_sample_div)
            out["cpu_baseline"] = {"value": cells / cdt, "unit": "cells/s", "cores": 1, "kind": "port",
                                   "sample": "numpy oracle on the southernmost 1/%d of the rows of every sub-grid of the same "
                                             "workload: %d cells in %.1f s" % (args.cpu_sample_div, cells, cdt),
                                   "host_cpus": os.cpu_count()}
            # the same span with the reference's per-cell Python loops in the quadratures, on a smaller sample (a few seconds)
            rs_div = max(64, args.cpu_sample_div)
            cells2, cdt2 = cpu_baseline(plan, rs_div, per_cell_loop=True)
            out["cpu_baseline_reference_shaped"] = {
                "value": cells2 / cdt2, "unit": "cells/s", "cores": 1, "kind": "port",
                "sample": "as cpu_baseline but the quadratures average cell by cell in Python like the reference (OGG:176-187, 585-599); "
                          "southernmost 1/%d of the rows: %d cells in %.1f s" % (rs_div, cells2, cdt2)}
        print(json.dumps(out))
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
