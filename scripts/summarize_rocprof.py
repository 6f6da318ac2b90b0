#!/usr/bin/env python3
"""Condense rocprofv3 output directories into the small summaries kept under profiles/.

    python scripts/summarize_rocprof.py <round-tag> <workload> <stats_dir> [<fetch_pmc_dir> <write_pmc_dir>]

Writes profiles/<tag>_kernel_stats_<workload>.csv (the --stats table, verbatim), profiles/<tag>_hbm_traffic_<workload>.md
and updates profiles/hbm_traffic.json (bytes per launch per kernel; read by bench.py for roofline.traffic).
HBM bytes follow MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE are in KiB, collected in separate --pmc passes;
on gfx950 FETCH_SIZE tallies 128-B read requests at 64 B, so the read side is doubled.
"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHORT = [("pass_a_kernel", "pass_a"), ("pass_b_kernel", "pass_b"), ("bipolar_quad_tail_kernel", "pass_tail"), ("pass_d_kernel", "pass_dpquad"),
         ("dpole_mesh_kernel", "dpole_mesh"), ("dpole_mesh_reset_kernel", "dpole_mesh"), ("dpole_quad_kernel", "dpole_quad"),
         ("dpole_quad_tables_kernel", "dpole_quad"),
         ("bipolar_quad_kernel", "bipolar_quad"), ("bipolar_tables_kernel", "bipolar_quad"), ("midas_angle_kernel<true", "midas_angle"),
         ("midas_angle_kernel<false", "angle_x"), ("bipolar_mesh_kernel", "bipolar_mesh"), ("tile_latlon_kernel", "tile_latlon"),
         ("dpole_eval_kernel", "dpole_mesh"), ("dpole_unwrap_kernel", "dpole_mesh"), ("dpole_chord", "dpole_quad"), ("dpole_h_kernel", "dpole_quad"), ("dpole_quad_reduce_kernel", "dpole_quad"),
         ("latlon_fused_kernel", "latlon_fused")]


def short(name):
    for pat, s in SHORT:
        if pat in name:
            return s
    return None


def counters(d, counter):
    f = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)[0]
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            s = short(r["Kernel_Name"])
            if s:
                per[(s, int(r["Grid_Size"]))].append(float(r["Counter_Value"]))
    return per


def main():
    tag, workload, stats_dir = sys.argv[1:4]
    out = os.path.join(ROOT, "profiles")
    stats = glob.glob(os.path.join(stats_dir, "**", "*kernel_stats.csv"), recursive=True)[0]
    shutil.copy(stats, os.path.join(out, "%s_kernel_stats_%s.csv" % (tag, workload)))
    if len(sys.argv) < 6:
        return
    fetch, write = counters(sys.argv[4], "FETCH_SIZE"), counters(sys.argv[5], "WRITE_SIZE")
    lines = ["# HBM traffic per launch, %s, workload %s" % (tag, workload), "",
             "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, KiB); read side doubled (gfx950 correction).", "",
             "| kernel | grid size | launches | read MB | written MB | total MB |", "|---|---|---|---|---|---|"]
    table = {}
    for key in sorted(set(fetch) | set(write)):
        rd = 2.0 * 1024 * sum(fetch.get(key, [0])) / max(len(fetch.get(key, [0])), 1)
        wr = 1024.0 * sum(write.get(key, [0])) / max(len(write.get(key, [0])), 1)
        lines.append("| %s | %d | %d | %.1f | %.1f | %.1f |" % (key[0], key[1], len(write.get(key, [])), rd / 1e6, wr / 1e6, (rd + wr) / 1e6))
        t = table.setdefault(key[0], {"launch_bytes": [], "launches": []})
        t["launch_bytes"].append(int(rd + wr))  # one entry per distinct kernel launch of the step
        t["launches"].append(len(write.get(key, [])))
    open(os.path.join(out, "%s_hbm_traffic_%s.md" % (tag, workload)), "w").write("\n".join(lines) + "\n")
    jf = os.path.join(out, "hbm_traffic.json")
    allj = json.load(open(jf)) if os.path.exists(jf) else {}
    # A stand-alone kernel may be launched several times per step with different grids (the three bipolar quadrature kernels): its bytes per
    # step are the sum over its grid sizes.  A launch of the fused pass happens ONCE per pass: a second grid size of the same pass_* kernel
    # in the trace is another plan (the other arc form timed after the region) -- the timed one is the grid size with the most launches,
    # and only that one counts (round 3 summed the two and doubled pass_b for the displaced-pole workloads).
    allj[workload] = {k: (int(v["launch_bytes"][max(range(len(v["launches"])), key=lambda q: v["launches"][q])]) if k.startswith("pass_")
                          else int(sum(v["launch_bytes"]))) for k, v in table.items()}
    import ctypes
    lib = ctypes.CDLL(os.path.join(ROOT, "ocean_model_grid_generator_amd", "csrc", "libogg_hip.so"))
    lib.ogg_version.restype = ctypes.c_char_p
    allj[workload]["_lib_src_hash"] = lib.ogg_version().decode().split(" src ")[-1]   # bench.py quotes these bytes only for this library
    allj["_source"] = ("scripts/summarize_rocprof.py; HBM bytes per STEP of each logical kernel (all its launches; read side x2), "
                       "last updated for " + tag)
    json.dump(allj, open(jf, "w"), indent=1, sort_keys=True)
    print("\n".join(lines))


if __name__ == "__main__":
    main()
