"""A/B timing of two (or more) builds of libogg_hip.so on ONE box: every build runs `bench.py --workload W` in its own process,
interleaved `--rounds` times, so that box-to-box differences (5-8 % between two MI355X boxes for the same binary) cancel.

usage: python scripts/ab_time.py --libs ab/libogg_hip_prev.so ocean_model_grid_generator_amd/csrc/libogg_hip.so [--workloads r8 ...]
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ap = argparse.ArgumentParser()
ap.add_argument("--libs", nargs="+", required=True)
ap.add_argument("--workloads", nargs="+", default=["r8"])
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--steps", type=int, default=200)
ap.add_argument("--extra", default="", help="more bench.py arguments, one string (e.g. --extra=\"--dp-arc literal\")")
ap.add_argument("--json", default=None)
args = ap.parse_args()
res = {}
for rnd in range(args.rounds):
    for wl in args.workloads:
        for lib in args.libs:
            env = dict(os.environ, OGG_LIB_PATH=os.path.abspath(lib))
            out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", wl, "--steps", str(args.steps), "--cpu-sample-div", "0", "--d2h", "0", "--checksum", "0", "--power-probe", "0", "--tune-strips", "0"] + args.extra.split(),
                                 env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
            if out.returncode:
                print(out.stderr[-2000:], flush=True)
                raise SystemExit(1)
            d = json.loads(out.stdout.strip().splitlines()[-1])
            res.setdefault((wl, lib), []).append((d["ms_per_step"], (d["pass_launches"] or {}).get("pass_b", {}).get("ms")))
            kern = {k: round(v["mean_ms"], 4) for k, v in (d.get("kernels") or {}).items()}
            res.setdefault((wl, lib, "kernels"), []).append(kern)
            pl = d["pass_launches"] or {}
            cs = d.get("field_checksums") or {}
            sig = {k: "".join(v[f][-4:] for f in ("x", "y", "dx", "dy", "area", "angle_dx")) for k, v in cs.items() if isinstance(v, dict)}   # bit fingerprints, shortened
            print(rnd, wl, os.path.basename(lib), "%.4f" % d["ms_per_step"], {k: round(v["ms"], 4) for k, v in pl.items()}, kern, sig, flush=True)
summary = {"%s|%s" % k: {"ms_per_step": [a for a, _ in v], "pass_b_ms": [b for _, b in v], "best_ms_per_step": min(a for a, _ in v),
                             "stand_alone_kernels_ms": res[k + ("kernels",)]} for k, v in res.items() if len(k) == 2}
print(json.dumps(summary))
if args.json:
    json.dump(summary, open(args.json, "w"), indent=1)
