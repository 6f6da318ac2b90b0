"""Sample the GPU's clocks and power (rocm-smi) while one kernel mix runs in a loop: is the pass power-limited?

usage: python scripts/clock_probe.py [--workload r8] [--seconds 6]
Runs, one after the other: idle, the stand-alone lat-lon kernel, the stand-alone cap kernels, the fused pass.
"""
import argparse
import json
import os
import subprocess
import sys
import threading
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import bench  # noqa: E402
from ocean_model_grid_generator_amd import supergrid  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="r8")
ap.add_argument("--seconds", type=float, default=6.0)
ap.add_argument("--json", default=None)
args = ap.parse_args()


def smi():
    out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp", "--json"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True).stdout
    try:
        d = json.loads(out)
        card = d[sorted(d)[0]]
        return {k: v for k, v in card.items() if any(t in k.lower() for t in ("sclk", "mclk", "fclk", "power", "temperature (sensor junction", "temperature (sensor memory", "socclk"))}
    except Exception as e:  # noqa: BLE001
        return {"error": str(e), "raw": out[:300]}


plan = supergrid.SupergridPlan(**bench.WORKLOADS[args.workload])
sg = supergrid.Supergrid(plan, rank=0, world=1, device="cuda:0", halo="recompute")
sg.launch = "pass"
sg.step()
torch.cuda.synchronize()
results = {}


def phase(name, fn):
    samples = []
    stop = threading.Event()

    def sampler():
        while not stop.is_set():
            samples.append(smi())
            time.sleep(0.5)

    th = threading.Thread(target=sampler)
    th.start()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < args.seconds:
        if fn is None:
            time.sleep(0.05)
        else:
            for _ in range(50):
                fn()
            torch.cuda.synchronize()
            n += 50
    dt = time.perf_counter() - t0
    stop.set()
    th.join()
    results[name] = {"ms_per_call": (dt / n * 1e3) if n else None, "samples": samples[1:]}
    print(name, "ms/call", results[name]["ms_per_call"], flush=True)
    for s in samples[1::3]:
        print("   ", s, flush=True)


phase("idle", None)
phase("pass", sg.run_pass)
sg.launch, sg.overlap = "kernels", False
caps = [x.name for x in plan.subs if x.kind in ("bipolar", "dpole")]


def latlon_only():
    sg.phase_a(kinds=("mercator", "latlon"))
    sg.phase_b(kinds=("mercator", "latlon"))


def caps_only():
    for c in caps:
        sg.phase_a(only=c)
        sg.phase_b(only=c)


phase("latlon_kernel_alone", latlon_only)
phase("cap_kernels_alone", caps_only)
sg.launch = "pass"
phase("pass_again", sg.run_pass)
if args.json:
    json.dump(results, open(args.json, "w"), indent=1)
