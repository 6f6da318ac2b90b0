"""Sweep of the device -> NetCDF stream (nc_stream) at 1/8 degree: writer threads x ring slots x slot size, fresh file each time.
usage: python scripts/nc_write_sweep.py [out.json]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ocean_model_grid_generator_amd import supergrid as SG  # noqa: E402

plan = SG.SupergridPlan(8.0)
g = SG.Supergrid(plan, device="cuda:0")
g.run_pass()
torch.cuda.synchronize()
cut = g.south_cut()
res = []
for target, stage in (("/tmp/ogg_sweep.nc", "host"), ("/tmp/ogg_sweep.nc", "device"), ("/dev/shm/ogg_sweep.nc", "device")):
    for threads, slots, mb in ((1, 4, 16), (4, 8, 16), (8, 16, 16), (16, 32, 8), (16, 32, 16), (12, 24, 32)):
        os.environ.update(OGG_NC_THREADS=str(threads), OGG_NC_SLOTS=str(slots), OGG_NC_SLOT_BYTES=str(mb << 20), OGG_NC_STAGE=stage)
        ts = []
        for rep in range(3):
            if os.path.exists(target):
                os.remove(target)
            t = time.perf_counter()
            with open(os.devnull, "w") as devnull, __import__("contextlib").redirect_stdout(devnull):
                nbytes, _ = g.write_nc(target, cut, no_changing_meta=True)
            ts.append(time.perf_counter() - t)
        res.append({"target": target, "stage": stage, "threads": threads, "slots": slots, "slot_MiB": mb, "seconds": ts, "GBps_best": nbytes / min(ts[1:]) / 1e9})
        print(res[-1], flush=True)
    os.remove(target)
# reference point: the same bytes device -> pinned host memory, no file
pinned = torch.empty(1 << 28, dtype=torch.uint8, pin_memory=True)
src = torch.empty(1 << 28, dtype=torch.uint8, device="cuda:0")
for _ in range(2):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(4):
        pinned.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t
res.append({"pinned_d2h_GBps": 4 * (1 << 28) / dt / 1e9})
print(res[-1])
if len(sys.argv) > 1:
    json.dump(res, open(sys.argv[1], "w"), indent=1)
