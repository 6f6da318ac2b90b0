"""Wall time of the drop-in main() at 1/8 degree on the GPU box: the device-resident pass path (default) against the
function-level path, with and without the 1.2 GB NetCDF file.  usage: python scripts/time_main.py [out.json]"""
import contextlib
import io
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ocean_model_grid_generator_amd.ocean_grid_generator as ogg  # noqa: E402

out = {}


def timed(label, reps=3, **kw):
    ts = []
    for _ in range(reps):
        t = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            ogg.main(8.0, no_changing_meta=True, **kw)
        ts.append(time.perf_counter() - t)
    out[label] = {"seconds_per_call": ts, "best": min(ts)}
    print(label, ["%.4f" % t for t in ts], flush=True)


timed("pass_path_no_file", gridfilename=None)
timed("pass_path_file_tmp", gridfilename="/tmp/ocean_hgrid_r8.nc")
out["file_bytes"] = os.path.getsize("/tmp/ocean_hgrid_r8.nc")
out["pass_path_file_GBps_best"] = out["file_bytes"] / out["pass_path_file_tmp"]["best"] / 1e9
timed("function_level_no_file", reps=2, gridfilename=None, path="functions")
timed("function_level_file_tmp", reps=2, gridfilename="/tmp/ocean_hgrid_r8_f.nc", path="functions")
same = open("/tmp/ocean_hgrid_r8.nc", "rb").read(1 << 20) == open("/tmp/ocean_hgrid_r8_f.nc", "rb").read(1 << 20)
out["first_MiB_of_the_two_files_equal"] = bool(same)
print(json.dumps(out))
if len(sys.argv) > 1:
    json.dump(out, open(sys.argv[1], "w"), indent=1)
