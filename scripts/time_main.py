import sys, time; sys.path.insert(0, '.')
import io, contextlib
import ocean_model_grid_generator_amd.ocean_grid_generator as ogg
for r in (8.0,):
    for rep in range(2):
        t = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            g = ogg.main(r, gridfilename=None, no_changing_meta=True, return_arrays=True)
        dt = time.perf_counter() - t
        print("main(-r %g) host arrays, no file: %.3f s  -> %.3e cells/s (PCIe + host stitching inclusive)" % (r, dt, g["area"].size / dt))
t = time.perf_counter()
with contextlib.redirect_stdout(io.StringIO()):
    ogg.main(8.0, gridfilename="/tmp/ocean_hgrid_r8.nc", no_changing_meta=True)
print("main(-r 8) incl. 1.2 GB NetCDF write to /tmp: %.3f s" % (time.perf_counter() - t))
