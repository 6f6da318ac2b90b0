"""Where does the bipolar mesh kernel spend its time?  mesh only / mesh + angle / mesh + scale factors, 1/8 degree cap."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from ocean_model_grid_generator_amd import _lib as L  # noqa: E402

Ni, Nj, lat0 = 5760, 960, 64.03160594077568
n = Nj + 1
mk = lambda c: torch.empty((n, c), dtype=torch.float64, device="cuda")
x, y, a, hi, hj = mk(Ni + 1), mk(Ni + 1), mk(Ni + 1), mk(Ni), mk(Ni + 1)
st = torch.cuda.current_stream().cuda_stream


def t(fn, reps=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print("mesh only            %.1f us" % t(lambda: L.call("ogg_bipolar_cap_mesh_angle_dev", Ni, Nj, lat0, -300.0, 0, n, x.data_ptr(), y.data_ptr(), None, None, None, st)))
print("mesh + angle         %.1f us" % t(lambda: L.call("ogg_bipolar_cap_mesh_angle_dev", Ni, Nj, lat0, -300.0, 0, n, x.data_ptr(), y.data_ptr(), None, None, a.data_ptr(), st)))
print("mesh + h_i, h_j      %.1f us" % t(lambda: L.call("ogg_bipolar_cap_mesh_angle_dev", Ni, Nj, lat0, -300.0, 0, n, x.data_ptr(), y.data_ptr(), hi.data_ptr(), hj.data_ptr(), None, st)))
