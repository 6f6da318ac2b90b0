import sys; sys.path.insert(0, '.')
import torch
from ocean_model_grid_generator_amd import supergrid
plan = supergrid.SupergridPlan(8.0)
sg = supergrid.Supergrid(plan)
for ov in (False, True):
    sg.overlap = ov
    for _ in range(5): sg.step()
    torch.cuda.synchronize()
    sg._events = {}
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): sg.run_pass()
    e1.record(); torch.cuda.synchronize()
    print("overlap", ov, "ms/step %.4f" % (e0.elapsed_time(e1) / 50), {k: round(v["mean_ms"], 4) for k, v in sg.kernel_times_ms().items()})
