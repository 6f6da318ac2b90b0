"""Cycle breakdown of the literal displaced-pole quadrature's strip walk (needs an experiment build of the library):
    scripts/ab_build.sh WORK prof -DOGG_DQ_PROFILE=1
    OGG_LIB_PATH=$PWD/ab/libogg_hip_prof.so python scripts/dq_profile_run.py r8_latdp r4_om4
"""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch  # noqa: E402

import bench  # noqa: E402
from ocean_model_grid_generator_amd import supergrid  # noqa: E402

for wl in sys.argv[1:]:
    plan = supergrid.SupergridPlan(dp_arc="literal", **bench.WORKLOADS[wl])
    sg = supergrid.Supergrid(plan, device="cuda:0")
    sg.launch = "pass"
    os.environ.pop("OGG_DQ_PROFILE", None)
    for _ in range(30):
        sg.run_pass()
    torch.cuda.synchronize()
    os.environ["OGG_DQ_PROFILE"] = "1"
    for _ in range(3):
        print("==", wl, file=sys.stderr, flush=True)
        sg.run_pass()
    os.environ.pop("OGG_DQ_PROFILE", None)
    del sg
