"""First start / last end of every role of launch B (needs an experiment build of the library):
    scripts/ab_build.sh WORK tl -DOGG_PASS_TIMELINE=1
    OGG_LIB_PATH=$PWD/ab/libogg_hip_tl.so python scripts/timeline_run.py r8 r16 r8_latdp
"""
import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, bench
from ocean_model_grid_generator_amd import supergrid
# OGG_TL_RANK / OGG_TL_WORLD: one rank's share of a band split instead of the whole grid
rank, world = int(os.environ.get("OGG_TL_RANK", "0")), int(os.environ.get("OGG_TL_WORLD", "1"))
for wl in sys.argv[1:]:
    plan = supergrid.SupergridPlan(dp_arc=os.environ.get("OGG_TL_ARC", "literal"), **bench.WORKLOADS[wl])
    sg = supergrid.Supergrid(plan, rank=rank, world=world, device="cuda:0", halo="recompute")
    sg.launch = "pass"
    os.environ.pop("OGG_TIMELINE", None)
    for _ in range(100): sg.run_pass()
    torch.cuda.synchronize()
    os.environ["OGG_TIMELINE"] = "1"
    for _ in range(3):
        print("==", wl, file=sys.stderr, flush=True)
        sg.run_pass()
    os.environ.pop("OGG_TIMELINE", None)
    del sg
