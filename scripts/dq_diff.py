"""Difference of the displaced-pole quadrature between two builds of the library on the same cap (an experiment tool):
    python scripts/dq_diff.py ab/libogg_hip_a.so ab/libogg_hip_b.so [Ni Nj]
Each build runs in its own process (the library path is fixed at import) and saves dxq, dyq, daq; then max |a - b| and where."""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import contextlib
    import io
    from ocean_model_grid_generator_amd import ocean_grid_generator as ogg
    ni, nj = int(sys.argv[3]), int(sys.argv[4])
    with contextlib.redirect_stdout(io.StringIO()):
        dx, dy, da = ogg.displacedPoleCap_metrics_quad(4, ni, nj, -300.0, -78.0, 80.0, 0.2)
    np.savez(sys.argv[2], dx=dx, dy=dy, da=da)
    sys.exit(0)
a, b = sys.argv[1], sys.argv[2]
ni, nj = (sys.argv[3], sys.argv[4]) if len(sys.argv) > 4 else ("720", "70")
outs = []
for k, lib in enumerate((a, b)):
    out = "/tmp/dq_diff_%d.npz" % k
    subprocess.run([sys.executable, os.path.abspath(__file__), "--child", out, ni, nj], env=dict(os.environ, OGG_LIB_PATH=os.path.abspath(lib)), check=True)
    outs.append(np.load(out))
for f in ("dx", "dy", "da"):
    d = np.abs(outs[0][f] - outs[1][f])
    nz = np.argwhere(d > 0)
    print(f, "max |diff| %.3e rel %.3e, differing %d of %d" % (d.max(), (d / np.abs(outs[0][f]).max()).max(), nz.shape[0], d.size),
          "rows", sorted(set(nz[:, 0]))[:12], "cols", sorted(set(nz[:, 1]))[:12] if nz.size else "")
