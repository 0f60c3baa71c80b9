"""Development aid: one registration with ICPX_TRACE=1 in the environment prints every
iteration record the host looks at (which kind of pass, what it left over)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401
from iterativeclosestpoint_amd import icpx, synth  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "small"
c = {"c3": synth.config_c3, "c4": synth.config_c4,
     "small": lambda: synth.config_c3(nu=300, nv=150, npts=120_000)}[which]()
metric = int(sys.argv[2]) if len(sys.argv) > 2 else c["metric"]
anchor = int(sys.argv[3]) if len(sys.argv) > 3 else 1
ctx = icpx.Context(0)
ctx.set_target(c["target_xyz"], c["target_tri"])
ctx.set_source(c["source_xyz"], c["source_normals"])
p = icpx.default_params()
p.metric, p.max_iterations, p.max_landmarks = metric, 50, len(c["source_xyz"])
p.max_distance = c["max_dist"]

ctx.set_anchor(anchor)
print(ctx.run(p)[1:], ctx.debug_stats())
