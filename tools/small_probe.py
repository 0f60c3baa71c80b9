"""Development aid: the reference's default use (200 landmarks, 50 iterations) timed per registration."""
import os
import sys
import time
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402
from iterativeclosestpoint_amd import icpx, synth  # noqa: E402

c = synth.config_c3(nu=200, nv=100, npts=50_000)
for landmarks in (200, 2000, 20000):
    ctx = icpx.Context(0)
    ctx.set_target(c["target_xyz"], c["target_tri"])
    ctx.set_source(c["source_xyz"], c["source_normals"])
    p = icpx.default_params()
    p.metric, p.max_iterations, p.max_landmarks, p.max_distance = 1, 50, landmarks, c["max_dist"]
    for _ in range(3):
        ctx.run(p)
    t = []
    for _ in range(20):
        t0 = time.perf_counter()
        T, it, mean, st = ctx.run(p)
        t.append(1e3 * (time.perf_counter() - t0))
    print("landmarks %d: registration of %d iterations %.3f ms median (%.1f us per iteration), mean distance %.2e"
          % (landmarks, it, np.median(t), 1e3 * np.median(t) / max(it, 1), mean))
