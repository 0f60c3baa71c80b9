"""One-off check on the large configuration (10M triangles, 16M noisy points, point-to-plane):
the GPU registration against the CPU oracle for a few iterations (the oracle needs about a
minute per iteration on 16 threads), then the GPU's own 50 iterations against the ground truth."""
import os
import sys
import time
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402
from iterativeclosestpoint_amd import icpx, synth  # noqa: E402
from oracle import oracle  # noqa: E402

import threading  # noqa: E402


def _heartbeat():
    t0 = time.perf_counter()
    while True:
        time.sleep(30)
        print("  ... %.0f s" % (time.perf_counter() - t0), flush=True)


threading.Thread(target=_heartbeat, daemon=True).start()
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 2
c = synth.config_c4()
n = len(c["source_xyz"])
ctx = icpx.Context(0)
ctx.set_target(c["target_xyz"], c["target_tri"])
ctx.set_source(c["source_xyz"])
p = icpx.default_params()
p.metric, p.max_landmarks, p.max_distance = 1, n, c["max_dist"]
p.max_iterations = iters
t0 = time.perf_counter()
T, it, mean, st = ctx.run(p)
print("gpu %d iterations: %.3f s, mean %.6e" % (it, time.perf_counter() - t0, mean), flush=True)
mesh = oracle.Mesh(c["target_xyz"], c["target_tri"])
t0 = time.perf_counter()
To, ito, mo, sto, no = mesh.icp(c["source_xyz"], metric=1, max_iter=iters, max_landmarks=n,
                                max_dist=c["max_dist"])
print("oracle %d iterations: %.1f s, mean %.6e" % (ito, time.perf_counter() - t0, mo), flush=True)
rel = np.abs(T - To).max() / np.abs(To).max()
print("relative difference of the 4x4 after %d iterations: %.3e" % (iters, rel), flush=True)
p.max_iterations = 50
t0 = time.perf_counter()
T50, it, mean, st = ctx.run(p)
dt = time.perf_counter() - t0
print("gpu 50 iterations: %.3f s (%.3f G correspondences/s), mean %.6e, |T*truth - I|max %.3e" %
      (dt, 50 * n / dt / 1e9, mean, np.abs(T50 @ c["truth"] - np.eye(4)).max()), flush=True)
