"""Development aid: which queries a settled full pass of icpx_run still hands to the tree
kernel (lean passes off), their certificates, contender sets and the true gaps."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401
from iterativeclosestpoint_amd import icpx, synth  # noqa: E402
from test_gpu_parity import _point_triangle_dist  # noqa: E402

c = synth.config_c3()
ctx = icpx.Context(0)
ctx.set_target(c["target_xyz"], c["target_tri"])
ctx.set_source(c["source_xyz"], c["source_normals"])
n = len(c["source_xyz"])
p = icpx.default_params()
p.metric, p.max_iterations, p.max_landmarks = 2, 30, n
p.max_distance = c["max_dist"]
ctx.set_lean(2 if len(sys.argv) > 1 else 1)
T, it, mean, st = ctx.run(p)
cnt = ctx.debug_counters()
print(cnt, ctx.debug_stats())
fb = np.zeros(max(cnt["fallback"], 1), np.uint32)
icpx._chk(icpx.load().icpx_debug_dump(ctx._h, 8, icpx._p(fb), len(fb)))
cert = ctx.debug_certificates(n)
order = ctx.debug_query_order(n)
sets = ctx.debug_contender_sets(n)
P = c["target_xyz"][c["target_tri"]].astype(np.float64)
x = c["source_xyz"].astype(np.float64) @ T[:3, :3].T + T[:3, 3]
cen = P.mean(1)
for pos in fb[:cnt["fallback"]][:40]:
    xq = x[order[pos]]
    near = np.argsort(((cen - xq) ** 2).sum(1))[:400]
    d = np.sort(_point_triangle_dist(xq[None], P[near])[0])
    print("pos %7d cert %.3e bit %d  d0 %.3e gaps %.3e %.3e %.3e  set %s" % (
        pos, cert[pos], int(cert[pos:pos + 1].view(np.uint32)[0] & 1), d[0], d[1] - d[0],
        d[2] - d[0], d[3] - d[0], sets[pos][:4]))
