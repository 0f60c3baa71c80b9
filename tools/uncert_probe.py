"""Development aid: which queries of a converged C3 registration hold no certificate, and how
close their nearest triangles are to one another (brute force over all triangles)."""
import os
import sys
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from iterativeclosestpoint_amd import icpx, synth  # noqa: E402
from test_gpu_parity import _point_triangle_dist  # noqa: E402

c = synth.config_c3()
ctx = icpx.Context(0)
ctx.set_target(c["target_xyz"], c["target_tri"])
ctx.set_source(c["source_xyz"], c["source_normals"])
n = len(c["source_xyz"])
T = np.eye(4)
for it in range(10):
    sums = ctx.accumulate(c["metric"], T, 1, c["max_dist"])
    M, nn, md, rc = icpx.solve(c["metric"], sums, ctx.origin)
    if it < 9:
        T = M @ T
cert = ctx.debug_certificates(n)
order = ctx.debug_query_order(n)
sets = ctx.debug_contender_sets(n)
bad = np.nonzero(cert == 0)[0]
print("no certificate:", len(bad), " sets:", int(((cert > 0) & ((cert.view(np.uint32) & 1) == 1)).sum()))
P = c["target_xyz"][c["target_tri"]].astype(np.float64)
x = c["source_xyz"].astype(np.float64) @ T[:3, :3].T + T[:3, 3]
cen = P.mean(1)
for pos in bad[:24]:
    xq = x[order[pos]]
    near = np.argsort(((cen - xq) ** 2).sum(1))[:400]
    d = np.sort(_point_triangle_dist(xq[None], P[near])[0])
    print("pos %7d  d0 %.3e  gaps %.3e %.3e %.3e %.3e  set %s" % (
        pos, d[0], d[1] - d[0], d[2] - d[0], d[3] - d[0], d[4] - d[0], sets[pos][:4]))
