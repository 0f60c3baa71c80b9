"""Development aid: per iteration of a host-driven registration, the share of queries holding
a near / far certificate, the typical move of a point, and the search time."""
import os
import sys
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iterativeclosestpoint_amd import icpx, synth  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "c4s"
c = {"c4s": lambda: synth.config_c4(nu=500, nv=400, npts=640_000),
     "c3": synth.config_c3, "c4": synth.config_c4,
     "c5big": lambda: synth.config_c5(nu=1000, nv=500, npts=1_200_000)}[which]()
ctx = icpx.Context(0)
ctx.set_target(c["target_xyz"], c["target_tri"])
ctx.set_source(c["source_xyz"], c["source_normals"])
n = len(c["source_xyz"])
T = np.eye(4)
p = c["source_xyz"][:: max(1, n // 1000)].astype(np.float64)
ctx.set_timing(True)
ps = 0.0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 14):
    sums = ctx.accumulate(c["metric"], T, 1, c["max_dist"])
    M, nn, md, rc = icpx.solve(c["metric"], sums, ctx.origin)
    Tn = M @ T
    mv = np.linalg.norm(p @ (Tn[:3, :3] - T[:3, :3]).T + (Tn[:3, 3] - T[:3, 3]), axis=1)
    cert = ctx.debug_certificates(n)
    s_ms = ctx.get_timing()[0]
    print("it %2d  search %.3f ms  near %.4f  far %.4f  median slack %.2e  next move %.2e  %s %s" % (
        it, s_ms - ps, (cert > 0).mean(), (cert < 0).mean(),
        float(np.median(cert[cert > 0])) if (cert > 0).any() else 0.0, float(np.median(mv)),
        ctx.debug_counters(), ctx.debug_stats()["local_passes"]), flush=True)
    if (cert > 0).any():
        print("       slack quantiles 1%% %.2e  3%% %.2e  10%% %.2e;  no certificate %.4f" % (
            *np.quantile(cert[cert > 0], [0.01, 0.03, 0.1]), (cert == 0).mean()), flush=True)
    ps = s_ms
    T = Tn
