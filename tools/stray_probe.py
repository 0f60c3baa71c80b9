"""What points anywhere in the target's box cost to search (no threshold): C3's torus, n points
uniform in its bounding box, cold and hinted passes under each tree kernel.

    python tools/stray_probe.py [points]
"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from iterativeclosestpoint_amd import icpx, synth as S  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
xyz, tri = S.bumpy_torus(1000, 500)
lo, hi = xyz.min(0).astype(np.float64), xyz.max(0).astype(np.float64)
pts = (np.stack([S.u01(61, k, n) for k in range(3)], -1) * (hi - lo) + lo).astype(np.float32)
for label, knobs in (("by size", {}), ("one wavefront per query", dict(group_width=64)),
                     ("eight lanes per query", dict(group_width=8)),
                     ("pooled lanes", dict(persistent=1, group_width=1, local=0)),
                     ("one walk per wavefront", dict(persistent=3, group_width=1, local=0))):
    ctx = icpx.Context(0)
    for k, v in knobs.items():
        getattr(ctx, "set_" + k)(v)
    ctx.set_target(xyz, tri)
    ctx.set_source(pts)
    ctx.correspond(np.eye(4), max_dist=1e9)
    out = []
    for hinted in (False, True, True):
        if not hinted:
            ctx.reset_hints()
        ctx.synchronize()
        t0 = time.perf_counter()
        got = ctx.correspond(np.eye(4) if not hinted else S.rigid(t=(1e-4, 0, 0)), max_dist=1e9)
        out.append(1e3 * (time.perf_counter() - t0))
    print("%-26s %d strays: cold %.2f ms, hinted %.2f / %.2f ms; mean distance %.3f" %
          (label, n, out[0], out[1], out[2], float(got["dist"].mean())), flush=True)
    ctx.close()
