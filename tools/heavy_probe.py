"""Development aid (statistics build): which batch of the cold pass walks longest, and what
its 64 queries look like (how far from the surface they start)."""
import ctypes
import os
import sys
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iterativeclosestpoint_amd import icpx, synth  # noqa: E402

c = synth.config_c4()
ctx = icpx.Context(0)
ctx.set_target(c["target_xyz"], c["target_tri"])
ctx.set_source(c["source_xyz"])
lib = icpx.load()
b = (ctypes.c_ulonglong * 37)()
rows = ctx.correspond(np.eye(4), 1, c["max_dist"])
lib.icpx_stats_read_batches(ctypes.c_void_p(ctx._h.value), b)
order = ctx.debug_query_order(len(c["source_xyz"]))
dist = np.zeros(len(order))
dist[rows["src_index"]] = rows["dist"]
edge = 1.6e-3
print("longest batch: %d node steps, starts at query %d; mean batch %.3f ms" % (b[0], b[36], b[2] * 1e-5 / b[3]))
for base in (int(b[36]), 0, 64 * 100000, 64 * 200000):
    k = order[base:base + 64]
    d = dist[k]
    p = c["source_xyz"][k]
    print("batch at %9d: distance to the surface min %.4f mean %.4f max %.4f (edge lengths: %.1f); extent of the batch %.4f; centre %s"
          % (base, d.min(), d.mean(), d.max(), d.mean() / edge, np.linalg.norm(p.max(0) - p.min(0)), p.mean(0).round(3)))
print("all queries: distance mean %.4f, 99 %% %.4f, max %.4f" % (dist.mean(), np.percentile(dist, 99), dist.max()))
