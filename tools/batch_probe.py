"""Development aid: how the wavefront walk's batches spread on a workload's cold pass (statistics
build: make -C .../csrc stats; ICPX_LIB=.../libicpx_stats.so): node steps of the longest batch,
its duration, and the batches and their time by node steps / 128."""
import argparse
import ctypes
import os
import sys
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iterativeclosestpoint_amd import icpx, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="c6")
ap.add_argument("--leaf", type=int, default=0)
a = ap.parse_args()
c = {"c3": synth.config_c3, "c4": synth.config_c4, "c6": synth.config_c6,
     "c5big": lambda: synth.config_c5(nu=1000, nv=500, npts=1_200_000)}[a.workload]()
ctx = icpx.Context(0)
if a.leaf:
    ctx.set_leaf_size(a.leaf)
ctx.set_target(c["target_xyz"], c["target_tri"])
ctx.set_source(c["source_xyz"], c["source_normals"])
lib = icpx.load()
h = ctypes.c_void_p(ctx._h.value if hasattr(ctx._h, "value") else ctx._h)
ctx.accumulate(c["metric"], np.eye(4), 1, c["max_dist"])
for label, T, reset in (("cold at the start pose", np.eye(4), True), ("hinted at the start pose", np.eye(4), False)):
    st = (ctypes.c_ulonglong * 8)()
    lib.icpx_stats_read(h, st)
    out = (ctypes.c_ulonglong * 37)()
    lib.icpx_stats_read_batches(h, out)
    if reset:
        ctx.reset_hints()
    ctx.set_timing(True)
    ctx.accumulate(c["metric"], T, 1, c["max_dist"])
    s_ms = ctx.get_timing()[0]
    lib.icpx_stats_read(h, st)
    lib.icpx_stats_read_batches(h, out)
    o = list(out)
    nb = max(o[3], 1)
    print("%s: search %.3f ms; batches %d; node steps per batch %.1f mean, %d longest (starts at query %d); "
          "ticks (10 ns) per batch %.0f mean, %d longest" % (label, s_ms, o[3], st[0] / nb, o[0], o[36], o[2] / nb, o[1]))
    print("   leaves entered %.1f, past the slab %.1f, triangle tests %.1f, screen rounds %.1f + %.1f per batch" %
          (st[1] / nb, st[7] / nb, st[6] / nb, st[4] / nb, st[5] / nb))
    print("   batches by node steps / 128:", o[4:20])
    print("   their ticks (us):           ", [round(v / 100) for v in o[20:36]])
