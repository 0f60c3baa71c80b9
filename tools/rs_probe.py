"""Development aid: where k_reduce_solve's time goes (statistics build: make -C .../csrc stats;
ICPX_LIB=.../libicpx_stats.so).  One registration of the default workload, then the stage
stamps of its last k_reduce_solve."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from iterativeclosestpoint_amd import icpx, synth

c = synth.config_c3()
ctx = icpx.Context(0)
ctx.set_target(c["target_xyz"], c["target_tri"])
ctx.set_source(c["source_xyz"], c["source_normals"])
p = icpx.default_params()
p.metric = int(sys.argv[1]) if len(sys.argv) > 1 else c["metric"]
p.max_landmarks = len(c["source_xyz"])
p.max_distance = c["max_dist"]
p.check_mean_distance = 0
p.max_iterations = 50
for _ in range(2):
    ctx.run(p)
out = (ctypes.c_ulonglong * 37)()
lib = icpx.load()
lib.icpx_stats_read_batches(ctypes.c_void_p(ctx._h.value if hasattr(ctx._h, "value") else ctx._h), out)
