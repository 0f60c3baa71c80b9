import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
import torch
from iterativeclosestpoint_amd import icpx, synth
c = synth.config_c3()
ctx = icpx.Context(0)
t0=time.perf_counter(); ctx.set_target(c["target_xyz"], c["target_tri"]); ctx.synchronize(); print("set_target %.1f ms"%(1e3*(time.perf_counter()-t0)))
p = icpx.default_params()
p.metric, p.max_iterations, p.max_landmarks, p.max_distance = 1, 50, len(c["source_xyz"]), 10.0
for i in range(4):
    src = c["source_xyz"] + np.float32(1e-4*i)
    t0=time.perf_counter(); ctx.set_source(src); ctx.synchronize(); t1=time.perf_counter()
    T,it,mean,st = ctx.run(p); t2=time.perf_counter()
    print("set_source %.2f ms, first registration %.2f ms"%(1e3*(t1-t0), 1e3*(t2-t1)))
