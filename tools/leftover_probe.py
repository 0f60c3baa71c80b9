"""Development aid: after a registration, what the queries that a settled pass still searches
look like (their certificates)."""
import argparse
import os
import sys
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iterativeclosestpoint_amd import icpx, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--workload", default="c5big")
ap.add_argument("--shard", default="0/1")
a = ap.parse_args()
def c6noisy():
    """The graded torus under a cloud with noise of two mean edge lengths."""
    c = synth.config_c6()
    pts, f, fn = synth.sample_surface(c["target_xyz"], c["target_tri"], 1_000_000)
    noise = np.stack([synth.gaussian(synth.SEED, 32 + 2 * k, len(pts)) for k in range(3)], -1) * 0.012
    c.update(source_xyz=synth.apply(c["truth"], pts.astype(np.float64) + noise))
    return c


def c6big():
    """Ten million graded triangles under sixteen million noisy points."""
    c = synth.config_c6(nu=2500, nv=2000, npts=16_000_000)
    pts, f, fn = synth.sample_surface(c["target_xyz"], c["target_tri"], 16_000_000)
    noise = np.stack([synth.gaussian(synth.SEED, 32 + 2 * k, len(pts)) for k in range(3)], -1) * (1e-3 * synth.diag(c["target_xyz"]))
    c.update(source_xyz=synth.apply(c["truth"], pts.astype(np.float64) + noise))
    return c


c = {"c3": synth.config_c3, "c4": synth.config_c4, "c6": synth.config_c6, "c6noisy": c6noisy, "c6big": c6big,
     "c5big": lambda: synth.config_c5(nu=1000, nv=500, npts=1_200_000)}[a.workload]()
r, w = (int(v) for v in a.shard.split("/"))
ctx = icpx.Context(0)
ctx.set_target(c["target_xyz"], c["target_tri"])
ctx.set_source(c["source_xyz"], c["source_normals"])
ctx.set_shard(r, w)
p = icpx.default_params()
p.metric, p.max_iterations, p.max_landmarks, p.max_distance = c["metric"], 50, len(c["source_xyz"]), c["max_dist"]
T, it, mean, st = ctx.run(p)
nq = 0
cnt = ctx.debug_counters()
print("after the run: counters of the last pass", cnt)
n = len(c["source_xyz"])
cert = ctx.debug_certificates(n)
bits = cert.view(np.uint32)
print("certificates: none %d, hopeless mark %d, positive single %d, positive set %d, far %d" % (
    (bits == 0).sum(), (bits == 4).sum(), ((cert > 0) & (bits & 1 == 0) & (bits != 4)).sum(),
    ((cert > 0) & (bits & 1 == 1)).sum(), (cert < 0).sum()))
isset = (cert > 0) & (bits & 1 == 1)
led = isset & (bits & 3 == 3)
lead = ctx.debug_leads(n)
print("sets %d, of which with the winner's lead on record %d (median lead %.2e, 1 %% %.2e); sets without: %d" % (
    isset.sum(), led.sum(), np.median(lead[led]) if led.any() else 0.0,
    np.percentile(lead[led], 1) if led.any() else 0.0, (isset & ~led).sum()))
sets = ctx.debug_contender_sets(n)
size = np.where((sets < 0).any(1), np.argmax(sets < 0, 1), 8)
print("members of the sets without a lead:", np.bincount(size[isset & ~led], minlength=9)[2:], " with:", np.bincount(size[led], minlength=9)[2:])
pos = cert[(cert > 0) & (bits != 4)]
print("positive certificates: percentiles 1 %.2e 10 %.2e 50 %.2e" % tuple(np.percentile(pos, [1, 10, 50])))
far = -cert[cert < 0]
if len(far):
    print("far certificates: percentiles 0.1 %.2e 1 %.2e 10 %.2e 50 %.2e" % tuple(np.percentile(far, [0.1, 1, 10, 50])))
nf = cnt["fallback"]
lst = np.zeros(max(nf, 1), np.uint32)
icpx._chk(icpx.load().icpx_debug_dump(ctx._h, 8, icpx._p(lst), nf))
lc = cert[lst[:nf]]
lb = lc.view(np.uint32)
print("the %d queries on the last pass's list: cert none %d, mark %d, positive %d (median %.2e), far %d (median %.2e)" % (
    nf, (lb == 0).sum(), (lb == 4).sum(), ((lc > 0) & (lb != 4)).sum(),
    np.median(lc[(lc > 0) & (lb != 4)]) if ((lc > 0) & (lb != 4)).any() else 0.0,
    (lc < 0).sum(), np.median(-lc[lc < 0]) if (lc < 0).any() else 0.0))
