"""Development aid: per-iteration search / finalize kernel times and wall time of a
cold-start registration on a workload."""
import argparse
import os
import sys
import time
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iterativeclosestpoint_amd import icpx, synth  # noqa: E402


def stats_read(ctx):
    """Traversal counters of a development build (make -C .../csrc stats; ICPX_LIB=...)."""
    import ctypes
    lib = icpx.load()
    if not hasattr(lib, "icpx_stats_read"):
        return None
    out = (ctypes.c_ulonglong * 8)()
    lib.icpx_stats_read(ctypes.c_void_p(ctx._h.value if hasattr(ctx._h, "value") else ctx._h), out)
    return list(out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c3")
    ap.add_argument("--width", type=int, default=0)
    ap.add_argument("--leaf", type=int, default=0)
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--refill", type=int, default=0)
    ap.add_argument("--local", type=int, default=1)
    ap.add_argument("--persistent", type=int, default=-1)
    ap.add_argument("--certs", action="store_true", help="summarise the certificates after every pass")
    a = ap.parse_args()
    c = {"c3": synth.config_c3, "c4": synth.config_c4,
         "c5big": lambda: synth.config_c5(nu=1000, nv=500, npts=1_200_000)}[a.workload]()
    metric = c["metric"]
    ctx = icpx.Context(0)
    if a.width:
        ctx.set_group_width(a.width)
    if a.leaf:
        ctx.set_leaf_size(a.leaf)
    if a.refill:
        ctx.set_refill(a.refill)
    if a.persistent >= 0:
        ctx.set_persistent(a.persistent)
    ctx.set_local(a.local)
    ctx.set_target(c["target_xyz"], c["target_tri"])
    ctx.set_source(c["source_xyz"], c["source_normals"])
    lo, hi = c["source_xyz"].min(0), c["source_xyz"].max(0)
    corners = np.array([[x, y, z, 1.0] for x in (lo[0], hi[0]) for y in (lo[1], hi[1])
                        for z in (lo[2], hi[2])])
    for rep in range(2):
        ctx.reset_hints()
        T = np.eye(4)
        ctx.set_timing(True)
        ps = pf = 0.0
        for it in range(a.iters):
            t0 = time.perf_counter()
            sums = ctx.accumulate(metric, T, 1, c["max_dist"])
            M, n, md, rc = icpx.solve(metric, sums, ctx.origin)
            Tn = M @ T
            # how far this step moves the cloud (largest displacement of a corner of its box)
            step_move = np.linalg.norm((corners @ (Tn - T).T)[:, :3], axis=1).max()
            T = Tn
            wall = 1e3 * (time.perf_counter() - t0)
            s_ms, s_n, f_ms, f_n = ctx.get_timing()
            if rep == 1:
                print("it %2d  wall %.3f  search %.3f  finalize %.3f  n %d  mean %.3e  step moves the "
                      "cloud by %.2e  %s" %
                      (it, wall, s_ms - ps, f_ms - pf, n, md, step_move, ctx.debug_counters()), flush=True)
            ps, pf = s_ms, f_ms
            if a.certs and rep == 1:
                cert = ctx.debug_certificates(len(c["source_xyz"]))
                isset = (cert > 0) & ((cert.view(np.uint32) & 1) == 1)
                print("       certificates after the pass: single %.4f  set %.4f  far %.4f  none %.4f;"
                      "  median single %.2e  median set %.2e" %
                      (((cert > 0) & ~isset).mean(), isset.mean(), (cert < 0).mean(), (cert == 0).mean(),
                       np.median(cert[(cert > 0) & ~isset]) if ((cert > 0) & ~isset).any() else 0.0,
                       np.median(cert[isset]) if isset.any() else 0.0), flush=True)
            st = stats_read(ctx)
            if st is not None and rep == 1:
                nw = len(c["source_xyz"]) / 64.0
                print("       per wavefront of a whole k_search3w pass: node steps %.1f (nobody "
                      "interested %.1f)  leaves %.1f  screen rounds %.1f + %.1f;  screens per query %.2f"
                      % (st[0] / nw, st[2] / nw, st[1] / nw, st[4] / nw, st[5] / nw,
                         st[3] / len(c["source_xyz"])), flush=True)
                print("       triangles met %.1f, of which some lane took the plane-and-circle bound "
                      "%.1f" % (st[6] / nw, st[7] / nw), flush=True)
        ctx.set_timing(False)


if __name__ == "__main__":
    main()
