"""Development aid: per-pass kernel times of the search on a workload, for a few
characteristic situations (cold pass at the initial pose, hinted pass at the same pose,
hinted pass at the registered pose)."""
import argparse
import sys
import os
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iterativeclosestpoint_amd import icpx, synth  # noqa: E402


def stats_read(ctx):
    """Traversal statistics of a development build (make -C .../csrc stats; ICPX_LIB=...)."""
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
    ap.add_argument("--widths", default="1,4,8")
    ap.add_argument("--leaf", type=int, default=2)
    ap.add_argument("--persistent", type=int, default=1)
    ap.add_argument("--local", type=int, default=1)
    ap.add_argument("--exhaustive", type=int, default=0)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--refill", type=int, default=0)
    a = ap.parse_args()
    c = {"c3": synth.config_c3, "c4": synth.config_c4,
         "c5big": lambda: synth.config_c5(nu=1000, nv=500, npts=1_200_000)}[a.workload]()
    metric = c["metric"]
    Tinv = np.linalg.inv(c["truth"])
    for w in [int(x) for x in a.widths.split(",")]:
        ctx = icpx.Context(0)
        ctx.set_group_width(w)
        ctx.set_exhaustive(a.exhaustive)
        ctx.set_leaf_size(a.leaf)
        ctx.set_persistent(a.persistent)
        ctx.set_local(a.local)
        if a.refill:
            ctx.set_refill(a.refill)
        ctx.set_target(c["target_xyz"], c["target_tri"])
        ctx.set_source(c["source_xyz"], c["source_normals"])
        ctx.accumulate(metric, np.eye(4), 1, c["max_dist"])  # builds tree, sorts queries

        def timed(T, reset):
            ctx.set_local(a.local)  # the shortcut's policy starts afresh for every phase
            ts, fs = [], []
            stats_read(ctx)
            for _ in range(a.reps):
                if reset:
                    ctx.reset_hints()
                ctx.set_timing(True)
                ctx.accumulate(metric, T, 1, c["max_dist"])
                s_ms, s_n, f_ms, f_n = ctx.get_timing()
                ts.append(s_ms)
                fs.append(f_ms)
            st = stats_read(ctx)
            if st is not None:
                nq = len(c["source_xyz"]) * a.reps
                print("   per query: node visits %.1f (wasted %.1f)  leaves %.1f  screens %.1f;"
                      "  per wavefront: loop trips %.0f refills %.1f" %
                      (st[0] / nq, st[2] / nq, st[1] / nq, st[3] / nq, st[4] / (nq / 64),
                       st[5] / (nq / 64)))
                print("   shortcut: let through on certificate %.4f of the queries, newly certified %.4f" %
                      (st[6] / nq, st[7] / nq))
                print("   wavefront walk, per wavefront: node steps %.1f  leaves entered %.1f  past the slab %.1f  "
                      "triangle tests %.1f  screen rounds %.1f + %.1f" %
                      (st[0] / (nq / 64), st[1] / (nq / 64), st[7] / (nq / 64), st[6] / (nq / 64),
                       st[4] / (nq / 64), st[5] / (nq / 64)))
            print("   finalize %.4f ms (min of %d)" % (min(fs), len(fs)))
            return min(ts), float(np.median(ts))
        cold = timed(np.eye(4), True)
        hot0 = timed(np.eye(4), False)
        ctx.accumulate(metric, Tinv, 1, c["max_dist"])
        hot1 = timed(Tinv, False)
        cold1 = timed(Tinv, True)
        print(ctx.debug_stats())
        print("width %2d  cold@start %.3f  hinted@start %.3f  hinted@registered %.3f  "
              "cold@registered %.3f ms (min)  build %.1f ms" %
              (w, cold[0], hot0[0], hot1[0], cold1[0], 1e3 * ctx.build_seconds), flush=True)
        ctx.close()


if __name__ == "__main__":
    main()
