"""Development aid: the cold pass (and the hinted pass at the same pose) of one rank's shard
of a workload, for several numbers of ranks; with a statistics build (make -C .../csrc stats;
ICPX_LIB=...) also what a wavefront's walk does."""
import argparse
import os
import sys
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iterativeclosestpoint_amd import icpx, synth  # noqa: E402
from tools.time_passes import stats_read  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="c4")
    ap.add_argument("--worlds", default="1,2,4,8")
    ap.add_argument("--order", type=int, default=2)
    ap.add_argument("--wave-blocks", default="19")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--curve", type=int, default=1)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    c = {"c3": synth.config_c3, "c4": synth.config_c4}[a.workload]()
    metric = c["metric"]
    ctx = icpx.Context(0)
    ctx.set_target(c["target_xyz"], c["target_tri"])
    ctx.set_source(c["source_xyz"], c["source_normals"])
    ctx.set_shard_order(a.order)
    ctx.set_query_curve(a.curve)
    ctx.set_seed(a.seed)
    for world in [int(v) for v in a.worlds.split(",")]:
        for rank in sorted({0, world // 2}):
            ctx.set_shard(rank, world)
            for wb in [int(v) for v in a.wave_blocks.split(",")]:
                ctx.set_wave_blocks(wb)
                ctx.accumulate(metric, np.eye(4), 1, c["max_dist"])
                stats_read(ctx)
                cold, hot = [], []
                for _ in range(a.reps):
                    ctx.reset_hints()
                    ctx.set_timing(True)
                    sums = ctx.accumulate(metric, np.eye(4), 1, c["max_dist"])
                    cold.append(ctx.get_timing()[0])
                st = stats_read(ctx)
                for _ in range(a.reps):
                    ctx.set_timing(True)
                    ctx.accumulate(metric, np.eye(4), 1, c["max_dist"])
                    hot.append(ctx.get_timing()[0])
                nq = max(int(sums[0]), 1)
                msg = "world %2d rank %2d blocks/CU %2d: cold %.3f ms (%.2f ns/query)  hinted %.3f ms  n %d" % (
                    world, rank, wb, min(cold), 1e6 * min(cold) / nq, min(hot), int(sums[0]))
                if st is not None:
                    nw = nq * a.reps / 64.0
                    msg += "  | per wavefront: node steps %.1f leaves %.1f past the slab %.1f triangle tests %.1f" % (
                        st[0] / nw, st[1] / nw, st[7] / nw, st[6] / nw)
                    lib = icpx.load()
                    if hasattr(lib, "icpx_stats_read_batches"):
                        import ctypes
                        b = (ctypes.c_ulonglong * 36)()
                        lib.icpx_stats_read_batches(ctypes.c_void_p(ctx._h.value), b)
                        if b[3]:
                            msg += "\n      batches (cold and hinted passes together): %d, node steps of the longest %d, " \
                                   "longest %.3f ms, mean %.3f ms" % (b[3], b[0], b[1] * 1e-5, b[2] * 1e-5 / b[3])
                            msg += "\n      batches by node steps / 128: " + " ".join(str(v) for v in b[4:20])
                            msg += "\n      ... their share of all batches' time: " + " ".join(
                                "%.3f" % (v / b[2]) for v in b[20:36])
                print(msg, flush=True)


if __name__ == "__main__":
    main()
