#!/usr/bin/env python
"""Benchmark of the ICP hot path on MI355X.

A step is one ICP iteration over the whole source cloud: fused closest-point
search + fp64 partial sums on the GPU, one all-reduce of the 40-double sum vector
(RCCL, stream-ordered, when more than one rank), solve and transform update on the
device.  The timed region is the first K iterations of a registration started cold
(identity transform, no winners carried over), so the expensive first pass is inside it.

Workload (default c3, the configuration the headline metric names): the synthetic
1,000,000-triangle bumpy torus against 1,000,000 area-uniform surface samples moved by
rotY 5 deg + 1 % of the diagonal, symmetric (plane-to-plane) metric, every point used.
With N ranks each rank holds its own 1M-point scan of the same surface (weak scaling);
the target and its BVH are replicated.

Prints ONE JSON line (rank 0).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak


def algorithmic_bytes_per_corr(nv, nt, nq, rows=False):
    """Compulsory traffic per searched point: its 12-byte source point (+ a 36-byte result
    row when rows are written) plus one amortised read of the search structure per pass:
    points, indices, BVH nodes at 4 triangles per leaf, face / edge / vertex normals."""
    structure = 12 * nv + 12 * nt + 32 * (2 * nt / 4 - 1) + 12 * nt + 36 * nt + 12 * nv
    return 12.0 + (36.0 if rows else 0.0) + structure / nq


def make_workload(name, rank, synth):
    if name == "c3":
        c = synth.config_c3()
    elif name == "c4":
        c = synth.config_c4()
    elif name == "c5big":  # partial overlap and 20 % outliers at C3's size (development)
        c = synth.config_c5(nu=1000, nv=500, npts=1_200_000)
    elif name == "c3small":  # for quick functional runs only
        c = synth.config_c3(nu=200, nv=100, npts=50_000)
    else:
        raise SystemExit("unknown workload %s" % name)
    if rank > 0:  # every rank scans the same surface with its own samples
        xyz, tri = c["target_xyz"], c["target_tri"]
        pts, f, fn = synth.sample_surface(xyz, tri, len(c["source_xyz"]), seed=synth.SEED + rank)
        M = c["truth"]
        if name == "c4":
            d = synth.diag(xyz)
            noise = np.stack([synth.gaussian(synth.SEED + rank, 32 + 2 * k, len(pts))
                              for k in range(3)], -1) * (1e-3 * d)
            c["source_xyz"] = synth.apply(M, pts.astype(np.float64) + noise)
        else:
            c["source_xyz"] = synth.apply(M, pts)
            c["source_normals"] = (fn.astype(np.float64) @ M[:3, :3].T).astype(np.float32)
    return c


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="c3")
    ap.add_argument("--width", type=int, default=0, help="lanes per query (0 = library default)")
    ap.add_argument("--leaf", type=int, default=0, help="triangles per leaf (0 = library default)")
    ap.add_argument("--local", type=int, default=-1,
                    help="shortcut mode (icpx_set_local); -1 = library default")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl",
                    help="process-group backend (gloo + --one-device rehearses N ranks on one GPU)")
    ap.add_argument("--one-device", action="store_true",
                    help="rehearsal: every rank uses GPU 0")
    ap.add_argument("--force-collective", action="store_true",
                    help="development: one rank, but through the multi-rank loop with its all-reduce")
    ap.add_argument("--no-kernel-timing", action="store_true",
                    help="development: leave out the HIP events around the search kernels")
    args = ap.parse_args()

    # Only the result line may reach stdout: libraries underneath (RCCL's version banner,
    # gloo's connection notes) write there too, so everything else is sent to stderr.
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from iterativeclosestpoint_amd import icpx, synth
    from iterativeclosestpoint_amd.distributed import ShardedICP

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run"
                         % (args.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    if args.one_device:
        local = 0
    torch.cuda.set_device(local)
    if world > 1 or args.force_collective:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    c = make_workload(args.workload, rank, synth)
    nq, nt, nv = len(c["source_xyz"]), len(c["target_tri"]), len(c["target_xyz"])
    metric = c["metric"]
    sh = ShardedICP(device=local, rank=0, world=1)  # the local scan is this rank's whole shard
    if args.width:
        sh.ctx.set_group_width(args.width)
    if args.leaf:
        sh.ctx.set_leaf_size(args.leaf)
    if args.local >= 0:
        sh.ctx.set_local(args.local)
    sh.set_target(c["target_xyz"], c["target_tri"])
    sh.set_source(c["source_xyz"], c["source_normals"])
    origin = sh.ctx.origin
    build_s = sh.ctx.build_seconds

    params = icpx.default_params()
    params.metric = metric
    params.max_landmarks = nq          # every point is a landmark
    params.max_distance = c["max_dist"]
    params.check_mean_distance = 0     # exactly K iterations

    def registration(iters):
        """iters ICP iterations from a cold start; returns (T, iterations, last mean distance)."""
        if iters <= 0:
            return np.eye(4), 0, 0.0
        params.max_iterations = iters
        T, it, md, status = sh.run_device(params, force_collective=args.force_collective)
        return T, it, md

    def fence():
        sh.ctx.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    registration(args.warmup)
    sh.ctx.set_timing(0 if args.no_kernel_timing else 2)  # the search kernels only
    fence()
    t0 = time.perf_counter()
    T, n_iter, mean_d = registration(args.steps)
    fence()
    elapsed = time.perf_counter() - t0
    s_ms, s_n, f_ms, f_n = sh.ctx.get_timing()
    sh.ctx.set_timing(False)
    if n_iter != args.steps:
        raise SystemExit("registration stopped after %d of %d iterations" % (n_iter, args.steps))
    n_acc = sh.ctx.accumulate(metric, T, 1, c["max_dist"])[0]  # outside the timed region
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64,
                         device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        total = float(nq) * world * args.steps
        b_alg = algorithmic_bytes_per_corr(nv, nt, nq)
        search_ms = s_ms / max(s_n, 1)
        achieved = b_alg * nq / (search_ms * 1e-3) / 1e9 if s_n else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_%s.json" % args.workload)
        if os.path.exists(tpath):
            with open(tpath) as fh:
                traffic = json.load(fh).get("k_search_hbm_bytes_per_launch")
        out = {
            "metric": "point-to-mesh correspondences/sec (ICP iterations/sec in config)",
            "value": total / elapsed,
            "unit": "correspondences/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "%s: %d-triangle bumpy torus vs %d source points per GPU, %s metric, "
                            "every point used, cold-start registration" % (
                                args.workload, nt, nq,
                                ("point-to-point", "point-to-plane", "symmetric")[metric]),
                "icp_iterations_per_s": args.steps / elapsed,
                "accepted_last_iteration": int(n_acc),
                "mean_distance_last_iteration": float(mean_d),
                "residual_vs_truth": float(np.abs(T @ c["truth"] - np.eye(4)).max())
                if "truth" in c else None,
                "target_build_seconds": build_s,
                "finalize_kernel_ms_avg": (f_ms / f_n) if f_n else None,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "k_search",
                "kernel_note": "one launch = the search kernels of one pass on the library's "
                               "stream (k_certify for large clouds, k_search_local or k_search3w, "
                               "then k_fallback / k_search3p and k_contested), HIP events around them",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "algorithmic_bytes_per_correspondence": b_alg,
                "kernel_ms_avg": search_ms,
                "launches": int(s_n),
            },
        }
        if not args.no_cpu_baseline and world == 1:
            from oracle import oracle as orc
            mesh = orc.Mesh(c["target_xyz"], c["target_tri"])
            nsample = min(nq, 1_000_000)
            threads = orc.num_threads()
            t0 = time.perf_counter()
            rows = mesh.correspond(c["source_xyz"][:nsample], step=1, max_dist=c["max_dist"],
                                   src_normals=None if c["source_normals"] is None
                                   else c["source_normals"][:nsample], grid=True, threads=threads)
            dt = time.perf_counter() - t0
            out["cpu_baseline"] = {
                "value": nsample / dt,
                "unit": "correspondences/s",
                "cores": threads,
                "kind": "port",
                "sample": "one correspondence-filter pass (identity transform) over the first %d "
                          "source points of the same workload, bucket-grid locator restatement of "
                          "the vtkSMPTools path, %.1f s" % (nsample, dt),
            }
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if world > 1 or args.force_collective:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
