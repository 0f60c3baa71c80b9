#!/usr/bin/env python
"""Benchmark of the ICP hot path on MI355X.

A STEP is one whole registration, i.e. one vtkICP::Update(): the workload's fixed number of
ICP iterations (50, vtkICP's default and the north star's figure) from a cold start (identity
transform, no winners carried over).  Every iteration is a fused closest-point search + fp64
partial sums on the GPU, one all-reduce of the 40-double sum vector (RCCL, stream-ordered,
when more than one rank), solve and transform update on the device.  --steps K times exactly
K such registrations after --warmup W untimed ones; because a step always contains the same
work (one cold pass, a few displaced passes, the converged rest), `value` does not depend on
K.  The K registrations are also timed one by one: median and spread are in `config`.
The search kernels of every fourth of them (--timing-every) carry HIP events, for the
`roofline` object's per-phase kernel times.

Workloads.  c3 (default, the configuration the headline metric names): the synthetic
1,000,000-triangle bumpy torus against 1,000,000 area-uniform surface samples moved by
rotY 5 deg + 1 % of the diagonal, symmetric (plane-to-plane) metric, every point used.
c4: 10,000,000 triangles against 16,000,000 noisy samples, point-to-plane.
Scaling.  weak (default): every rank holds its own scan of the replicated surface.
strong (--scaling strong; what config 4 asks for): every rank loads the SAME cloud and
searches its contiguous share of it (icpx_set_shard); rank 0 then repeats the registration
alone on the whole cloud and requires the same 4x4 to 1e-8 (config.one_rank_check_rel).

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
ITERATIONS = 50        # per registration: vtkICP's default (ICP/vtkICP.cxx:31), the north star's run


def structure_bytes(nv, nt):
    """One read of the search structure: points, indices, BVH nodes at 4 triangles per leaf,
    face / edge / vertex normals."""
    return 12 * nv + 12 * nt + 32 * (2 * nt / 4 - 1) + 12 * nt + 36 * nt + 12 * nv


def make_workload(name, rank, synth, own_scan):
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
    if own_scan and rank > 0:  # weak scaling: every rank scans the same surface with its own samples
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
    ap.add_argument("--steps", type=int, default=20, help="timed registrations")
    ap.add_argument("--warmup", type=int, default=3, help="untimed registrations first")
    ap.add_argument("--workload", default="c3")
    ap.add_argument("--scaling", default="weak", choices=("weak", "strong"))
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
    ap.add_argument("--emulate-shard", default="",
                    help="R/N: this one process runs rank R's share of an N-rank strong-scaling job "
                         "through the multi-rank loop, without a collective (what one rank of N does; "
                         "0/1 = one rank's share of a weak-scaling job)")
    ap.add_argument("--shard-order", type=int, default=-1,
                    help="development: 1 shards by the curve order (default), 0 by blocks of indices")
    ap.add_argument("--certify-min", type=int, default=-1,
                    help="development: queries per rank from which the large-cloud policy applies")
    ap.add_argument("--lean", type=int, default=-1, help="development: icpx_set_lean")
    ap.add_argument("--noisy-policy", type=int, default=-1, help="development: icpx_set_noisy_policy")
    ap.add_argument("--curve", type=int, default=-1, help="development: icpx_set_query_curve")
    ap.add_argument("--dump-passes", action="store_true",
                    help="development: the search kernels' median time of every pass into config")
    ap.add_argument("--timing-every", type=int, default=4,
                    help="the search kernels of every N-th timed registration carry HIP events")
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
    collective = world > 1 or args.force_collective
    if collective:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29541")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world,
                                    device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    strong = args.scaling == "strong"
    emu = None
    if args.emulate_shard:
        emu = tuple(int(v) for v in args.emulate_shard.split("/"))
        if world != 1 or len(emu) != 2 or not 0 <= emu[0] < emu[1]:
            raise SystemExit("--emulate-shard R/N needs one process and 0 <= R < N")
        strong = True
    c = make_workload(args.workload, rank, synth, own_scan=not strong)
    npts, nt, nv = len(c["source_xyz"]), len(c["target_tri"]), len(c["target_xyz"])
    metric = c["metric"]
    # weak: the local scan is this rank's whole shard; strong: a share of the one cloud
    if emu:
        sh = ShardedICP(device=local, rank=emu[0], world=emu[1])
    else:
        sh = ShardedICP(device=local, rank=rank if strong else 0, world=world if strong else 1)
    if args.shard_order >= 0:
        sh.ctx.set_shard_order(args.shard_order)
    if args.certify_min >= 0:
        sh.ctx.set_certify_min(args.certify_min)
    if args.lean >= 0:
        sh.ctx.set_lean(args.lean)
    if args.noisy_policy >= 0:
        sh.ctx.set_noisy_policy(args.noisy_policy)
    if args.curve >= 0:
        sh.ctx.set_query_curve(args.curve)
    if args.width:
        sh.ctx.set_group_width(args.width)
    if args.leaf:
        sh.ctx.set_leaf_size(args.leaf)
    if args.local >= 0:
        sh.ctx.set_local(args.local)
    sh.set_target(c["target_xyz"], c["target_tri"])
    sh.set_source(c["source_xyz"], c["source_normals"])
    build_s = sh.ctx.build_seconds
    from iterativeclosestpoint_amd.distributed import shard_size
    nq_rank = npts if not strong else shard_size(npts, rank, world)
    nq_job = npts * world if not strong else npts
    if emu:  # (with --shard-order 0 or 1 the shard is an equal share to within a point)
        nq_rank = nq_job = shard_size(npts, emu[0], emu[1]) if args.shard_order < 0 or args.shard_order == 2 \
            else npts * (emu[0] + 1) // emu[1] - npts * emu[0] // emu[1]

    params = icpx.default_params()
    params.metric = metric
    params.max_landmarks = npts         # every point is a landmark
    params.max_distance = c["max_dist"]
    params.check_mean_distance = 0      # exactly ITERATIONS iterations
    params.max_iterations = ITERATIONS

    def registration():
        """One vtkICP::Update(): ITERATIONS iterations from a cold start."""
        T, it, md, status = sh.run_device(params, force_collective=args.force_collective,
                                          emulate=emu is not None)
        if it != ITERATIONS:
            raise SystemExit("registration stopped after %d of %d iterations" % (it, ITERATIONS))
        return T, md

    def fence():
        sh.ctx.synchronize()
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        registration()
    # The search kernels of every `timing_every`-th timed registration carry HIP events (the
    # first and the last search kernel of each pass, on the library's stream).  Not of every
    # one: a kernel with a completion event holds the next kernel back by ~5 us, three times
    # per pass, 6 % of a registration of the default workload.
    every = 0 if args.no_kernel_timing else max(1, args.timing_every)
    sh.ctx.set_timing(0)
    fence()
    each, timed, pass_rows, kind_rows = [], [], [], []
    t0 = time.perf_counter()
    for i in range(args.steps):
        with_events = every > 0 and i % every == 0
        if with_events:
            sh.ctx.set_timing(2)       # the search kernels only
        t1 = time.perf_counter()
        T, mean_d = registration()     # (returns when the last iteration's transform is back)
        each.append(time.perf_counter() - t1)
        timed.append(with_events)
        if with_events:
            pm_, kinds_, _ = sh.ctx.get_pass_timings()
            if len(pm_) == ITERATIONS:
                pass_rows.append(pm_)
                kind_rows.append(kinds_)
            sh.ctx.set_timing(0)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64,
                         device="cuda" if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    one_rank_check = None
    if strong and world > 1 and rank == 0:
        # the same registration on one rank with the whole cloud: the sharded sums must
        # lead to the same transform
        one = icpx.Context(local)
        one.set_target(c["target_xyz"], c["target_tri"])
        one.set_source(c["source_xyz"], c["source_normals"])
        T1 = one.run(params)[0]
        one_rank_check = float(np.abs(T - T1).max() / np.abs(T1).max())
        one.close()
        # (the shards' sums add up in another order than one rank's: a few 1e-16 of the sums,
        # which 50 iterations on a noisy cloud amplify; the north star's gate is 1e-6)
        if one_rank_check > 1e-8:
            raise SystemExit("sharded and one-rank transforms differ by %.3e" % one_rank_check)

    if rank == 0:
        total = float(nq_job) * ITERATIONS * args.steps
        each = np.array(each)
        out = {
            "metric": "point-to-mesh correspondences/sec (ICP iterations/sec in config)",
            "value": total / elapsed,
            "unit": "correspondences/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "%s: %d-triangle bumpy torus vs %d source points %s, %s metric, every "
                            "point used; one step = one cold-start registration of %d iterations" % (
                                args.workload, nt, npts,
                                "sharded over the ranks" if strong else "per GPU",
                                ("point-to-point", "point-to-plane", "symmetric")[metric],
                                ITERATIONS),
                "iterations_per_step": ITERATIONS,
                "icp_iterations_per_s": ITERATIONS * args.steps / elapsed,
                "registration_ms_median": float(1e3 * np.median(each)),
                "registration_ms_min": float(1e3 * each.min()),
                "registration_ms_max": float(1e3 * each.max()),
                "registration_ms_median_with_kernel_events":
                    float(1e3 * np.median(each[np.array(timed)])) if any(timed) else None,
                "registrations_with_kernel_events": int(sum(timed)),
                "mean_distance_last_iteration": float(mean_d),
                "residual_vs_truth": float(np.abs(T @ c["truth"] - np.eye(4)).max())
                if "truth" in c else None,
                "target_build_seconds": build_s,
                "world_size": dist.get_world_size() if collective else 1,
                "backend": dist.get_backend() if collective else "none",
                "one_rank_check_rel": one_rank_check,
                "emulated_shard": ("rank %d of %d: this process ran that rank's share of the job (%d points) "
                                   "through the multi-rank loop without a collective; value is that "
                                   "rank's rate" % (emu[0], emu[1], nq_rank)) if emu else None,
                "passes": sh.ctx.debug_stats(),
            },
        }
        if pass_rows:
            # rank 0's search kernels, pass by pass, of the registrations that carried events
            pm = np.array(pass_rows)
            cold = float(np.median(pm[:, 0]))                       # k_search3w, cold start
            steady = float(np.median(pm[:, ITERATIONS // 2:]))      # converged passes
            b_struct = structure_bytes(nv, nt)
            b_cold = 12.0 * nq_rank + b_struct     # every point once + one read of the structure
            b_steady = 20.0 * nq_rank              # point 12 B + certificate 4 B + winner 4 B
            steady_kernels = "k_certify (large clouds) + k_search_local + k_fallback + k_contested"
            # A settled pass of a cloud below four million points is a lean one: the certificate
            # check and the fused rows and sums in ONE kernel (kind 2), whose time is therefore
            # the whole pass but the reduction: its bytes are the check's (point 12 B,
            # certificate 4 B) plus the rows' (point 12 B,
            # winner 4 B, kept index 4 B, the triangle's points 36 B, its {normal, cell} record
            # 16 B, the source normal 12 B where the metric reads it).
            lean = np.array(kind_rows)[:, ITERATIONS // 2:] == 2
            steady_lean = bool(lean.mean() > 0.5)
            if steady_lean:
                steady = float(np.median(pm[:, ITERATIONS // 2:][lean]))
                b_steady = (16.0 + 72.0 + (12.0 if c["source_normals"] is not None else 0.0)) * nq_rank
                steady_kernels = "k_finalize_mx<metric, sums, LEAN> (certificate check + rows + sums in one kernel)"
            ach_cold = b_cold / (cold * 1e-3) / 1e9
            ach_steady = b_steady / (steady * 1e-3) / 1e9
            search_total = float(pm.sum(1).mean())
            if args.dump_passes:
                out["config"]["pass_ms"] = [round(float(v), 4) for v in np.median(pm, 0)]
                out["config"]["pass_kind"] = [int(v) for v in np.array(kind_rows)[0]]
            out["config"]["cold_pass_corr_per_s"] = nq_rank / (cold * 1e-3)
            out["config"]["steady_pass_corr_per_s"] = nq_rank / (steady * 1e-3)
            out["config"]["search_kernels_ms_per_registration"] = search_total
            out["config"]["lean_passes_per_registration"] = float((np.array(kind_rows) == 2).sum(1).mean())
            # The top-level figures are those of the kernel with the largest total per
            # registration (the top row of the rocprofv3 summary of this command under
            # profiles/): the lean kernel's passes together, or the tree passes'; both phases
            # are below either way.
            kinds_a = np.array(kind_rows)
            tot_lean = float((pm * (kinds_a == 2)).sum(1).mean())
            tot_tree = float((pm * (kinds_a == 0)).sum(1).mean())
            lean_top = steady_lean and tot_lean > tot_tree
            out["config"]["lean_kernel_ms_per_registration"] = tot_lean
            out["config"]["tree_pass_kernels_ms_per_registration"] = tot_tree
            out["roofline"] = {
                "bound": "hbm",
                "kernel": ("k_finalize_mx<metric, sums, LEAN> (a settled pass: certificate check + rows + "
                           "sums in one kernel; the largest total share of a registration's GPU time)")
                if lean_top else
                ("k_search3w<4,false> (the cold pass: one walk per wavefront over the "
                 "whole cloud; the largest single share of a registration's GPU time)"),
                "achieved": ach_steady if lean_top else ach_cold,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": (ach_steady if lean_top else ach_cold) / HBM_PEAK_GBS,
                "traffic": None,
                "algorithmic_bytes_per_launch": b_steady if lean_top else b_cold,
                "kernel_ms": steady if lean_top else cold,
                "launches": int((kinds_a == 2).sum()) if lean_top else int(len(pass_rows)),
                "phases": {
                    "cold_pass": {"kernels": "k_search3w + k_contested", "ms": cold,
                                  "algorithmic_bytes": b_cold, "achieved_GBs": ach_cold,
                                  "frac": ach_cold / HBM_PEAK_GBS},
                    "steady_pass": {"kernels": steady_kernels, "lean": steady_lean,
                                    "ms": steady, "algorithmic_bytes": b_steady,
                                    "achieved_GBs": ach_steady,
                                    "frac": ach_steady / HBM_PEAK_GBS},
                },
            }
        else:
            out["roofline"] = {"bound": "hbm", "kernel": "k_search3w", "achieved": None,
                               "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None}
        if not args.no_cpu_baseline and world == 1:
            from oracle import oracle as orc
            mesh = orc.Mesh(c["target_xyz"], c["target_tri"])
            nsample = min(npts, 1_000_000 if nt <= 2_000_000 else 250_000)
            threads = orc.num_threads()
            t0 = time.perf_counter()
            mesh.correspond(c["source_xyz"][:nsample], step=1, max_dist=c["max_dist"],
                            src_normals=None if c["source_normals"] is None
                            else c["source_normals"][:nsample], grid=True, threads=threads)
            dt = time.perf_counter() - t0
            out["cpu_baseline"] = {
                "value": nsample / dt,
                "unit": "correspondences/s",
                "cores": threads,
                "kind": "port",
                "sample": "one correspondence-filter pass at the start pose (what the GPU's cold "
                          "pass does: compare config.cold_pass_corr_per_s) over the first %d source "
                          "points of the same workload, bucket-grid locator restatement of the "
                          "vtkSMPTools path, %.1f s" % (nsample, dt),
            }
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if collective:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
