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
`roofline` object's per-kernel times.

Workloads.  c3 (default, the configuration the headline metric names): the synthetic
1,000,000-triangle bumpy torus against 1,000,000 area-uniform surface samples moved by
rotY 5 deg + 1 % of the diagonal, symmetric (plane-to-plane) metric, every point used.
c4: 10,000,000 triangles against 16,000,000 noisy samples, point-to-plane (the north star's
target run).  The default one-GPU run measures c3 (the top-level figures) and then c4 with 1
warm-up + 3 registrations, reported under config.c4 with its own roofline figures, and c6
briefly (config.c6).
c5big: partial overlap with 20 % outliers at c3's size; c6: c3's size on a graded mesh (edge
lengths over a range of 65, squares to slivers), point-to-plane.
Scaling.  weak (default): every rank holds its own scan of the replicated surface.
strong (--scaling strong; what config 4 asks for): every rank loads the SAME cloud and
searches its share of it (icpx_set_shard); rank 0 then repeats the registration alone on the
whole cloud and requires the same 4x4 to 1e-8 (config.one_rank_check_rel).
--emulate-shard R/N: this one process runs what rank R of an N-rank strong-scaling job runs
(its shard, the multi-rank loop without the collective): the only multi-GPU evidence a
one-GPU box can give.

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
    elif name == "c5big":  # partial overlap and 20 % outliers at C3's size
        c = synth.config_c5(nu=1000, nv=500, npts=1_200_000)
    elif name == "c6":  # a graded mesh at C3's size (edge lengths over a range of 65)
        c = synth.config_c6()
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


def measured_traffic(workload):
    """HBM-side bytes per launch of the two top kernels from the PMC passes kept under
    profiles/ (tools/traffic.sh: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE runs of this
    command, each by itself, FETCH_SIZE doubled as the gfx950 correction for wide reads
    prescribes).  Counters cannot be collected inside the timed run, so the figure is that
    earlier run's; None when no such file is there."""
    path = os.path.join(ROOT, "profiles", "r03_%s_traffic.json" % workload)
    try:
        t = json.load(open(path))
    except (OSError, ValueError):
        return None, None
    out = {}
    for k, v in t.items():
        flat = k.replace(" ", "")
        key = "k_search3w_cold" if flat.startswith("k_search3w<4,false") else \
            "k_finalize_mx_lean" if (flat.startswith("k_finalize_mx") and flat.endswith("true>")) else None
        if key and key not in out:
            out[key] = {"fetch_max": v["fetch_bytes_max_x2"], "fetch_mean": v["fetch_bytes_mean_x2"],
                        "write_mean": v["write_bytes_mean"]}
    return out, os.path.relpath(path, ROOT)


def run_workload(args, name, steps, warmup, strong, emu, rank, world, local, collective, modules):
    """Times `steps` registrations of one workload; returns what the JSON line is made from."""
    torch, dist, icpx, synth, ShardedICP, shard_size = modules
    c = make_workload(name, rank, synth, own_scan=not strong)
    npts, nt, nv = len(c["source_xyz"]), len(c["target_tri"]), len(c["target_xyz"])
    metric = c["metric"]
    # weak: the local scan is this rank's whole shard; strong: a share of the one cloud
    if emu:
        sh = ShardedICP(device=local, rank=emu[0], world=emu[1])
    else:
        sh = ShardedICP(device=local, rank=rank if strong else 0, world=world if strong else 1)
    if args.width:
        sh.ctx.set_group_width(args.width)
    if args.leaf:
        sh.ctx.set_leaf_size(args.leaf)
    if args.local >= 0:
        sh.ctx.set_local(args.local)
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
    if args.bail >= 0:
        sh.ctx.set_bail(args.bail)
    sh.set_target(c["target_xyz"], c["target_tri"])
    sh.set_source(c["source_xyz"], c["source_normals"])
    build_s = sh.ctx.build_seconds
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

    for _ in range(warmup):
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
    for i in range(steps):
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

    each = np.array(each)
    reg_ms = float(1e3 * np.median(each))
    total = float(nq_job) * ITERATIONS * steps
    res = {
        "value": total / elapsed,
        "ms_per_step": 1e3 * elapsed / steps,
        "steps": steps,
        "warmup": warmup,
        "config": {
            "workload": "%s: %d-triangle bumpy torus vs %d source points %s, %s metric, every "
                        "point used; one step = one cold-start registration of %d iterations" % (
                            name, nt, npts, "sharded over the ranks" if strong else "per GPU",
                            ("point-to-point", "point-to-plane", "symmetric")[metric], ITERATIONS),
            "iterations_per_step": ITERATIONS,
            "icp_iterations_per_s": ITERATIONS * steps / elapsed,
            "registration_ms_median": reg_ms,
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
        "roofline": {"bound": "hbm", "kernel": "k_search3w", "achieved": None, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": None, "traffic": None},
    }
    cfg = res["config"]
    if pass_rows:
        # rank 0's search kernels, pass by pass, of the registrations that carried events
        pm = np.array(pass_rows)
        kinds_a = np.array(kind_rows)
        cold = float(np.median(pm[:, 0]))                       # k_search3w, cold start
        steady = float(np.median(pm[:, ITERATIONS // 2:]))      # converged passes
        b_struct = structure_bytes(nv, nt)
        b_cold = 12.0 * nq_rank + b_struct     # every point once + one read of the structure
        b_steady = 20.0 * nq_rank              # point 12 B + certificate 4 B + winner 4 B
        steady_kernels = "k_certify (large and noisy clouds) + k_search_local + k_fallback + k_contested"
        # A settled pass of a clean cloud below four million points is a lean one: the
        # certificate check and the fused rows and sums in ONE kernel (kind 2), whose time is
        # therefore the whole pass but the reduction: its bytes are the check's (point 12 B,
        # certificate 4 B) plus the rows' (point 12 B, winner 4 B, kept index 4 B, the
        # triangle's points 36 B, its {normal, cell} record 16 B, the source normal 12 B where
        # the metric reads it).
        lean = kinds_a[:, ITERATIONS // 2:] == 2
        steady_lean = bool(lean.mean() > 0.5)
        b_lean = (16.0 + 72.0 + (12.0 if c["source_normals"] is not None else 0.0)) * nq_rank
        if steady_lean:
            steady = float(np.median(pm[:, ITERATIONS // 2:][lean]))
            b_steady = b_lean
            steady_kernels = "k_finalize_mx<metric, sums, LEAN> (certificate check + rows + sums in one kernel)"
        ach_cold = b_cold / (cold * 1e-3) / 1e9
        ach_steady = b_steady / (steady * 1e-3) / 1e9
        search_total = float(pm.sum(1).mean())
        tot_lean = float((pm * (kinds_a == 2)).sum(1).mean())
        tot_tree = float((pm * (kinds_a == 0)).sum(1).mean())
        n_lean = float((kinds_a == 2).sum(1).mean())
        n_tree = float((kinds_a == 0).sum(1).mean())
        reg_ev = cfg["registration_ms_median_with_kernel_events"] or reg_ms
        if args.dump_passes:
            cfg["pass_ms"] = [round(float(v), 4) for v in np.median(pm, 0)]
            cfg["pass_kind"] = [int(v) for v in kinds_a[0]]
        cfg["cold_pass_corr_per_s"] = nq_rank / (cold * 1e-3)
        cfg["steady_pass_corr_per_s"] = nq_rank / (steady * 1e-3)
        cfg["search_kernels_ms_per_registration"] = search_total
        cfg["lean_passes_per_registration"] = n_lean
        cfg["lean_kernel_ms_per_registration"] = tot_lean
        cfg["tree_pass_kernels_ms_per_registration"] = tot_tree
        traffic, traffic_src = measured_traffic(name) if world == 1 and not emu else (None, None)
        t_cold = traffic.get("k_search3w_cold") if traffic else None
        t_lean = traffic.get("k_finalize_mx_lean") if traffic else None
        # The top-level figures are those of the kernel that SEARCHES, on its most expensive
        # launch (the cold pass): the one walk per wavefront over the whole cloud.  It is the
        # top row of the rocprofv3 summary on every workload but the clean default one, where
        # the 43 launches of the lean kernel (which only checks that nothing needs searching)
        # add up to about as much; `kernels` lists both with their shares of a registration.
        kernels = [{
            "kernel": "k_search3w<4,*> + k_contested (whole-cloud tree passes: the cold pass and the hinted ones after it)",
            "launches_per_registration": n_tree,
            "ms_per_registration": tot_tree,
            "share_of_registration": tot_tree / reg_ev,
            "ms_cold_launch": cold,
            "algorithmic_bytes_per_launch": b_cold,
            "achieved_GBs_cold_launch": ach_cold,
            "frac_cold_launch": ach_cold / HBM_PEAK_GBS,
            "achieved_GBs_mean_launch": b_cold * n_tree / (tot_tree * 1e-3) / 1e9 if tot_tree > 0 else None,
            "traffic_bytes_cold_launch": t_cold["fetch_max"] + t_cold["write_mean"] if t_cold else None,
        }]
        if n_lean > 0:
            kernels.append({
                "kernel": "k_finalize_mx<metric, sums, LEAN> (a settled pass: certificate check + rows + sums in one kernel; it searches nothing)",
                "launches_per_registration": n_lean,
                "ms_per_registration": tot_lean,
                "share_of_registration": tot_lean / reg_ev,
                "ms_per_launch": tot_lean / n_lean,
                "algorithmic_bytes_per_launch": b_lean,
                "achieved_GBs": b_lean / (tot_lean / n_lean * 1e-3) / 1e9,
                "frac": b_lean / (tot_lean / n_lean * 1e-3) / 1e9 / HBM_PEAK_GBS,
                "traffic_bytes_per_launch": t_lean["fetch_mean"] + t_lean["write_mean"] if t_lean else None,
            })
        res["roofline"] = {
            "bound": "hbm",
            "kernel": "k_search3w<4,false> (the cold pass: one walk per wavefront over the whole "
                      "cloud; the kernel that searches, on its most expensive launch)",
            "achieved": ach_cold,
            "peak": HBM_PEAK_GBS,
            "unit": "GB/s",
            "frac": ach_cold / HBM_PEAK_GBS,
            "traffic": t_cold["fetch_max"] + t_cold["write_mean"] if t_cold else None,
            "traffic_source": ("%s: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                               "command (counters cannot run inside the timed region); FETCH_SIZE x 2, "
                               "the cold launch is the kernel's largest" % traffic_src) if t_cold else None,
            "algorithmic_bytes_per_launch": b_cold,
            "kernel_ms": cold,
            "launches": int(len(pass_rows)),
            "kernels": kernels,
            # the whole registration as one figure: the algorithmic bytes of its 50 passes (each
            # a read of every point and of the search structure) over its wall time
            "registration": {"algorithmic_bytes": ITERATIONS * b_cold, "ms": reg_ms,
                             "achieved_GBs": ITERATIONS * b_cold / (reg_ms * 1e-3) / 1e9,
                             "frac": ITERATIONS * b_cold / (reg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS},
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
    sh.ctx.close()
    return res, c


def cpu_baseline(c, repeats=3):
    """The CPU restatement of the vtkSMPTools path on the host's cores: the correspondence
    filter's pass at the start pose (what the GPU's cold pass does), repeated like the
    reference's own timing harness (ICP/Testing/Cxx/TestPolyDataCorrespondenceFilter.cxx:55-64:
    Update(); Modified(); in a loop, mean taken)."""
    from oracle import oracle as orc
    npts, nt = len(c["source_xyz"]), len(c["target_tri"])
    mesh = orc.Mesh(c["target_xyz"], c["target_tri"])
    nsample = min(npts, 1_000_000 if nt <= 2_000_000 else 250_000)
    threads = orc.num_threads()
    times = []
    for _ in range(repeats):
        t0 = time.perf_counter()
        mesh.correspond(c["source_xyz"][:nsample], step=1, max_dist=c["max_dist"],
                        src_normals=None if c["source_normals"] is None
                        else c["source_normals"][:nsample], grid=True, threads=threads)
        times.append(time.perf_counter() - t0)
    dt = float(np.mean(times))
    return {
        "value": nsample / dt,
        "unit": "correspondences/s",
        "cores": threads,
        "kind": "port",
        "sample": "mean of %d correspondence-filter passes at the start pose (what the GPU's cold "
                  "pass does: compare config.cold_pass_corr_per_s) over the first %d source "
                  "points of the same workload, bucket-grid locator restatement of the "
                  "vtkSMPTools path, %.1f s each (%s)" % (
                      repeats, nsample, dt, ", ".join("%.2f" % t for t in times)),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20, help="timed registrations")
    ap.add_argument("--warmup", type=int, default=3, help="untimed registrations first")
    ap.add_argument("--workload", default="",
                    help="c3 (default), c4, c5big, c3small; by default a one-GPU run measures c3 and "
                         "then c4 (config.c4)")
    ap.add_argument("--no-c4", action="store_true", help="the default run without its c4 part")
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
                    help="development: 2 shards by chunks of the curve order (default), 1 by ranges "
                         "of it, 0 by blocks of indices")
    ap.add_argument("--certify-min", type=int, default=-1,
                    help="development: queries per rank from which the large-cloud policy applies")
    ap.add_argument("--lean", type=int, default=-1, help="development: icpx_set_lean")
    ap.add_argument("--noisy-policy", type=int, default=-1, help="development: icpx_set_noisy_policy")
    ap.add_argument("--curve", type=int, default=-1, help="development: icpx_set_query_curve")
    ap.add_argument("--bail", type=int, default=-1, help="development: icpx_set_bail")
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
    from iterativeclosestpoint_amd.distributed import ShardedICP, shard_size

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
    modules = (torch, dist, icpx, synth, ShardedICP, shard_size)
    name = args.workload or "c3"
    res, c = run_workload(args, name, args.steps, args.warmup, strong, emu, rank, world, local,
                          collective, modules)
    # The north star's own target run, timed by the same command: only on one GPU (several
    # ranks keep to the one workload, whose failure modes the one line depends on), and only
    # when no workload was asked for by name.
    c4 = None
    if not args.workload and not args.no_c4 and world == 1 and not emu and not args.force_collective:
        try:
            r4, c4data = run_workload(args, "c4", 3, 1, False, None, rank, world, local, collective, modules)
            c4 = {"value": r4["value"], "unit": "correspondences/s", "ms_per_step": r4["ms_per_step"],
                  "steps": r4["steps"], "warmup": r4["warmup"], "roofline": r4["roofline"]}
            c4.update(r4["config"])
            del c4data
        except Exception as exc:  # (the headline figures above stand whatever happens here)
            c4 = {"error": repr(exc)}
    # ... and a target the benchmark's tori are not: a million triangles whose edge lengths
    # range over a factor of 65 (the walk's batches given up on), briefly
    c6 = None
    if not args.workload and not args.no_c4 and world == 1 and not emu and not args.force_collective:
        try:
            r6, c6data = run_workload(args, "c6", 5, 1, False, None, rank, world, local, collective, modules)
            c6 = {"value": r6["value"], "unit": "correspondences/s", "ms_per_step": r6["ms_per_step"],
                  "steps": r6["steps"], "warmup": r6["warmup"], "workload": r6["config"]["workload"],
                  "registration_ms_median": r6["config"]["registration_ms_median"],
                  "cold_pass_ms": r6["roofline"].get("kernel_ms"), "passes": r6["config"]["passes"]}
            del c6data
        except Exception as exc:
            c6 = {"error": repr(exc)}

    if rank == 0:
        out = {
            "metric": "point-to-mesh correspondences/sec (ICP iterations/sec in config)",
            "value": res["value"],
            "unit": "correspondences/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": res["ms_per_step"],
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": res["config"],
            "roofline": res["roofline"],
        }
        if c4 is not None:
            out["config"]["c4"] = c4
        if c6 is not None:
            out["config"]["c6"] = c6
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(c)
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out) + "\n").encode())
    if collective:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
