"""Shards of the kept source points (icpx_set_shard) and the loop of several ranks
(icpx_run_advance): rows and sums of the shards against the whole, the loop against one
rank's fused loop with lean passes off, on and made to fail, and the host's policy switches
across their boundaries."""
import os

import numpy as np
import pytest
import torch  # noqa: F401  (before libicpx.so is loaded: both bring a HIP runtime, torch's must come first)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def icpx():
    from iterativeclosestpoint_amd import icpx as m
    assert m.device_count() >= 1, "no HIP device visible"
    return m


@pytest.fixture(scope="module")
def synth():
    from iterativeclosestpoint_amd import synth as s
    return s


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as o
    return o


def rel_err(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / np.abs(np.asarray(b)).max())


@pytest.mark.parametrize("order", (0, 1, 2))
def test_shard_rows_and_sums_add_up(icpx, synth, order):
    """Three shards, cut each of the three ways: every kept point is in exactly one of them,
    a shard's rows are the whole's rows of its points (bit for bit, by ascending source
    index), and the shards' sums add up to the whole's."""
    from iterativeclosestpoint_amd.distributed import shard_size
    xyz, tri = synth.bumpy_torus(200, 100)
    pts, f, fn = synth.sample_surface(xyz, tri, 60_001, seed=3)
    M = synth.rigid(ry=4.0, t=(0.02, -0.01, 0.01))
    src = synth.apply(M, pts)
    sn = (fn.astype(np.float64) @ M[:3, :3].T).astype(np.float32)
    T = synth.rigid(rz=1.0, t=(0.001, 0.002, -0.001))
    step, world, md = 2, 3, 0.05
    ctx = icpx.Context(0)
    ctx.set_target(xyz, tri)
    ctx.set_source(src, sn)
    whole = ctx.correspond(T, step=step, max_dist=md, min_dot=0.2, use_source_normals=True)
    whole_sums = ctx.accumulate(2, T, step=step, max_dist=md, min_dot=0.2, use_source_normals=1)
    assert 0 < len(whole["cell"]) < (len(src) + step - 1) // step   # the threshold drops some
    ctx.set_shard_order(order)
    seen, sums = [], np.zeros(icpx.NSUMS)
    nk = (len(src) + step - 1) // step
    for r in range(world):
        ctx.set_shard(r, world)
        rows = ctx.correspond(T, step=step, max_dist=md, min_dot=0.2, use_source_normals=True)
        assert np.all(np.diff(rows["src_index"]) > 0)
        at = np.searchsorted(whole["src_index"], rows["src_index"])
        assert np.array_equal(whole["src_index"][at], rows["src_index"])
        for k in ("cell", "dist", "closest", "normal", "origin", "origin_normal"):
            assert np.array_equal(rows[k], whole[k][at]), (order, r, k)
        seen.append(rows["src_index"])
        part = ctx.accumulate(2, T, step=step, max_dist=md, min_dot=0.2, use_source_normals=1)
        assert part[0] == len(rows["cell"])
        sums += part
        if order == 2:
            # (every point of the shard, accepted or not: the mirror of the library's split)
            assert len(ctx.correspond(T, step=step, max_dist=1e9)["cell"]) == shard_size(nk, r, world)
    allidx = np.concatenate(seen)
    assert len(np.unique(allidx)) == len(allidx) == len(whole["src_index"])
    assert sums[0] == whole_sums[0]
    assert np.abs(sums - whole_sums).max() <= 1e-12 * np.abs(whole_sums).max()
    if order == 2:
        assert sum(shard_size(nk, r, world) for r in range(world)) == nk


def test_shared_loop_of_one_rank_equals_the_fused_loop(icpx, synth):
    """icpx_run_advance / icpx_run_update with nothing between them against icpx_run: the same
    arithmetic in the same order, so the same bits, lean passes included."""
    from iterativeclosestpoint_amd.distributed import ShardedICP
    c = synth.config_c3(nu=300, nv=150, npts=120_000)
    sh = ShardedICP(device=0, rank=0, world=1)
    sh.set_target(c["target_xyz"], c["target_tri"])
    sh.set_source(c["source_xyz"], c["source_normals"])
    p = icpx.default_params()
    p.metric, p.max_iterations, p.max_landmarks = 2, 40, len(c["source_xyz"])
    p.max_distance = c["max_dist"]
    ref = sh.ctx.run(p)
    for mode in (0, 1, 3):
        sh.ctx.set_lean(mode)
        before = sh.ctx.debug_stats()
        got = sh.run_device(p, emulate=True)
        after = sh.ctx.debug_stats()
        assert np.array_equal(got[0], ref[0]) and got[1:] == ref[1:], mode
        lean = after["lean_passes"] - before["lean_passes"]
        failed = after["lean_failures"] - before["lean_failures"]
        assert (lean == 0) if mode == 0 else (lean >= 20), (mode, lean)
        assert (failed >= 5) if mode == 3 else (failed <= 3), (mode, failed)
    # a loop that counts its iterations itself must not be mixed with one that does not
    sh.ctx.run_begin(p)
    assert sh.ctx.run_advance(sh.sums_dev.data_ptr())
    with pytest.raises(icpx.IcpxError):
        sh.ctx.run_enqueue(sh.sums_dev.data_ptr())
    sh.ctx.run_update(sh.sums_dev.data_ptr())
    sh.ctx.run_end()


def _rank_lean_modes(rank, world, port, out):
    """One rank of test_two_ranks_with_lean_passes (spawned)."""
    import torch.distributed as dist
    from iterativeclosestpoint_amd import icpx as ix, synth as sy
    from iterativeclosestpoint_amd.distributed import ShardedICP
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    c = sy.config_c3(nu=300, nv=150, npts=120_001)
    sh = ShardedICP(device=0, rank=rank, world=world)
    sh.set_target(c["target_xyz"], c["target_tri"])
    sh.set_source(c["source_xyz"], c["source_normals"])
    p = ix.default_params()
    p.metric, p.max_iterations, p.max_landmarks = 2, 40, len(c["source_xyz"])
    p.max_distance = c["max_dist"]
    res = []
    for mode in (0, 1, 3):
        sh.ctx.set_lean(mode)
        before = sh.ctx.debug_stats()
        T, it, mean, st = sh.run_device(p)
        after = sh.ctx.debug_stats()
        res.append((mode, T, it, mean, st, after["lean_passes"] - before["lean_passes"],
                    after["lean_failures"] - before["lean_failures"]))
    out.put((rank, res))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_with_lean_passes(icpx, synth):
    """Two ranks (two processes sharing this GPU, gloo carrying the sums): with lean passes
    off, on, and with every third one made to fail on rank 0 alone (rank 1 learns of it
    through the all-reduce and drops the pass as well): the same transform to the last bit in
    all three, on both ranks, and the one rank's to rounding."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    mpc = mp.get_context("spawn")
    q = mpc.Queue()
    procs = [mpc.Process(target=_rank_lean_modes, args=(r, 2, port, q)) for r in range(2)]
    for pr in procs:
        pr.start()
    res = dict(q.get(timeout=300) for _ in procs)
    for pr in procs:
        pr.join(60)
        assert pr.exitcode == 0
    c = synth.config_c3(nu=300, nv=150, npts=120_001)
    one = icpx.Context(0)
    one.set_target(c["target_xyz"], c["target_tri"])
    one.set_source(c["source_xyz"], c["source_normals"])
    p = icpx.default_params()
    p.metric, p.max_iterations, p.max_landmarks = 2, 40, len(c["source_xyz"])
    p.max_distance = c["max_dist"]
    T1, it1, mean1, st1 = one.run(p)
    base = res[0][0]
    for r in (0, 1):
        for mode, T, it, mean, st, lean, failed in res[r]:
            assert np.array_equal(T, base[1]) and (it, mean, st) == base[2:5], (r, mode)
            assert (it, st) == (it1, st1) and rel_err(T, T1) <= 1e-10
            assert (lean == 0) if mode == 0 else (lean >= 15), (r, mode, lean)
            assert (failed >= 4) if mode == 3 else (failed <= 3), (r, mode, failed)
    assert [x[5:] for x in res[0]] == [x[5:] for x in res[1]]   # both ranks took the same decisions


def test_policy_switches_change_nothing(icpx, synth, oracle):
    """The host's policy on both sides of each of its boundaries (development switches in place
    of cloud sizes): a noisy scan of 300 000 points registers to the same bits whether its
    certificates are checked by the light kernel or by the shortcut, whether a noisy cloud goes
    by the large cloud's policy or not, along either curve, whether the sets of certified queries
    are weighed by the fused finalize or by k_contested, every pass or only when the winner's lead has run out; the rows at the end are the oracle's.  Then the density rule: a cloud just below and just above a quarter of the
    triangles."""
    xyz, tri = synth.bumpy_torus(300, 150)
    pts, f, fn = synth.sample_surface(xyz, tri, 300_000, seed=11)
    d = synth.diag(xyz)
    edge = float(np.linalg.norm(xyz[tri[:, 0]] - xyz[tri[:, 1]], axis=1).mean())
    noise = np.stack([synth.gaussian(11, 32 + 2 * k, len(pts)) for k in range(3)], -1) * (2.0 * edge)
    M = synth.rigid(ry=4.0, t=(0.01 * d / np.sqrt(3),) * 3)
    src = synth.apply(M, pts.astype(np.float64) + noise)
    p = icpx.default_params()
    p.metric, p.max_iterations, p.max_landmarks, p.max_distance = 1, 14, len(src), 10.0
    ctx = icpx.Context(0)
    ctx.set_target(xyz, tri)
    ctx.set_source(src)
    res = {}
    for name, noisy, cmin, local, curve, weigh in (("default", 1, -1, 1, 1, 3), ("by size alone", 0, -1, 1, 1, 3),
                                                    ("large", 1, 100_000, 1, 1, 3), ("light kernel", 1, -1, 3, 1, 3),
                                                    ("morton", 1, -1, 1, 0, 3), ("every set weighed by k_contested", 1, -1, 1, 1, 0),
                                                    ("every set weighed by the finalize", 1, -1, 1, 1, 1),
                                                    ("leads kept, k_contested", 1, -1, 1, 1, 2),
                                                    ("large, no leads", 1, 100_000, 1, 1, 0)):
        ctx.set_local(local)
        ctx.set_noisy_policy(noisy)
        ctx.set_certify_min(cmin)
        ctx.set_query_curve(curve)
        ctx.set_weigh(weigh)
        before = ctx.debug_stats()
        T, it, mean, st = ctx.run(p)
        after = ctx.debug_stats()
        res[name] = (T, it, mean, st, {k: after[k] - before[k] for k in ("certify_passes", "direct_passes",
                                                                          "local_passes", "tree_passes")})
    T0 = res["default"][0]
    assert np.abs(T0 @ M - np.eye(4)).max() < 2e-2   # (14 iterations: the turn about the torus's axis is still coming in)
    for name, r in res.items():
        # (the other curve orders the queries, and so the additions, differently)
        if name == "morton":
            assert rel_err(r[0], T0) <= 1e-9 and r[1] == res["default"][1], name
        else:
            assert np.array_equal(r[0], T0) and r[1:4] == res["default"][1:4], name
    assert res["default"][4]["direct_passes"] >= 5, res["default"][4]     # noisy: the large cloud's policy
    assert res["by size alone"][4]["certify_passes"] == 0, res["by size alone"][4]
    assert res["large"][4]["certify_passes"] >= 5, res["large"][4]
    ctx.set_local(1)
    ctx.set_noisy_policy(1)
    ctx.set_certify_min(-1)
    ctx.set_query_curve(1)
    ctx.set_weigh(3)
    rows = ctx.correspond(T0, step=1, max_dist=10.0)
    pick = (np.arange(3_000) * 97 + 3).astype(np.int64)
    mesh = oracle.Mesh(xyz, tri)
    ref = mesh.correspond(src[pick], T=T0, step=1, max_dist=10.0, grid=True)
    for k in ("cell", "dist", "closest"):
        assert np.array_equal(rows[k][pick], ref[k]), k
    # the density rule (256 nq >= nt picks the walk per wavefront for whole passes), and a cloud
    # that was on the other side of it while the rule was 4 nq >= nt
    nt = len(tri)
    for n in (nt // 256 - 100, nt // 256 + 100, nt // 4 - 100):
        ctx2 = icpx.Context(0)
        ctx2.set_group_width(1)
        ctx2.set_target(xyz, tri)
        ctx2.set_source(src[:n])
        got = ctx2.correspond(np.eye(4), step=1, max_dist=10.0)
        again = ctx2.correspond(synth.rigid(t=(1e-4, 0, 0)), step=1, max_dist=10.0)
        pick = np.arange(0, n, 9)
        ref = mesh.correspond(src[:n][pick], step=1, max_dist=10.0, grid=True)
        ref2 = mesh.correspond(src[:n][pick], T=synth.rigid(t=(1e-4, 0, 0)), step=1, max_dist=10.0, grid=True)
        for k in ("cell", "dist", "closest"):
            assert np.array_equal(got[k][pick], ref[k]), (n, k)
            assert np.array_equal(again[k][pick], ref2[k]), (n, k)
    # the cold pass's first bound: Morton curve alone, grid and curve, grid alone; the points
    # start a hundred edge lengths off the surface, where the three seeds differ most
    far = synth.apply(synth.rigid(rx=3.0, t=(0.05 * d, 0.02 * d, 0.0)), src[:120_000])
    ctx3 = icpx.Context(0)
    ctx3.set_group_width(1)
    ctx3.set_target(xyz, tri)
    ctx3.set_source(far)
    base = None
    for mode in (1, 0, 2):
        ctx3.set_seed(mode)
        ctx3.reset_hints()
        got = ctx3.correspond(np.eye(4), step=1, max_dist=10.0)
        if base is None:
            base = got
            pick = np.arange(0, len(far), 41)
            ref = mesh.correspond(far[pick], step=1, max_dist=10.0, grid=True)
            for k in ("cell", "dist", "closest"):
                assert np.array_equal(got[k][pick], ref[k]), k
        else:
            for k in ("cell", "dist", "closest", "src_index"):
                assert np.array_equal(got[k], base[k]), (mode, k)


def _job_in_turn(icpx, ranks, p):
    """The ranks of a job run in turn on the one GPU (one stream), the all-reduce replaced by a
    sum of their vectors on the device; every rank must decide alike."""
    stream = ranks[0].stream
    for sh in ranks:
        sh.ctx.set_stream(stream.cuda_stream)
        sh.ctx.run_begin(p)
    ptrs = [sh.sums_dev.data_ptr() for sh in ranks]
    passes = 0
    with torch.cuda.stream(stream):
        while True:
            alive = [sh.ctx.run_advance(ptr) for sh, ptr in zip(ranks, ptrs)]
            assert all(a == alive[0] for a in alive), alive
            if not alive[0]:
                break
            total = torch.stack([sh.sums_dev for sh in ranks]).sum(0)
            for sh, ptr in zip(ranks, ptrs):
                sh.sums_dev.copy_(total)
                sh.ctx.run_update(ptr)
            passes += 1
            assert passes < 1000
    return [sh.ctx.run_end() for sh in ranks]


@pytest.mark.parametrize("npts,world,metric", [(10, 16, 1), (5, 3, 1), (7, 4, 0), (2000, 3, 2), (70_000, 5, 1)])
def test_job_in_one_process_with_small_and_empty_shards(icpx, synth, npts, world, metric):
    """A job of several ranks, some of which get a handful of points or none (ten points over
    sixteen ranks), run in turn in this process with the job's sums behind every solve: every
    rank ends with the same transform, iteration count and status, and they are the one-rank
    run's (to rounding: the shards' sums add up in another order); too few rows for the metric
    stop all ranks alike."""
    from iterativeclosestpoint_amd.distributed import ShardedICP
    c = synth.config_c3(nu=120, nv=60, npts=max(npts, 16))
    src, sn = c["source_xyz"][:npts], c["source_normals"][:npts]
    p = icpx.default_params()
    p.metric, p.max_iterations, p.max_landmarks, p.max_distance = metric, 12, npts, c["max_dist"]
    one = icpx.Context(0)
    one.set_target(c["target_xyz"], c["target_tri"])
    one.set_source(src, sn)
    ref = one.run(p)
    ranks = [ShardedICP(device=0, rank=r, world=world) for r in range(world)]
    torch.cuda.synchronize()
    for sh in ranks:
        sh.set_target(c["target_xyz"], c["target_tri"])
        sh.set_source(src, sn)
    res = _job_in_turn(icpx, ranks, p)
    for r in res[1:]:
        assert np.array_equal(r[0], res[0][0]) and r[1:] == res[0][1:]
    assert res[0][1] == ref[1] and res[0][3] == ref[3]
    assert rel_err(res[0][0], ref[0]) <= 1e-9
    if npts == 5:
        assert ref[3] == 2 and ref[1] == 0    # five rows are too few for the six unknowns: nothing is solved
