"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same
seeded inputs.  Bars: cell ids bit-exact (except true equidistant ties, which are
counted and must be zero on these seeds), distances and closest points bit-exact,
normals within 1e-6 (the device's acos may differ from libm's by an ulp before the
float rounding), sums within 1e-11 relative, 4x4 transforms within 1e-6 relative."""
import os
import sys

import numpy as np
import pytest
import torch  # noqa: F401  (before libicpx.so is loaded: both bring a HIP runtime, torch's must come first)

from helpers import hand_cases, rel_err, sums_from_rows

pytestmark = pytest.mark.gpu

WIDTHS = (1, 4, 8, 16, 32, 64)


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
def small(synth, oracle):
    c = synth.config_c1()
    mesh = oracle.Mesh(c["target_xyz"], c["target_tri"])
    return c, mesh


def queries_around(synth, xyz, tri, n, seed):
    """Mixed query set: on the surface, near it, far inside/outside, exactly on vertices."""
    pts, f, fn = synth.sample_surface(xyz, tri, n, seed=seed)
    r = synth.u01(seed, 7, n)
    off = np.where(r < 0.25, 0.0, np.where(r < 0.6, 0.01, np.where(r < 0.9, 0.15, 1.5)))
    sign = np.where(synth.u01(seed, 8, n) < 0.5, -1.0, 1.0)
    q = pts.astype(np.float64) + fn.astype(np.float64) * (off * sign)[:, None]
    q = q.astype(np.float32)
    nv = min(n // 10, len(xyz))
    q[:nv] = xyz[:nv]  # exactly on vertices: every incident triangle ties
    return q


def check_rows(got, ref, allow_ties=0):
    assert len(got["cell"]) == len(ref["cell"])
    assert np.array_equal(got["src_index"], ref["src_index"])
    diff = np.nonzero(got["cell"] != ref["cell"])[0]
    assert len(diff) <= allow_ties, "cell ids differ at rows %s" % diff[:10]
    assert np.array_equal(got["dist"], ref["dist"]), "distances are not bit-identical"
    assert np.array_equal(got["origin"], ref["origin"])
    same = got["cell"] == ref["cell"]
    assert np.array_equal(got["closest"][same], ref["closest"][same])
    assert np.abs(got["normal"][same] - ref["normal"][same]).max() <= 1e-6


@pytest.mark.parametrize("exhaustive", (False, True))
@pytest.mark.parametrize("width", WIDTHS)
def test_closest_point_matches_bruteforce(icpx, synth, small, width, exhaustive):
    c, mesh = small
    q = queries_around(synth, c["target_xyz"], c["target_tri"], 3000, seed=11 + width)
    ctx = icpx.Context(0)
    ctx.set_group_width(width)
    ctx.set_exhaustive(exhaustive)
    ctx.set_target(c["target_xyz"], c["target_tri"])
    ctx.set_source(q)
    got = ctx.correspond(np.eye(4), step=1, max_dist=1e30)
    ref = mesh.correspond(q, step=1, max_dist=1e30, grid=False)
    assert len(ref["cell"]) == len(q)
    check_rows(got, ref)
    # second pass over the same context starts from the previous winners: same answer
    got2 = ctx.correspond(np.eye(4), step=1, max_dist=1e30)
    check_rows(got2, ref)
    assert {0, 1, 2} <= set(np.unique(ref["kind"]))  # face, edge and vertex normals all hit


def test_wavefront_walk_deals_out_every_batch(icpx, synth, small):
    """One walk per wavefront: batches of 64 queries are dealt out in chunks of 128 batches to
    eight runs with a pool each.  Cloud sizes on both sides of a batch, of a chunk and of a
    whole cycle of chunks: every query is served once (rows equal to the oracle's), cold and
    from the last winners."""
    c, mesh = small
    nmax = 8 * 128 * 64 + 130
    q_all = queries_around(synth, c["target_xyz"], c["target_tri"], nmax, seed=23)
    ref_all = mesh.correspond(q_all, step=1, max_dist=1e30, grid=True)
    assert np.array_equal(ref_all["src_index"], np.arange(nmax))
    ctx = icpx.Context(0)
    ctx.set_group_width(1)
    ctx.set_persistent(3)
    ctx.set_target(c["target_xyz"], c["target_tri"])
    for n in (1, 63, 64, 65, 128 * 64 - 1, 128 * 64, 128 * 64 + 1, 3 * 128 * 64 + 64 * 5 + 7,
              8 * 128 * 64 - 1, 8 * 128 * 64, nmax):
        ctx.set_source(q_all[:n])
        ref = {k: v[:n] for k, v in ref_all.items()}
        check_rows(ctx.correspond(np.eye(4), step=1, max_dist=1e30), ref)
        check_rows(ctx.correspond(np.eye(4), step=1, max_dist=1e30), ref)
    ctx.close()


def test_cube_known_answers(icpx):
    v = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [0, 0, 1], [1, 0, 1], [1, 1, 1],
                  [0, 1, 1]], np.float32)
    t = np.array([[0, 2, 1], [0, 3, 2], [4, 5, 6], [4, 6, 7], [0, 1, 5], [0, 5, 4], [2, 3, 7],
                  [2, 7, 6], [1, 2, 6], [1, 6, 5], [0, 4, 7], [0, 7, 3]], np.int32)
    q = np.array([[2, 2, 2], [0.5, 0.5, 2], [0.5, 2, 2], [0.25, 0.5, 0.9]], np.float32)
    ctx = icpx.Context(0)
    ctx.set_target(v, t)
    ctx.set_source(q)
    r = ctx.correspond(np.eye(4), max_dist=100.0)
    s3, s2 = 1 / np.sqrt(3), 1 / np.sqrt(2)
    assert np.allclose(r["closest"], [[1, 1, 1], [0.5, 0.5, 1], [0.5, 1, 1], [0.25, 0.5, 1]])
    assert np.allclose(r["normal"], [[s3, s3, s3], [0, 0, 1], [0, s2, s2], [0, 0, 1]], atol=1e-7)
    assert np.allclose(r["dist"], [np.sqrt(3), 1, np.sqrt(2), 0.1], rtol=1e-6)  # 0.9f != 0.9
    assert list(r["cell"]) == [2, 2, 3, 3] or r["cell"][0] in (2, 3, 7, 8, 9)


def test_filter_threshold_stride_and_transform(icpx, synth, small):
    c, mesh = small
    T = synth.rigid(rx=3, ry=-4, rz=2, t=(0.02, -0.01, 0.03))
    ctx = icpx.Context(0)
    ctx.set_target(c["target_xyz"], c["target_tri"])
    ctx.set_source(c["source_xyz"])
    for step, md in ((1, 0.05), (7, 0.2), (20, 10.0), (2048, 10.0), (5000, 10.0)):
        got = ctx.correspond(T, step=step, max_dist=md)
        ref = mesh.correspond(c["source_xyz"], T=T, step=step, max_dist=md, grid=False)
        check_rows(got, ref)
    got = ctx.correspond(T, step=1, max_dist=1e-9)  # nothing passes the threshold
    assert len(got["cell"]) == 0


def test_symmetric_rows_and_min_dot(icpx, synth, small):
    c, mesh = small
    sn = synth.vertex_normals(c["source_xyz"], c["target_tri"])
    ctx = icpx.Context(0)
    ctx.set_target(c["target_xyz"], c["target_tri"])
    ctx.set_source(c["source_xyz"], sn)
    for md in (0.0, 0.7):
        got = ctx.correspond(np.eye(4), step=2, max_dist=10.0, min_dot=md,
                             use_source_normals=True)
        ref = mesh.correspond(c["source_xyz"], step=2, max_dist=10.0, src_normals=sn, min_dot=md,
                              grid=False)
        check_rows(got, ref)
        assert np.array_equal(got["origin_normal"], ref["origin_normal"])
    assert len(ref["cell"]) < (len(sn) + 1) // 2  # the dot test rejected something


def test_user_cell_normals(icpx, synth, small, oracle):
    c, _ = small
    cn = synth.face_normals(c["target_xyz"], c["target_tri"]).astype(np.float32)
    cn = -cn  # flipped on purpose: must be used verbatim
    mesh = oracle.Mesh(c["target_xyz"], c["target_tri"], cn)
    q = queries_around(synth, c["target_xyz"], c["target_tri"], 1000, seed=5)
    ctx = icpx.Context(0)
    ctx.set_target(c["target_xyz"], c["target_tri"], cn)
    ctx.set_source(q)
    check_rows(ctx.correspond(np.eye(4), max_dist=1e30),
               mesh.correspond(q, max_dist=1e30, grid=False))


@pytest.mark.parametrize("metric", (0, 1, 2))
def test_fused_sums_match_rows(icpx, synth, small, metric):
    c, mesh = small
    sn = synth.vertex_normals(c["source_xyz"], c["target_tri"])
    ctx = icpx.Context(0)
    ctx.set_target(c["target_xyz"], c["target_tri"])
    ctx.set_source(c["source_xyz"], sn)
    T = synth.rigid(ry=2.0, t=(0.0, 0.01, 0.0))
    ref = mesh.correspond(c["source_xyz"], T=T, step=1, max_dist=0.2,
                          src_normals=sn if metric == 2 else None, grid=False)
    want = sums_from_rows(metric, ref["origin"], ref["closest"], ref["normal"],
                          ref["origin_normal"], ctx.origin)
    got = ctx.accumulate(metric, T, step=1, max_dist=0.2)
    assert got[0] == want[0] == len(ref["cell"])
    scale = np.abs(want).max()
    assert np.abs(got - want).max() <= 1e-11 * scale
    again = ctx.accumulate(metric, T, step=1, max_dist=0.2)
    assert np.array_equal(got, again), "sums are not reproducible run to run"
    M, n, md, rc = icpx.solve(metric, got, ctx.origin)
    Mo, st = __import__("oracle.oracle", fromlist=["solve"]).solve(
        metric, ref["origin"], ref["closest"], ref["origin_normal"], ref["normal"])
    assert rc == 0 and st == 0
    assert rel_err(M, Mo) <= 1e-9


@pytest.mark.parametrize("metric,landmarks", ((0, 100), (1, 100), (1, 1000), (2, 100), (2, 2048)))
def test_icp_run_matches_oracle(icpx, synth, small, metric, landmarks):
    """C1 / C2: Bunny stand-in against its rigidly perturbed copy."""
    c, mesh = small
    sn = synth.vertex_normals(c["source_xyz"], c["target_tri"]) if metric == 2 else None
    md = 10.0 if metric == 0 else c2_threshold(synth, c)
    ctx = icpx.Context(0)
    ctx.set_target(c["target_xyz"], c["target_tri"])
    ctx.set_source(c["source_xyz"], sn)
    p = icpx.default_params()
    p.metric, p.max_iterations, p.max_landmarks, p.max_distance = metric, 50, landmarks, md
    T, it, mean, st = ctx.run(p)
    To, ito, mo, sto, n = mesh.icp(c["source_xyz"], metric=metric, max_iter=50,
                                   max_landmarks=landmarks, max_dist=md, src_normals=sn)
    assert (it, st) == (ito, sto)
    assert rel_err(T, To) <= 1e-6
    # and it actually registers: T undoes the demo perturbation
    assert np.abs(T @ synth.demo_perturbation() - np.eye(4)).max() < 1e-3
    # the reference's own float solver lands within float noise of the same transform
    T32 = mesh.icp(c["source_xyz"], metric=metric, max_iter=50, max_landmarks=landmarks,
                   max_dist=md, src_normals=sn, precision=32)[0]
    assert rel_err(T, T32) <= 2e-4


@pytest.mark.gpu
@pytest.mark.parametrize("metric", [0, 1, 2])
def test_device_loop_matches_host_loop(icpx, synth, small, metric):
    """The device-resident loop (solve, transform and stop test on the GPU) takes the same
    steps as the loop that brings the sums to the host every iteration, including the early
    stop, for a run driven through run_begin/enqueue/update/end as a multi-GPU driver does."""
    c, mesh = small
    sn = synth.vertex_normals(c["source_xyz"], c["target_tri"]) if metric == 2 else None
    ctx = icpx.Context(0)
    ctx.set_target(c["target_xyz"], c["target_tri"])
    ctx.set_source(c["source_xyz"], sn)
    for check in (0, 1):
        p = icpx.default_params()
        p.metric, p.max_iterations, p.max_landmarks = metric, 40, 2000
        p.max_distance = c2_threshold(synth, c)
        p.check_mean_distance, p.max_mean_distance = check, 2e-3
        Td, itd, md, std = ctx.run(p)                      # device loop
        ctx.set_allreduce(lambda sums: sums)               # a host hook selects the host loop
        Th, ith, mh, sth = ctx.run(p)
        ctx.set_allreduce(None)
        assert (itd, std) == (ith, sth)
        assert check == 0 or (std == 1 and itd < 40)
        assert rel_err(Td, Th) <= 1e-12 and abs(md - mh) <= 1e-12 * max(1.0, abs(mh))
        ctx.run_begin(p)                                   # the same, driven step by step
        for _ in range(p.max_iterations):
            if ctx.run_enqueue():
                break
            ctx.run_update()
        Ts, its, ms, sts = ctx.run_end()
        assert (its, sts) == (itd, std) and np.array_equal(Ts, Td) and ms == md


def c2_threshold(synth, c):
    return 0.25 * synth.diag(c["target_xyz"])


def test_icp_mean_distance_early_out(icpx, synth, small):
    c, mesh = small
    ctx = icpx.Context(0)
    ctx.set_target(c["target_xyz"], c["target_tri"])
    ctx.set_source(c["source_xyz"])
    p = icpx.default_params()
    p.metric, p.max_iterations, p.max_landmarks = 1, 100, 100
    p.check_mean_distance, p.max_mean_distance = 1, 1e-3
    T, it, mean, st = ctx.run(p)
    To, ito, mo, sto, n = mesh.icp(c["source_xyz"], metric=1, max_iter=100, max_landmarks=100,
                                   check_mean=True, max_mean=1e-3)
    assert st == sto == 1 and it == ito and it < 100
    assert abs(mean - mo) <= 1e-9 and mean <= 1e-3
    assert rel_err(T, To) <= 1e-6


def test_partial_overlap_outliers_and_empty(icpx, synth, oracle):
    """C5: half-overlapping scans with 20 % outliers, then a threshold nothing passes."""
    c = synth.config_c5()
    mesh = oracle.Mesh(c["target_xyz"], c["target_tri"])
    ctx = icpx.Context(0)
    ctx.set_target(c["target_xyz"], c["target_tri"])
    ctx.set_source(c["source_xyz"])
    got = ctx.correspond(np.eye(4), step=1, max_dist=c["max_dist"])
    ref = mesh.correspond(c["source_xyz"], step=1, max_dist=c["max_dist"], grid=True)
    check_rows(got, ref)
    frac = len(ref["cell"]) / len(c["source_xyz"])
    assert 0.2 < frac < 0.7  # the threshold rejects outliers and the non-overlapping half
    p = icpx.default_params()
    p.metric, p.max_iterations, p.max_landmarks, p.max_distance = 1, 10, 10 ** 9, c["max_dist"]
    T, it, mean, st = ctx.run(p)
    To, ito, mo, sto, n = mesh.icp(c["source_xyz"], metric=1, max_iter=10,
                                   max_landmarks=10 ** 9, max_dist=c["max_dist"])
    assert (it, st) == (ito, sto)
    assert rel_err(T, To) <= 1e-6
    assert np.abs(T @ c["truth"] - np.eye(4)).max() < 2e-2  # from 3.5e-2 before alignment
    p.max_distance = 1e-12  # empty correspondence set: stop, keep the transform, say so
    T, it, mean, st = ctx.run(p)
    assert st == icpx.RUN_TOO_FEW and it == 0 and np.array_equal(T, np.eye(4))
    assert mesh.icp(c["source_xyz"], metric=1, max_dist=1e-12)[3] == 2


def test_midsize_against_grid_oracle(icpx, synth, oracle):
    """200k triangles / 100k noisy samples: every width agrees with the bucket-grid oracle."""
    xyz, tri = synth.bumpy_torus(400, 250)
    pts, f, fn = synth.sample_surface(xyz, tri, 100_000, seed=3)
    noise = np.stack([synth.gaussian(3, 60 + 2 * k, len(pts)) for k in range(3)], -1) * 3e-3
    T = synth.rigid(ry=5.0, t=(0.02, 0.02, 0.02))
    src = synth.apply(T, pts.astype(np.float64) + noise)
    mesh = oracle.Mesh(xyz, tri)
    ref = mesh.correspond(src, step=1, max_dist=10.0, grid=True)
    for width, exhaustive, leaf, pooled in (
            (8, False, 2, 1), (64, False, 2, 1), (0, False, 2, 1), (4, False, 2, 1), (16, False, 2, 1),
            (8, True, 2, 1), (1, False, 1, 1), (1, False, 2, 1), (1, False, 4, 1), (1, False, 4, 4),
            (1, False, 1, 3), (1, False, 2, 3), (1, False, 4, 3)):
        ctx = icpx.Context(0)
        ctx.set_group_width(width)
        ctx.set_exhaustive(exhaustive)
        ctx.set_leaf_size(leaf)
        ctx.set_persistent(pooled)  # 1 pooled lanes, 3 one walk per wavefront, 4 by regime
        ctx.set_target(xyz, tri)
        ctx.set_source(src)
        check_rows(ctx.correspond(np.eye(4), max_dist=10.0), ref)          # cold
        check_rows(ctx.correspond(synth.rigid(rz=0.3), max_dist=10.0),     # hinted, moved
                   mesh.correspond(src, T=synth.rigid(rz=0.3), step=1, max_dist=10.0, grid=True))
        ctx.close()


def test_graded_mesh_and_batches_given_up_on(icpx, synth, oracle):
    """A target whose triangles' sizes range over a factor of 65 (squares to slivers), the
    cloud a few dozen of its small edge lengths off: a few batches of the wavefront walk have
    thousands of boxes within their bounds, and the walk gives up on them (their queries are
    searched one wavefront each).  Rows equal the oracle's with the library's own choice, with
    a budget so short that most batches are given up on, and with none; the registrations
    those three run are the same to the last bit."""
    c = synth.config_c6(300, 200, 80_000)
    xyz, tri, src = c["target_xyz"], c["target_tri"], c["source_xyz"]
    mesh = oracle.Mesh(xyz, tri)
    ref = mesh.correspond(src, step=1, max_dist=10.0, grid=True)
    T2 = synth.rigid(rz=0.3)
    ref2 = mesh.correspond(src, T=T2, step=1, max_dist=10.0, grid=True)
    p = icpx.default_params()
    p.metric, p.max_iterations, p.max_landmarks, p.max_distance = 1, 12, len(src), 10.0
    runs = []
    for bail in (1, 20, 0):
        ctx = icpx.Context(0)
        ctx.set_persistent(3)  # whole passes by the wavefront walk, whatever the density
        ctx.set_bail(bail)
        ctx.set_target(xyz, tri)
        ctx.set_source(src)
        check_rows(ctx.correspond(np.eye(4), max_dist=10.0), ref)   # cold
        check_rows(ctx.correspond(T2, max_dist=10.0), ref2)         # hinted, moved
        st = ctx.debug_stats()
        assert st["mesh_grading"] > 8.0
        assert (st["bail_passes"] > 0) == (bail != 0)
        runs.append(ctx.run(p))
        ctx.close()
    for T, it, mean, status in runs[1:]:
        assert np.array_equal(T, runs[0][0]) and it == runs[0][1] and mean == runs[0][2]
    # the oracle's own loop from the same start
    To = mesh.icp(src, metric=1, max_iter=12, max_landmarks=len(src), max_dist=10.0)[0]
    assert rel_err(runs[0][0], To) < 1e-6
    # a cloud whose given-up queries are more than go one wavefront each: the pooled lanes
    # serve them, drawing from the pools the walk has just drawn from
    c2 = synth.config_c6(300, 200, 420_000)
    src2 = c2["source_xyz"]
    ctx = icpx.Context(0)
    ctx.set_persistent(3)
    ctx.set_bail(2)
    ctx.set_target(xyz, tri)
    ctx.set_source(src2)
    got = ctx.correspond(np.eye(4), max_dist=10.0)
    assert ctx.debug_counters()["pool"] > 2 * 98304, ctx.debug_counters()
    pick = np.arange(0, len(src2), 5)
    ref = mesh.correspond(src2[pick], step=1, max_dist=10.0, grid=True)
    for k in ("cell", "dist", "closest"):
        assert np.array_equal(got[k][pick], ref[k]), k
    ctx.close()


def test_vtkicp_surface(icpx, synth, small):
    from iterativeclosestpoint_amd import vtkICP, vtkPolyData
    c, mesh = small
    tgt = vtkPolyData(c["target_xyz"], c["target_tri"])
    src = vtkPolyData(c["source_xyz"], c["target_tri"])
    src.SetPointNormals(synth.vertex_normals(c["source_xyz"], c["target_tri"]))
    icp = vtkICP.New()
    assert icp.GetMaximumNumberOfIterations() == 50 and icp.GetMaximumDistance() == 10.0
    assert icp.GetMetricModeAsString() == "Point-to-plane"
    icp.CheckMeanDistanceOn()
    icp.SetMaximumMeanDistance(0.001)
    icp.SetMetricToPlaneToPlane()
    icp.SetMaximumNumberOfIterations(100)
    icp.SetMaximumNumberOfLandmarks(100)
    icp.SetSource(src)
    icp.SetTarget(tgt)
    icp.Update()
    M = icp.GetMatrix()
    assert np.abs(M @ synth.demo_perturbation() - np.eye(4)).max() < 5e-3
    assert 0 < icp.GetNumberOfIterations() < 100 and icp.GetMeanDistance() <= 0.001
    To, ito, mo, sto, n = mesh.icp(
        c["source_xyz"], metric=2, max_iter=100, max_landmarks=100, check_mean=True,
        max_mean=0.001, src_normals=synth.vertex_normals(c["source_xyz"], c["target_tri"]))
    assert icp.GetNumberOfIterations() == ito and rel_err(M, To) <= 1e-6
    icp.Inverse()
    assert icp.GetSource() is tgt and icp.GetTarget() is src
    bad = vtkICP()
    bad.SetSource(vtkPolyData(c["source_xyz"].astype(np.float64), c["target_tri"]))
    bad.SetTarget(tgt)
    bad.Update()  # double points: error reported, matrix untouched
    assert np.array_equal(bad.GetMatrix(), np.eye(4))


def test_filter_surface(icpx, synth, oracle):
    """vtkPolyDataCorrespondenceFilter mirror: two outputs, stride rule, threshold, origin
    normals with MinNormalDot, transform, and the reversed search direction."""
    from iterativeclosestpoint_amd import vtkPolyData, vtkPolyDataCorrespondenceFilter as F
    xyz, tri = synth.bumpy_torus(96, 48)
    T = synth.rigid(ry=4.0, t=(0.02, 0.0, -0.01))
    moved = synth.apply(T, xyz)
    mesh_t = oracle.Mesh(xyz, tri)
    f = F.New()
    f.SetInputData(0, vtkPolyData(moved, tri))
    f.SetInputData(1, vtkPolyData(xyz, tri))
    f.SetMaximumNumberOfLandmarks(1000)                        # step = 4608 // 1000 = 4
    f.SetMaximumDistance(0.05)
    f.Update()
    ref = mesh_t.correspond(moved, step=len(moved) // 1000, max_dist=0.05, grid=True)
    assert 0 < len(ref["cell"]) < len(moved[::4])              # the threshold rejects some
    assert np.array_equal(f.GetOutput(0).GetPoints(), ref["origin"])
    assert np.array_equal(f.GetOutput(1).GetPoints(), ref["closest"])
    assert np.abs(f.GetOutput(1).GetPointNormals() - ref["normal"]).max() <= 1e-6
    assert f.GetOutput(0).GetPointNormals() is None
    assert np.array_equal(f.Cells, ref["cell"]) and np.array_equal(f.SourceIndices, ref["src_index"])
    # with the inverse transform and origin normals; a MinNormalDot that rejects some rows
    Tinv = np.linalg.inv(T)
    f.SetTransform(Tinv)
    f.IncludeOriginNormalsOn()
    f.SetMinNormalDot(0.9999)
    f.SetMaximumNumberOfLandmarks(10 ** 9)
    f.Update()
    sn = oracle.point_normals(moved, tri)
    ref = mesh_t.correspond(moved, T=Tinv, step=1, max_dist=0.05, src_normals=sn, min_dot=0.9999,
                            grid=True)
    assert 0 < len(ref["cell"]) < len(moved)                   # the normals are not rotated (Q7)
    assert np.array_equal(f.GetOutput(0).GetPoints(), ref["origin"])
    assert np.array_equal(f.GetOutput(0).GetPointNormals(), ref["origin_normal"])
    assert np.array_equal(f.GetOutput(1).GetPoints(), ref["closest"])
    # reversed direction: port 1's points are searched against port 0's mesh
    g = F.New()
    g.SetInputData(0, vtkPolyData(moved, tri))
    g.SetInputData(1, vtkPolyData(xyz, tri))
    g.SetTargetToSource()
    g.Update()
    ref = oracle.Mesh(moved, tri).correspond(xyz, step=1, max_dist=10.0, grid=True)
    assert np.array_equal(g.GetOutput(0).GetPoints(), ref["origin"])
    assert np.array_equal(g.GetOutput(1).GetPoints(), ref["closest"])
    assert np.array_equal(g.Cells, ref["cell"])
    # double output precision: the rows as computed, unrounded; no origin normals on that path
    f.SetOutputPrecision(F.DOUBLE_PRECISION)
    f.Update()
    ref = mesh_t.correspond_f64(moved, T=Tinv, step=1, max_dist=0.05, grid=True)
    out0, out1 = f.GetOutput(0), f.GetOutput(1)
    assert out0.GetPoints().dtype == np.float64 and out1.GetPointNormals().dtype == np.float64
    assert len(ref["cell"]) > 0 and np.array_equal(f.Cells, ref["cell"])
    assert np.array_equal(out0.GetPoints(), ref["origin"])
    assert np.array_equal(out1.GetPoints(), ref["closest"])
    assert np.array_equal(f.Distances, ref["dist"])
    assert np.abs(out1.GetPointNormals() - ref["normal"]).max() <= 1e-14
    assert out0.GetPointNormals() is None


def test_double_rows_on_edges_and_vertices(icpx, synth, oracle, small):
    """icpx_correspond_f64 where the closest point falls on faces, edges and vertices (the
    edge and vertex normals are summed from the fp64 face normals on the spot), with
    caller-supplied cell normals as well."""
    c, mesh = small
    q = queries_around(synth, c["target_xyz"], c["target_tri"], 3000, seed=5)
    ref = mesh.correspond_f64(q, step=1, max_dist=1e30, grid=False)
    kinds = mesh.correspond(q, step=1, max_dist=1e30, grid=False)["kind"]
    assert {0, 1, 2} <= set(np.unique(kinds))
    ctx = icpx.Context(0)
    ctx.set_target(c["target_xyz"], c["target_tri"])
    ctx.set_source(q)
    got = ctx.correspond_f64(np.eye(4), step=1, max_dist=1e30)
    for k in ("cell", "src_index", "origin", "closest", "dist"):
        assert np.array_equal(got[k], ref[k]), k
    assert np.abs(got["normal"] - ref["normal"]).max() <= 1e-14
    cn = synth.gaussian(3, 7, 3 * len(c["target_tri"])).reshape(-1, 3).astype(np.float32)
    cn /= np.linalg.norm(cn, axis=1, keepdims=True)
    ctx.set_target(c["target_xyz"], c["target_tri"], cn)
    ref = oracle.Mesh(c["target_xyz"], c["target_tri"], cn).correspond_f64(
        q, T=synth.rigid(rz=0.2), step=3, max_dist=0.5, grid=False)
    got = ctx.correspond_f64(synth.rigid(rz=0.2), step=3, max_dist=0.5)
    assert 0 < len(ref["cell"]) and np.array_equal(got["cell"], ref["cell"])
    assert np.array_equal(got["closest"], ref["closest"])
    assert np.abs(got["normal"] - ref["normal"]).max() <= 1e-14


def test_golden_fixtures(icpx, synth):
    """The committed oracle outputs (tests/golden) reproduced on the GPU."""
    import os
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    g = np.load(os.path.join(gold, "c1_oracle.npz"))
    c = synth.config_c1()
    sn = synth.vertex_normals(c["source_xyz"], c["target_tri"])
    ctx = icpx.Context(0)
    ctx.set_target(c["target_xyz"], c["target_tri"])
    ctx.set_source(c["source_xyz"], sn)
    got = ctx.correspond(g["T"], step=int(g["step"]), max_dist=float(g["max_dist"]), min_dot=0.0,
                         use_source_normals=True)
    ref = {k[5:]: g[k] for k in g.files if k.startswith("rows_")}
    check_rows(got, ref)
    for metric in (0, 1, 2):
        sums = ctx.accumulate(metric, g["T"], step=int(g["step"]), max_dist=float(g["max_dist"]),
                              use_source_normals=True)
        M, n, md, rc = icpx.solve(metric, sums, ctx.origin)
        assert rc == 0 and rel_err(M, g["solve_%d" % metric]) <= 1e-9
        p = icpx.default_params()
        p.metric, p.max_iterations, p.max_landmarks = metric, 50, 100
        p.max_distance = 10.0 if metric == 0 else 0.25 * synth.diag(c["target_xyz"])
        T, it, mean, st = ctx.run(p)
        assert it == int(g["icp_%d_iters" % metric]) and rel_err(T, g["icp_%d" % metric]) <= 1e-6
    cube = np.load(os.path.join(gold, "cube_oracle.npz"))
    ctx = icpx.Context(0)
    ctx.set_target(cube["xyz"], cube["tri"])
    ctx.set_source(cube["q"])
    got = ctx.correspond(np.eye(4), max_dist=100.0)
    for k in ("cell", "dist", "closest"):
        assert np.array_equal(got[k], cube["rows_" + k]), k
    assert np.abs(got["normal"] - cube["rows_normal"]).max() <= 1e-6


def test_degenerate_and_duplicate_triangles(icpx, synth, oracle):
    """Zero-area triangles are skipped (as the locator skips them); duplicated triangles tie
    exactly and the smaller cell id wins; unwelded seams tie across different point ids."""
    xyz, tri = synth.bumpy_torus(24, 12)
    nv = len(xyz)
    xyz2 = np.concatenate([xyz, xyz[:50]])                      # unwelded copies of 50 points
    extra = np.array([[0, 0, 1], [5, 5, 5], [3, 4, 3]], np.int32)  # zero-area
    dup = tri[:40].copy()                                       # exact duplicates (ids nt+3..)
    seam = tri[40:80].copy()
    seam[seam < 50] += nv                                       # same geometry, other point ids
    tri2 = np.ascontiguousarray(np.concatenate([extra, tri, dup, seam]))
    mesh = oracle.Mesh(xyz2, tri2)
    q = queries_around(synth, xyz, tri, 2000, seed=77)
    for width, exhaustive in ((8, False), (4, False), (64, False), (8, True), (1, False)):
        ctx = icpx.Context(0)
        ctx.set_group_width(width)
        ctx.set_exhaustive(exhaustive)
        ctx.set_target(xyz2, tri2)
        ctx.set_source(q)
        check_rows(ctx.correspond(np.eye(4), max_dist=1e30),
                   mesh.correspond(q, max_dist=1e30, grid=False))


def test_many_way_ties(icpx, synth, oracle):
    """A 40-triangle fan: above the apex every triangle is exactly equidistant, which is
    more contenders than the kernels' tie rings hold; the smallest cell id must win."""
    k = 40
    ang = np.arange(k) * (2 * np.pi / k)
    xyz = np.concatenate([[[0, 0, 1]], np.stack([np.cos(ang), np.sin(ang), 0 * ang], -1)])
    xyz = xyz.astype(np.float32)
    tri = np.array([[0, 1 + i, 1 + (i + 1) % k] for i in range(k)], np.int32)
    tri = np.roll(tri, 7, axis=0)                     # winner is not simply the first visited
    q = np.array([[0, 0, 2], [0, 0, 1], [0, 0, 1.5], [0.01, 0, 2], [0.5, 0.5, 0.5], [2, 0, 0],
                  [0, 0, -1], [1, 0, 0]], np.float32)
    q = np.concatenate([q, queries_around(synth, xyz, tri, 300, seed=9)])
    mesh = oracle.Mesh(xyz, tri)
    ref = mesh.correspond(q, max_dist=1e30, grid=False)
    assert ref["cell"][0] == 0 and ref["kind"][0] == 2
    for width in (1, 4, 8, 64):
        ctx = icpx.Context(0)
        ctx.set_group_width(width)
        ctx.set_target(xyz, tri)
        ctx.set_source(q)
        check_rows(ctx.correspond(np.eye(4), max_dist=1e30), ref)
        check_rows(ctx.correspond(np.eye(4), max_dist=1e30), ref)   # hinted pass


def test_random_tiny_meshes_with_duplicates(icpx, oracle):
    """Hundreds of tiny random triangle soups (shared points, exact duplicate and permuted
    triangles, queries on the points themselves): every kernel variant returns the oracle's
    (dist2, cell id) minimum."""
    rng = np.random.default_rng(1)
    for trial in range(120):
        nvp = rng.integers(4, 9)
        xyz = rng.uniform(-1, 1, (nvp, 3)).astype(np.float32)
        ntr = rng.integers(2, 9)
        tri = np.array([rng.choice(nvp, 3, replace=False) for _ in range(ntr)], np.int32)
        ndup = rng.integers(0, 4)
        if ndup:
            tri = np.concatenate([tri, tri[rng.integers(0, ntr, ndup)]])
        tri = np.ascontiguousarray(tri[rng.permutation(len(tri))])
        q = np.concatenate([xyz, rng.uniform(-1.5, 1.5, (6, 3)).astype(np.float32)])
        ref = oracle.Mesh(xyz, tri).correspond(q, max_dist=1e30, grid=False)
        for width, leaf in ((1, 1), (1, 2), (1, 4), (8, 2), (64, 2)):
            if trial % 3 and width != 1:
                continue
            ctx = icpx.Context(0)
            ctx.set_group_width(width)
            ctx.set_leaf_size(leaf)
            ctx.set_target(xyz, tri)
            ctx.set_source(q)
            got = ctx.correspond(np.eye(4), max_dist=1e30)
            assert np.array_equal(got["cell"], ref["cell"]), (trial, width, leaf)
            assert np.array_equal(got["dist"], ref["dist"])
            ctx.close()


def test_hinted_shortcut_matches_tree_walk(icpx, synth, oracle):
    """Lane-per-query path over a few ICP-like passes: with the neighbour-list shortcut and
    pooled lanes on, the sums are bit-identical to the plain tree walk's, and the rows match
    the oracle.  Residuals from large to tiny exercise list rebuilds and the fallback."""
    xyz, tri = synth.bumpy_torus(300, 200)
    pts, f, fn = synth.sample_surface(xyz, tri, 60_000, seed=21)
    noise = np.stack([synth.gaussian(21, 70 + 2 * k, len(pts)) for k in range(3)], -1) * 2e-4
    far = synth.u01(21, 90, len(pts)) < 0.005                    # a few far outliers
    noise[far] *= 100.0
    T0 = synth.rigid(ry=3.0, t=(0.02, -0.01, 0.015))
    src = synth.apply(T0, pts.astype(np.float64) + noise)
    Tinv = np.linalg.inv(T0)
    seq = [np.eye(4), np.eye(4), 0.5 * (np.eye(4) + Tinv), Tinv, Tinv,
           Tinv @ synth.rigid(rz=0.01), Tinv]
    ctxs = []
    for local, persistent in ((True, 1), (False, 3), (False, 1)):
        c = icpx.Context(0)
        c.set_group_width(1)
        c.set_local(local)
        c.set_persistent(persistent)  # 1 pooled lanes, 3 one walk per wavefront
        c.set_target(xyz, tri)
        c.set_source(src)
        ctxs.append(c)
    for i, T in enumerate(seq):
        if i == 3:
            ctxs[0].set_local(True)  # two far passes in a row suspend the shortcut: start afresh
        sums = [c.accumulate(1, T, step=1, max_dist=0.05) for c in ctxs]
        assert np.array_equal(sums[0], sums[1]) and np.array_equal(sums[1], sums[2])
    st = ctxs[0].debug_stats()
    assert st["local_passes"] >= 4 and st["nbr_builds"] >= 1, st
    assert ctxs[1].debug_stats()["local_passes"] == 0
    mesh = oracle.Mesh(xyz, tri)
    ref = mesh.correspond(src, T=Tinv, step=1, max_dist=0.05, grid=True)
    for c in ctxs:
        check_rows(c.correspond(Tinv, step=1, max_dist=0.05), ref)
    # clean samples exactly on the surface: residual ~1e-8, the smallest list radius
    src2 = synth.apply(T0, pts)
    c = ctxs[0]
    c.set_source(src2)
    for T in (np.eye(4), Tinv, Tinv, Tinv):
        c.accumulate(1, T, step=1, max_dist=0.05)
    check_rows(c.correspond(Tinv, step=1, max_dist=0.05),
               mesh.correspond(src2, T=Tinv, step=1, max_dist=0.05, grid=True))
    assert c.debug_stats()["local_passes"] > st["local_passes"]


def test_full_size_c3(icpx, synth, oracle):
    """The benchmark's own workload (1M triangles, 1M points, symmetric metric): rows of a
    20 000-point sample against the oracle, properties of the whole pass, and the registration."""
    c = synth.config_c3()
    ctx = icpx.Context(0)
    ctx.set_target(c["target_xyz"], c["target_tri"])
    ctx.set_source(c["source_xyz"], c["source_normals"])
    rows = ctx.correspond(np.eye(4), step=1, max_dist=10.0, use_source_normals=True)
    n = len(c["source_xyz"])
    assert len(rows["cell"]) == n and np.array_equal(rows["src_index"], np.arange(n))
    pick = (np.arange(20_000) * 49).astype(np.int64)
    mesh = oracle.Mesh(c["target_xyz"], c["target_tri"])
    ref = mesh.correspond(c["source_xyz"][pick], step=1, max_dist=10.0,
                          src_normals=c["source_normals"][pick], grid=True)
    assert np.array_equal(rows["cell"][pick], ref["cell"])
    assert np.array_equal(rows["dist"][pick], ref["dist"])
    assert np.array_equal(rows["closest"][pick], ref["closest"])
    assert np.abs(rows["normal"][pick] - ref["normal"]).max() <= 1e-6
    # idempotence: the second pass starts from the first one's winners (shortcut and
    # surface walk) and must return the same rows
    again = ctx.correspond(np.eye(4), step=1, max_dist=10.0, use_source_normals=True)
    for k in ("cell", "dist", "closest", "normal"):
        assert np.array_equal(rows[k], again[k]), k
    # every emitted point lies on its triangle's plane to rounding, at the emitted distance
    P = c["target_xyz"].astype(np.float64)[c["target_tri"][rows["cell"]]]
    nrm = np.cross(P[:, 1] - P[:, 0], P[:, 2] - P[:, 0])
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    off = np.abs(np.einsum("ij,ij->i", rows["closest"].astype(np.float64) - P[:, 0], nrm))
    assert off.max() <= 1e-6
    d = np.linalg.norm(rows["origin"].astype(np.float64) - rows["closest"].astype(np.float64), axis=1)
    assert np.abs(d - rows["dist"]).max() <= 1e-6
    p = icpx.default_params()
    p.metric, p.max_iterations, p.max_landmarks, p.max_distance = 2, 30, n, 10.0
    T, it, mean, st = ctx.run(p)
    assert it == 30 and np.abs(T @ c["truth"] - np.eye(4)).max() < 1e-6 and mean < 1e-6
    # the benchmark's registration (50 iterations) with lean passes off and on: the same bits
    p.max_iterations = 50
    res = {}
    for mode in (0, 1):
        ctx.set_lean(mode)
        before = ctx.debug_stats()["lean_passes"]
        res[mode] = ctx.run(p) + (ctx.debug_stats()["lean_passes"] - before,)
    assert np.array_equal(res[0][0], res[1][0]) and res[0][1:4] == res[1][1:4]
    assert res[0][4] == 0 and res[1][4] >= 35, (res[0][4], res[1][4])


def test_large_target_c4_sample(icpx, synth, oracle):
    """The 10M-triangle target of the large configuration with a 200 000-point noisy scan:
    tree depth, index widths and the shortcut's policy at scale; 5 000 rows against the oracle."""
    xyz, tri = synth.bumpy_torus(2500, 2000)
    pts, f, fn = synth.sample_surface(xyz, tri, 200_000, seed=5)
    d = synth.diag(xyz)
    noise = np.stack([synth.gaussian(5, 32 + 2 * k, len(pts)) for k in range(3)], -1) * (1e-3 * d)
    M = synth.rigid(ry=5.0, t=(0.01 * d / np.sqrt(3),) * 3)
    src = synth.apply(M, pts.astype(np.float64) + noise)
    ctx = icpx.Context(0)
    ctx.set_group_width(1)  # the lane-per-query path, as 16M points would take
    ctx.set_target(xyz, tri)
    ctx.set_source(src)
    p = icpx.default_params()
    p.metric, p.max_iterations, p.max_landmarks, p.max_distance = 1, 12, len(src), 10.0
    T, it, mean, st = ctx.run(p)
    assert it == 12 and np.abs(T @ M - np.eye(4)).max() < 2e-4  # noise floor of 200k samples
    rows = ctx.correspond(T, step=1, max_dist=10.0)
    pick = (np.arange(5_000) * 37).astype(np.int64)
    mesh = oracle.Mesh(xyz, tri)
    ref = mesh.correspond(src[pick], T=T, step=1, max_dist=10.0, grid=True)
    for k in ("cell", "dist", "closest"):
        assert np.array_equal(rows[k][pick], ref[k]), k
    assert np.abs(rows["normal"][pick] - ref["normal"]).max() <= 1e-6


def test_full_size_c4(icpx, synth, oracle):
    """The target run itself: 10M triangles, 16M noisy points, point-to-plane, through the
    default kernels (no knobs).  (1) The 4x4 after two iterations and every 3200th row of the
    first two passes against golden vectors the CPU oracle produced on the same inputs
    (tools/make_c4_golden.py).  (2) The 50-iteration registration against the ground truth.
    (3) The light certificate kernel, and both tree kernels enqueued over a large cloud's
    leftovers, did run (the wavefront walk over a list in a pass after a move that breaks most
    certificates).  (4) At the converged pose: a 5000-row sample against the oracle run
    here, and properties of all 16M rows."""
    gold = np.load(os.path.join(os.path.dirname(__file__), "golden", "c4_two_iterations.npz"))
    c = synth.config_c4()
    n = len(c["source_xyz"])
    assert n == 16_000_000 and len(c["target_tri"]) == 10_000_000
    ctx = icpx.Context(0)
    ctx.set_target(c["target_xyz"], c["target_tri"])
    ctx.set_source(c["source_xyz"])
    p = icpx.default_params()
    p.metric, p.max_landmarks, p.max_distance = 1, n, c["max_dist"]
    p.max_iterations = 2
    T2, it, mean, st = ctx.run(p)
    assert it == 2 and rel_err(T2, gold["T2"]) <= 1e-9
    p.max_iterations = 1
    assert rel_err(ctx.run(p)[0], gold["T1"]) <= 1e-9
    stride = int(gold["sample_stride"])
    for k in (0, 1):
        rows = ctx.correspond(gold["pose%d" % k], step=1, max_dist=c["max_dist"])
        assert len(rows["cell"]) == int(gold["n%d" % k]) == n
        pick = gold["src_index%d" % k]
        assert np.array_equal(pick, np.arange(0, n, stride))
        for key in ("cell", "dist", "closest"):
            assert np.array_equal(rows[key][pick], gold["%s%d" % (key, k)]), (k, key)
        assert np.abs(rows["normal"][pick] - gold["normal%d" % k]).max() <= 1e-6
        d = np.linalg.norm(rows["origin"].astype(np.float64) - rows["closest"], axis=1)
        assert abs(d.mean() - float(gold["mean%d" % k])) <= 1e-12
        del rows, d
    p.max_iterations = 50
    T, it, mean, st = ctx.run(p)
    assert it == 50 and st == icpx.RUN_MAX_ITERATIONS
    assert np.abs(T @ c["truth"] - np.eye(4)).max() < 3e-5   # the noise floor of the scan
    stats = ctx.debug_stats()
    assert stats["certify_passes"] >= 40, stats   # certificates checked by the light kernel
    assert stats["split_pool"] >= 30, stats
    # (noisy points are beyond the neighbour lists' reach: after two listed passes that showed
    # it, the light kernel's list went straight to the tree kernels)
    assert stats["direct_passes"] >= 30, stats
    # A move of a sixteenth of an edge length breaks most certificates: the light kernel's list
    # is then most of the cloud and goes to the wavefront walk, not to the pooled lanes.
    moved = synth.rigid(t=(1e-4, 0.0, 0.0)) @ T
    rows = ctx.correspond(moved, step=1, max_dist=c["max_dist"])
    after = ctx.debug_stats()
    assert after["certify_passes"] == stats["certify_passes"] + 1, (stats, after)
    assert after["split_passes"] == stats["split_passes"] + 1, (stats, after)
    # (the list's length: on the wavefront walk's side of the split, so that kernel served it)
    assert ctx.debug_counters()["fallback"] >= n // 2, ctx.debug_counters()
    pick = (np.arange(2_000) * 7919 + 5).astype(np.int64)
    mesh = oracle.Mesh(c["target_xyz"], c["target_tri"])
    ref = mesh.correspond(c["source_xyz"][pick], T=moved, step=1, max_dist=c["max_dist"], grid=True)
    assert np.array_equal(rows["src_index"], np.arange(n))
    for key in ("cell", "dist", "closest"):
        assert np.array_equal(rows[key][pick], ref[key]), key
    del rows
    rows = ctx.correspond(T, step=1, max_dist=c["max_dist"])
    assert len(rows["cell"]) == n and np.array_equal(rows["src_index"], np.arange(n))
    pick = (np.arange(5_000) * 3137 + 11).astype(np.int64)
    ref = mesh.correspond(c["source_xyz"][pick], T=T, step=1, max_dist=c["max_dist"], grid=True)
    for key in ("cell", "dist", "closest"):
        assert np.array_equal(rows[key][pick], ref[key]), key
    assert np.abs(rows["normal"][pick] - ref["normal"]).max() <= 1e-6
    # every emitted point lies on its triangle's plane to rounding, at the emitted distance
    P0 = c["target_xyz"][c["target_tri"][rows["cell"], 0]].astype(np.float64)
    P1 = c["target_xyz"][c["target_tri"][rows["cell"], 1]].astype(np.float64)
    P2 = c["target_xyz"][c["target_tri"][rows["cell"], 2]].astype(np.float64)
    nrm = np.cross(P1 - P0, P2 - P0)
    del P1, P2
    nrm /= np.linalg.norm(nrm, axis=1, keepdims=True)
    off = np.abs(np.einsum("ij,ij->i", rows["closest"].astype(np.float64) - P0, nrm))
    assert off.max() <= 1e-6
    del P0, nrm, off
    d = np.linalg.norm(rows["origin"].astype(np.float64) - rows["closest"].astype(np.float64), axis=1)
    assert np.abs(d - rows["dist"]).max() <= 1e-6
    assert abs(d.mean() - mean) <= 1e-6            # the loop's last mean distance is one solve old
    del rows, d
    # (5) The whole loop against the oracle on this geometry: MaximumNumberOfLandmarks = 1e6, so
    # the filter's stride rule keeps every 16th of the 16M points
    # (ICP/vtkPolyDataCorrespondenceFilter.cxx:473-477), 50 point-to-plane iterations
    # (ICP/vtkICP.cxx:189-246).  The golden 4x4 after every iteration comes from the CPU oracle
    # (tools/make_c4_loop_golden.py); the north star's gate is 1e-6 relative on the final one.
    loop = np.load(os.path.join(os.path.dirname(__file__), "golden", "c4_loop_1m_landmarks.npz"))
    assert int(loop["landmarks"]) == 1_000_000 and int(loop["step"]) == 16 and loop["T"].shape == (50, 4, 4)
    p.max_landmarks = int(loop["landmarks"])
    gaps = {}
    for k in (1, 2, 3, 5, 10, 20, 50):
        p.max_iterations = k
        Tk, it, mean_k, st = ctx.run(p)
        assert it == k
        gaps[k] = rel_err(Tk, loop["T"][k - 1])
        # (the library's mean distance is that of the last pass searched: the oracle's pass k)
        assert abs(mean_k - float(loop["mean"][k - 1])) <= 1e-9, (k, mean_k, float(loop["mean"][k - 1]))
    print("50-iteration loop, 1e6 landmarks on the 10M-triangle target, relative gap to the "
          "oracle's 4x4 after k iterations:", {k: "%.2e" % v for k, v in gaps.items()})
    # (the first solve sees the same rows: rounding alone.  From the second on, the two
    # transforms differ in their last bits, a few of a million noisy points lie within that of
    # an edge between two triangles and take the other one's normal: 1e-9 of the sums)
    assert gaps[1] <= 1e-12 and max(gaps[k] for k in (2, 3)) <= 1e-8, gaps
    assert max(gaps.values()) <= 1e-6, gaps
    assert np.all(loop["n"] == 1_000_000)


def test_full_size_c5(icpx, synth, oracle):
    """The partial-overlap, outlier-heavy pair at the benchmark's size (0.59M triangles, 0.84M
    points of which a quarter end up within the threshold): rows at the start pose and at the
    registered pose against the oracle (a 6000-point sample each, accepted and rejected alike),
    properties of every accepted row, and the same registration with lean passes off and on and
    with the noisy cloud's policy off and on: identical bits."""
    c = synth.config_c5(nu=1000, nv=500, npts=1_200_000)
    n = len(c["source_xyz"])
    ctx = icpx.Context(0)
    ctx.set_target(c["target_xyz"], c["target_tri"])
    ctx.set_source(c["source_xyz"])
    mesh = oracle.Mesh(c["target_xyz"], c["target_tri"])
    p = icpx.default_params()
    p.metric, p.max_iterations, p.max_landmarks, p.max_distance = c["metric"], 50, n, c["max_dist"]
    res = {}
    for name, lean, noisy in (("default", 1, 1), ("no lean passes", 0, 1), ("policy by size alone", 1, 0)):
        ctx.set_lean(lean)
        ctx.set_noisy_policy(noisy)
        res[name] = ctx.run(p)
    ctx.set_lean(1)
    ctx.set_noisy_policy(1)
    T, it, mean, st = res["default"]
    assert it == 50 and st == icpx.RUN_MAX_ITERATIONS
    for name, r in res.items():
        assert np.array_equal(r[0], T) and r[1:] == res["default"][1:], name
    pick = (np.arange(6_000) * 139 + 7).astype(np.int64)
    for pose in (np.eye(4), T):
        rows = ctx.correspond(pose, step=1, max_dist=c["max_dist"])
        assert 0.15 * n < len(rows["cell"]) < 0.5 * n          # the threshold drops most
        ref = mesh.correspond(c["source_xyz"][pick], T=pose, step=1, max_dist=c["max_dist"], grid=True)
        at = np.searchsorted(rows["src_index"], pick[ref["src_index"]])
        assert np.array_equal(rows["src_index"][at], pick[ref["src_index"]])
        # (the sample's accepted rows are exactly those of the library's that fall in the sample)
        assert np.isin(rows["src_index"], pick).sum() == len(ref["cell"])
        for key in ("cell", "dist", "closest"):
            assert np.array_equal(rows[key][at], ref[key]), key
        assert np.abs(rows["normal"][at] - ref["normal"]).max() <= 1e-6
        assert rows["dist"].max() < c["max_dist"]
        d = np.linalg.norm(rows["origin"].astype(np.float64) - rows["closest"].astype(np.float64), axis=1)
        assert np.abs(d - rows["dist"]).max() <= 1e-6
    # too small a threshold: nothing is accepted and the run stops for want of rows (the
    # reference has no such handling: ICP/icp/plane2plane.cxx:41-42 takes the mean of nothing)
    p.max_distance = 1e-9
    Tn, itn, _, stn = ctx.run(p)
    assert stn == icpx.RUN_TOO_FEW and itn == 0 and np.array_equal(Tn, np.eye(4))


def test_device_loop_with_rccl_collective(icpx, synth, small):
    """The multi-GPU driver's loop on one rank: search -> all-reduce of the sums through
    the nccl (RCCL) backend -> solve, all ordered on torch's stream, against icpx_run."""
    import torch
    import torch.distributed as dist
    from iterativeclosestpoint_amd.distributed import ShardedICP

    c, mesh = small
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    created = not dist.is_initialized()
    if created:
        dist.init_process_group("nccl", rank=0, world_size=1,
                                device_id=torch.device("cuda", 0))
    try:
        sh = ShardedICP(device=0, rank=0, world=1)
        sh.set_target(c["target_xyz"], c["target_tri"])
        sh.set_source(c["source_xyz"])
        p = icpx.default_params()
        p.metric, p.max_iterations, p.max_landmarks = 1, 40, 1000
        p.max_distance = c2_threshold(synth, c)
        p.check_mean_distance, p.max_mean_distance = 1, 1e-3
        Tc, itc, mc, stc = sh.run_device(p, force_collective=True)
        ctx = icpx.Context(0)
        ctx.set_target(c["target_xyz"], c["target_tri"])
        ctx.set_source(c["source_xyz"])
        T, it, m, st = ctx.run(p)
        assert (itc, stc) == (it, st) and st == 1
        assert np.array_equal(Tc, T) and mc == m
    finally:
        if created:
            dist.destroy_process_group()


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_shortcut_on_triangle_soup(icpx, synth, oracle, seed):
    """Hinted passes (neighbour lists, surface walk, fallback, ties) on a mesh that is nothing
    like a smooth manifold: overlapping random triangles of very different sizes, slivers,
    duplicates and a flat fan, with queries on, near and far from it, under a drifting
    transform.  Every pass's rows equal the oracle's."""
    n = 6000
    c = (np.stack([synth.u01(seed, 10 + k, n) for k in range(3)], -1) - 0.5) * 2.0
    size = 0.002 + 0.2 * synth.u01(seed, 20, n) ** 4                 # sizes over two decades
    e1 = (np.stack([synth.u01(seed, 30 + k, n) for k in range(3)], -1) - 0.5) * size[:, None]
    e2 = (np.stack([synth.u01(seed, 40 + k, n) for k in range(3)], -1) - 0.5) * size[:, None]
    sliver = synth.u01(seed, 50, n) < 0.1
    e2[sliver] = e1[sliver] * 0.5 + e2[sliver] * 1e-4                # slivers
    P = np.stack([c, c + e1, c + e2], 1)
    fan_c = np.array([0.1, 0.2, 0.3])
    ang = np.linspace(0, 2 * np.pi, 25)
    fan = np.stack([np.repeat(fan_c[None], 24, 0),
                    fan_c + 0.05 * np.stack([np.cos(ang[:-1]), np.sin(ang[:-1]), 0 * ang[:-1]], -1),
                    fan_c + 0.05 * np.stack([np.cos(ang[1:]), np.sin(ang[1:]), 0 * ang[1:]], -1)], 1)
    P = np.concatenate([P, fan, P[:200]], 0)                         # and 200 duplicates
    xyz = np.ascontiguousarray(P.reshape(-1, 3), np.float32)
    tri = np.arange(len(xyz), dtype=np.int32).reshape(-1, 3)
    mesh = oracle.Mesh(xyz, tri)
    nq = 20000
    t = (synth.u01(seed, 60, nq) * len(tri)).astype(np.int64) % len(tri)
    w = np.stack([synth.u01(seed, 61 + k, nq) for k in range(3)], -1) + 1e-3
    w /= w.sum(1, keepdims=True)
    on = np.einsum("ij,ijk->ik", w, xyz[tri[t]].astype(np.float64))
    off = np.stack([synth.gaussian(seed, 70 + 2 * k, nq) for k in range(3)], -1)
    scale = np.where(synth.u01(seed, 80, nq) < 0.5, 0.0, 10.0 ** (-6 + 5 * synth.u01(seed, 81, nq)))
    q = on + off * scale[:, None]
    q[:50] = fan_c + np.array([0, 0, 1.0]) * np.linspace(1e-6, 0.2, 50)[:, None]  # above the fan
    q = q.astype(np.float32)
    ctxs = []
    # shortcut with certificates, tree only, shortcut without certificates, shortcut with the
    # certificates checked by their own kernel (what large clouds get)
    modes = (1, 0, 2, 3)
    for local in modes:
        c_ = icpx.Context(0)
        c_.set_group_width(1)
        c_.set_local(local)
        c_.set_target(xyz, tri)
        c_.set_source(q)
        ctxs.append(c_)
    Ts = [np.eye(4), np.eye(4), synth.rigid(rz=0.02, t=(1e-4, 0, 0)), synth.rigid(rz=0.02, t=(1e-4, 0, 0)),
          synth.rigid(rx=0.5, t=(0.003, -0.002, 0.001)), np.eye(4), np.eye(4)]
    # then a slow drift, the regime in which certificates let winners stand: here they must
    # do so only where a search would agree, on slivers, duplicates and the fan as well
    for k, m in enumerate((1e-9, 1e-8, 1e-7, 3e-7, 1e-6, 1e-8, 1e-5, 1e-8, 1e-4, 1e-8)):
        Ts.append(synth.rigid(rx=10 * m, ry=-7 * m, t=(m, 0.5 * m, -m)) @ Ts[-1])
    for i, T in enumerate(Ts):
        ref = mesh.correspond(q, T=T, step=1, max_dist=10.0, grid=True)
        for c_, local in zip(ctxs, modes):
            if i in (1, 5):
                c_.set_local(local)  # (restarts the shortcut's policy after far passes)
            check_rows(c_.correspond(T, step=1, max_dist=10.0), ref)
            c_.accumulate(1, T, step=1, max_dist=10.0)  # a second hinted pass at the same pose
    assert ctxs[0].debug_stats()["local_passes"] >= 20
    assert (ctxs[0].debug_certificates(nq) > 0).mean() > 0.3


def test_certificates_change_nothing(icpx, synth, oracle):
    """A drifting registration-like sequence of poses, from moves far below the triangle size
    to moves beyond it: the shortcut with certificates (winners kept without a search while
    the accumulated move stays inside the certified slack), the shortcut without them and
    the tree alone return bit-identical rows in every pass, and those are the oracle's."""
    xyz, tri = synth.bumpy_torus(240, 120)
    pts, f, fn = synth.sample_surface(xyz, tri, 70_000, seed=21)
    off = np.stack([synth.gaussian(5, 90 + 2 * k, len(pts)) for k in range(3)], -1)
    scale = np.where(synth.u01(5, 96, len(pts)) < 0.7, 0.0, 10.0 ** (-7 + 4 * synth.u01(5, 97, len(pts))))
    src = (pts.astype(np.float64) + off * scale[:, None]).astype(np.float32)
    mesh = oracle.Mesh(xyz, tri)
    ctxs = []
    for local in (1, 2, 0, 3):  # (3: the certificates checked by their own kernel)
        c_ = icpx.Context(0)
        c_.set_group_width(1)
        c_.set_local(local)
        c_.set_target(xyz, tri)
        c_.set_source(src)
        ctxs.append(c_)
    moves = [0, 0, 1e-8, 3e-8, 1e-7, 1e-7, 1e-6, 1e-8, 1e-5, 1e-8, 1e-4, 1e-8, 1e-8, 2e-3, 1e-8, 1e-8]
    T = np.eye(4)
    kept = []
    for i, m in enumerate(moves):
        T = synth.rigid(rx=20 * m, rz=-30 * m, t=(m, -0.5 * m, 0.25 * m)) @ T
        rows = [c_.correspond(T, step=1, max_dist=10.0) for c_ in ctxs]
        check_rows(rows[0], rows[1])
        check_rows(rows[0], rows[2])
        check_rows(rows[0], rows[3])
        if i in (0, 3, 7, 11, 15):
            check_rows(rows[0], mesh.correspond(src, T=T, step=1, max_dist=10.0, grid=True))
        sums = [c_.accumulate(1, T, step=1, max_dist=10.0) for c_ in ctxs]  # and a fused pass
        assert np.array_equal(sums[0], sums[1]) and np.array_equal(sums[0], sums[2])
        assert np.array_equal(sums[0], sums[3])
        kept.append(float((ctxs[0].debug_certificates(len(src)) > 0).mean()))
        # (with the certificates checked by their own kernel a long list of leftovers goes to
        # another tree kernel, whose certificates are as sound but not the same numbers)
        kept3 = float((ctxs[3].debug_certificates(len(src)) > 0).mean())
        assert abs(kept3 - kept[-1]) < 0.05
    assert ctxs[0].debug_stats()["local_passes"] >= 20
    assert min(kept[2:8]) > 0.9 and kept[-1] > 0.9  # most queries hold a certificate


def test_certified_sets_of_exact_ties_weighed_by_k_contested(icpx, synth, small):
    """Points that sit exactly on the mesh's vertices and on midpoints of its edges tie between
    two to eight triangles for good: they stand on set certificates, and every settled pass
    weighs their sets again, dozens of them in every 512 consecutive queries (the path of
    k_contested that lists the marked queries and deals them out a lane apiece, where cell
    ids are read only when two squared distances are equal).  Rows equal the oracle's in every
    pass, whoever weighs (k_contested, the finalize, with and without leads)."""
    c, mesh = small
    xyz, tri = c["target_xyz"], c["target_tri"]
    e = np.concatenate([tri[:, [0, 1]], tri[:, [1, 2]], tri[:, [2, 0]]])
    mid = (0.5 * (xyz[e[:, 0]].astype(np.float64) + xyz[e[:, 1]])).astype(np.float32)
    pts, f, fn = synth.sample_surface(xyz, tri, 4000, seed=9)
    src = np.concatenate([xyz, mid[::3], pts])
    moves = [0, 0, 0, 1e-9, 1e-9, 3e-9, 0, 1e-8]
    base = None
    for weigh in (0, 1, 2, 3):
        ctx = icpx.Context(0)
        ctx.set_group_width(1)
        ctx.set_weigh(weigh)
        ctx.set_target(xyz, tri)
        ctx.set_source(src)
        T = np.eye(4)
        out = []
        for i, m in enumerate(moves):
            T = synth.rigid(rz=10 * m, t=(m, 0, -m)) @ T
            rows = ctx.correspond(T, step=1, max_dist=10.0)
            if weigh == 0 and i in (0, 2, 5, 7):
                check_rows(rows, mesh.correspond(src, T=T, step=1, max_dist=10.0, grid=True))
            out.append((rows, ctx.accumulate(1, T, step=1, max_dist=10.0)))
        bits = ctx.debug_certificates(len(src)).view(np.uint32)
        assert ((bits & 1) == 1).sum() > len(src) // 8   # the tied points stand on sets
        ctx.close()
        if base is None:
            base = out
        for (r, s_), (r0, s0) in zip(out, base):
            check_rows(r, r0)
            assert np.array_equal(s_, s0)


def test_scene_far_from_the_origin(icpx, synth, oracle):
    """Coordinates four hundred times the scene's extent: every rounding allowance is four
    hundred times what it is about the origin, the neighbour lists are not worth building, and a cloud of
    300 000 points is then served by the certificates' own kernel and the tree kernels.  Rows
    are the oracle's, cold, hinted and after moves; the registration is the oracle's loop's."""
    xyz, tri = synth.bumpy_torus(160, 80)
    shift = np.array([1000.0, -500.0, 250.0])
    xyz = (xyz.astype(np.float64) + shift).astype(np.float32)
    pts, f, fn = synth.sample_surface(xyz, tri, 300_000, seed=13)
    ctr = 0.5 * (xyz.max(0).astype(np.float64) + xyz.min(0))
    R = synth.rigid(ry=2.0, t=(0.01, -0.005, 0.008))
    M = np.eye(4)
    M[:3, :3] = R[:3, :3]
    M[:3, 3] = R[:3, 3] + ctr - R[:3, :3] @ ctr
    src = synth.apply(M, pts)
    mesh = oracle.Mesh(xyz, tri)
    ctx = icpx.Context(0)
    ctx.set_target(xyz, tri)
    ctx.set_source(src)
    T = np.eye(4)
    for m in (0, 0, 1e-4, 1e-2, 1e-4, 1e-4):
        T = synth.rigid(t=(m, -m, 0.5 * m)) @ T
        got = ctx.correspond(T, step=1, max_dist=10.0)
        pick = np.arange(0, len(src), 7)
        ref = mesh.correspond(src[pick], T=T, step=1, max_dist=10.0, grid=True)
        for k in ("cell", "dist", "closest"):
            assert np.array_equal(got[k][pick], ref[k]), (m, k)
    p = icpx.default_params()
    p.metric, p.max_iterations, p.max_landmarks, p.max_distance = 1, 20, len(src), 10.0
    Tg, it, mean, st = ctx.run(p)
    stats = ctx.debug_stats()
    assert stats["nbr_eps"] == 0.0 and stats["direct_passes"] > 0, stats
    To = mesh.icp(src, metric=1, max_iter=20, max_landmarks=len(src), max_dist=10.0)[0]
    # (a rotation that differs in its eighth digit moves a point a thousand units from the
    # origin by a float's last bit, which is what the rows are rounded to: the two loops agree
    # in where they put the cloud, to 1e-6 of the coordinates' size, not in the 4x4's entries)
    p64 = src.astype(np.float64)
    where_g = p64 @ Tg[:3, :3].T + Tg[:3, 3]
    where_o = p64 @ To[:3, :3].T + To[:3, 3]
    gap = float(np.abs(where_g - where_o).max() / np.abs(p64).max())
    print("scene far from the origin: the loops' clouds differ by %.2e of the coordinates' size" % gap)
    assert gap < 1e-6
    assert np.abs(where_g - pts.astype(np.float64)).max() < 2e-3   # ... and it registers
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("shape", ["two_sheets", "slivers", "flat"])
def test_near_coincident_sheets_slivers_and_a_flat_target(icpx, synth, oracle, shape):
    """Targets that stress one rule each.  two_sheets: a second copy of the surface 1e-5 away,
    so every query has two contenders that share no edge and differ in the sixth digit (pair
    and set certificates between triangles that are not neighbours).  slivers: every triangle
    thirty times as long as wide (boxes that overlap dozens of others).  flat: a planar square,
    a singular system for the point-to-plane solve, all boxes of zero height.  Rows are the
    oracle's on a cold pass, a hinted one and after small and large moves, on a noisy and a
    clean cloud; the registration is the oracle's loop's."""
    if shape == "two_sheets":
        xyz0, tri0 = synth.bumpy_torus(120, 60)
        xyz = np.concatenate([xyz0, xyz0 + np.float32([1e-5, 1e-5, 1e-5])])
        tri = np.concatenate([tri0, tri0 + len(xyz0)]).astype(np.int32)
    elif shape == "slivers":
        xyz, tri = synth.bumpy_torus(960, 30)
    else:
        n = 100
        g = np.linspace(0.0, 1.0, n)
        X, Y = np.meshgrid(g, g, indexing="ij")
        xyz = np.stack([X, Y, 0.0 * X], -1).reshape(-1, 3).astype(np.float32)
        i, j = np.meshgrid(np.arange(n - 1), np.arange(n - 1), indexing="ij")
        p00 = (i * n + j).ravel()
        tri = np.concatenate([np.stack([p00, p00 + n, p00 + n + 1], -1),
                              np.stack([p00, p00 + n + 1, p00 + 1], -1)]).astype(np.int32)
    pts, f, fn = synth.sample_surface(xyz, tri, 300_000, seed=29)
    noise = np.stack([synth.gaussian(31, 2 * k, len(pts)) for k in range(3)], -1) * 2e-3
    M = synth.rigid(ry=1.0 if shape != "flat" else 0.0, rx=1.0, t=(0.004, -0.002, 0.006))
    mesh = oracle.Mesh(xyz, tri)
    for label, cloud in (("noisy", pts.astype(np.float64) + noise), ("clean", pts.astype(np.float64))):
        src = synth.apply(M, cloud)
        ctx = icpx.Context(0)
        ctx.set_target(xyz, tri)
        ctx.set_source(src)
        T = np.eye(4)
        for m in (0, 0, 1e-5, 1e-5, 3e-3, 1e-5, 1e-6, 1e-6):
            T = synth.rigid(t=(m, -m, 0.5 * m)) @ T
            got = ctx.correspond(T, step=1, max_dist=10.0)
            pick = np.arange(0, len(src), 11)
            ref = mesh.correspond(src[pick], T=T, step=1, max_dist=10.0, grid=True)
            for k in ("cell", "dist", "closest"):
                assert np.array_equal(got[k][pick], ref[k]), (shape, label, m, k)
        if label == "clean":
            sub = src[::10]
            ctx.set_source(sub)
            p = icpx.default_params()
            p.metric, p.max_iterations, p.max_landmarks, p.max_distance = 1, 12, len(sub), 10.0
            Tg, it, mean, st = ctx.run(p)
            To, ito, meano = mesh.icp(sub, metric=1, max_iter=12, max_landmarks=len(sub), max_dist=10.0)[:3]
            assert it == ito
            # (a planar target leaves the in-plane motion undetermined: the solver's answer
            # there is whatever its pivoting makes of a singular system, equal to the oracle's
            # because the sums and the elimination order are)
            assert np.abs(Tg - To).max() < (1e-7 if shape == "flat" else 1e-9), np.abs(Tg - To).max()
        ctx.close()


@pytest.mark.gpu
def test_every_policy_gives_the_same_rows(icpx):
    """Random settings of the policy knobs (which kernel walks, lists, certificates, batches
    given up on, who weighs the sets, seeds, query order, lean and anchored passes) on a clean,
    a noisy, a graded and a doubled surface: rows equal to the plainest setting's bit for bit
    over twelve passes of moves from none to large, registrations equal to rounding."""
    import importlib.util
    spec = importlib.util.spec_from_file_location(
        "policy_fuzz", os.path.join(os.path.dirname(__file__), "..", "tools", "policy_fuzz.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.fuzz(8, 17, only=("noisy", "graded", "graded noisy", "two sheets")) is None


@pytest.mark.gpu
def test_inputs_that_are_not_numbers(icpx, synth, small):
    """The reference leaves these undefined (its locator derives a bucket index from the
    point).  Here a source coordinate that is not a number, or beyond 1e12, is refused and
    leaves the context without a source; so is a transform beyond the limits; and normals
    that are not numbers, which put a NaN into the sums and the solve, end the registration
    with RUN_DIVERGED and the last good transform, on the device's loop and the host's."""
    c, _ = small
    xyz, tri, src = c["target_xyz"], c["target_tri"], c["source_xyz"]
    ctx = icpx.Context(0)
    ctx.set_target(xyz, tri)
    p = icpx.default_params()
    p.metric, p.max_iterations, p.max_landmarks, p.max_distance = 1, 10, len(src), 10.0
    for bad in (np.nan, np.inf, -np.inf, 2e12):
        pts = src.copy()
        pts[len(pts) // 2, 1] = bad
        with pytest.raises(icpx.IcpxError) as e:
            ctx.set_source(pts)
        assert e.value.code == icpx.E_INVALID
        with pytest.raises(icpx.IcpxError):
            ctx.run(p)      # (no source)
    ctx.set_source(src)
    T = np.eye(4)
    T[0, 0] = 1e7
    with pytest.raises(icpx.IcpxError):
        ctx.correspond(T)
    T = np.eye(4)
    T[2, 3] = np.nan
    with pytest.raises(icpx.IcpxError):
        ctx.correspond(T)
    bad_xyz = xyz.copy()
    bad_xyz[3, 0] = np.nan
    with pytest.raises(icpx.IcpxError):
        icpx.Context(0).set_target(bad_xyz, tri)
    good = ctx.run(p)
    assert good[3] == icpx.RUN_MAX_ITERATIONS and np.isfinite(good[0]).all()
    # cell normals of the user's with a NaN among them: rows carry it into the sums
    cn = synth.face_normals(xyz, tri).astype(np.float32)
    cn[::7, 2] = np.nan
    ctx2 = icpx.Context(0)
    ctx2.set_target(xyz, tri, cell_normals=cn)
    ctx2.set_source(src)
    Td, it, mean, st = ctx2.run(p)
    assert st == icpx.RUN_DIVERGED and it == 0 and np.array_equal(Td, np.eye(4))
    ctx2.set_allreduce(lambda sums: None)   # the loop on the host
    Th, ith, meanh, sth = ctx2.run(p)
    assert sth == icpx.RUN_DIVERGED and ith == 0 and np.array_equal(Th, np.eye(4))
    ctx2.close()
    ctx.close()


@pytest.mark.gpu
def test_strays_and_queries_the_pooled_lanes_give_up_on(icpx, synth, oracle):
    """Points anywhere in the target's box among a cloud on its surface, no threshold: a lane
    of the pooled kernel that walks alone takes a trip to memory per box, thousands for a
    stray in the hole of the torus, so it gives the query up after a budget of steps and the
    kernel behind serves it one wavefront to itself.  Rows with a budget of 32 steps (most
    queries given up on), the default and none are the same, cold, hinted and with
    certificates, and the oracle's on a sample that holds every stray."""
    xyz, tri = synth.bumpy_torus(200, 100)
    pts, f, fn = synth.sample_surface(xyz, tri, 150_000, seed=71)
    lo, hi = xyz.min(0).astype(np.float64), xyz.max(0).astype(np.float64)
    n = 600
    stray = np.stack([synth.u01(73, k, n) for k in range(3)], -1) * (hi - lo) + lo
    src = synth.apply(synth.rigid(ry=1.5, t=(0.01, 0.0, -0.01)), pts)
    where = np.arange(n) * (len(src) // n)
    src[where] = stray.astype(np.float32)
    mesh = oracle.Mesh(xyz, tri)
    pick = np.union1d(where, np.arange(0, len(src), 37))
    rows = {}
    # (... and the wavefront walk with the strays set aside before their batches' walks)
    for budget, knobs in ((32, dict(local=0, persistent=1)), (32, {}), (512, {}), (512, dict(defer=1, persistent=3)),
                          (0, dict(local=0, persistent=1))):
        ctx = icpx.Context(0)
        ctx.set_step_budget(budget)
        for k, v in knobs.items():
            getattr(ctx, "set_" + k)(v)
        ctx.set_target(xyz, tri)
        ctx.set_source(src)
        T = np.eye(4)
        out = []
        for i, m in enumerate((0, 0, 1e-5, 2e-3, 1e-5, 1e-6)):
            T = synth.rigid(t=(m, -m, 0.5 * m)) @ T
            got = ctx.correspond(T, step=1, max_dist=1e9)
            out.append({k: got[k].copy() for k in ("cell", "dist", "closest")})
            if budget == 32 and knobs and i == 0:
                gave = ctx.debug_counters()["pool"]   # (the pass's third counter: queries given up on)
                assert gave > len(src) // 4, gave
            if "defer" in knobs and i == 1:
                aside = ctx.debug_counters()["pool"]
                assert n // 2 < aside <= n, aside   # (the strays, and nobody else)
        rows[(budget, tuple(sorted(knobs)))] = out
        ctx.close()
    base = rows[(0, ("local", "persistent"))]
    for key, out in rows.items():
        for i, (a, b) in enumerate(zip(base, out)):
            for k in a:
                assert np.array_equal(a[k], b[k]), (key, i, k)
    T = np.eye(4)
    for i, m in enumerate((0, 0, 1e-5)):
        T = synth.rigid(t=(m, -m, 0.5 * m)) @ T
        if i != 1:
            ref = mesh.correspond(src[pick], T=T, step=1, max_dist=1e9, grid=True)
            for k in ("cell", "dist", "closest"):
                assert np.array_equal(base[i][k][pick], ref[k]), (i, k)


def _point_triangle_dist(q, P):
    """Distances from points q (n,3) to triangles P (m,3,3), all pairs, float64 (Ericson)."""
    a, b, c = P[None, :, 0], P[None, :, 1], P[None, :, 2]
    p = q[:, None, :]
    ab, ac, ap = b - a, c - a, p - a
    d1, d2 = (ab * ap).sum(-1), (ac * ap).sum(-1)
    bp = p - b
    d3, d4 = (ab * bp).sum(-1), (ac * bp).sum(-1)
    cp = p - c
    d5, d6 = (ab * cp).sum(-1), (ac * cp).sum(-1)
    va, vb, vc = d3 * d6 - d5 * d4, d5 * d2 - d1 * d6, d1 * d4 - d3 * d2
    with np.errstate(divide="ignore", invalid="ignore"):
        den = va + vb + vc
        v, w = vb / den, vc / den
        pt = a + ab * v[..., None] + ac * w[..., None]
        t_ab = (d1 / (d1 - d3))[..., None]
        t_ac = (d2 / (d2 - d6))[..., None]
        t_bc = ((d4 - d3) / ((d4 - d3) + (d5 - d6)))[..., None]
    pt = np.where(((va <= 0) & (d4 - d3 >= 0) & (d5 - d6 >= 0))[..., None], b + (c - b) * t_bc, pt)
    pt = np.where(((vb <= 0) & (d2 >= 0) & (d6 <= 0))[..., None], a + ac * t_ac, pt)
    pt = np.where(((d6 >= 0) & (d5 <= d6))[..., None], c + 0 * p, pt)
    pt = np.where(((vc <= 0) & (d1 >= 0) & (d3 <= 0))[..., None], a + ab * t_ab, pt)
    pt = np.where(((d3 >= 0) & (d4 <= d3))[..., None], b + 0 * p, pt)
    pt = np.where(((d1 <= 0) & (d2 <= 0))[..., None], a + 0 * p, pt)
    return np.sqrt(((p - pt) ** 2).sum(-1))


def _closest_on_triangles(q, P):
    """Closest points and distances from points q (n,3) to their nearest of triangles P
    (m,3,3), all pairs in float64 by Ericson's region walk (Real-Time Collision Detection,
    5.1.5): no arithmetic in common with the restated vtkTriangle::EvaluatePosition."""
    a, b, c = P[None, :, 0], P[None, :, 1], P[None, :, 2]
    p = q[:, None, :]
    ab, ac, ap = b - a, c - a, p - a
    d1, d2 = (ab * ap).sum(-1), (ac * ap).sum(-1)
    bp = p - b
    d3, d4 = (ab * bp).sum(-1), (ac * bp).sum(-1)
    cp = p - c
    d5, d6 = (ab * cp).sum(-1), (ac * cp).sum(-1)
    va, vb, vc = d3 * d6 - d5 * d4, d5 * d2 - d1 * d6, d1 * d4 - d3 * d2
    with np.errstate(divide="ignore", invalid="ignore"):
        den = va + vb + vc
        v, w = vb / den, vc / den
        pt = a + ab * v[..., None] + ac * w[..., None]
        t_ab = (d1 / (d1 - d3))[..., None]
        t_ac = (d2 / (d2 - d6))[..., None]
        t_bc = ((d4 - d3) / ((d4 - d3) + (d5 - d6)))[..., None]
    pt = np.where(((va <= 0) & (d4 - d3 >= 0) & (d5 - d6 >= 0))[..., None], b + (c - b) * t_bc, pt)
    pt = np.where(((vb <= 0) & (d2 >= 0) & (d6 <= 0))[..., None], a + ac * t_ac, pt)
    pt = np.where(((d6 >= 0) & (d5 <= d6))[..., None], c + 0 * p, pt)
    pt = np.where(((vc <= 0) & (d1 >= 0) & (d3 <= 0))[..., None], a + ab * t_ab, pt)
    pt = np.where(((d3 >= 0) & (d4 <= d3))[..., None], b + 0 * p, pt)
    pt = np.where(((d1 <= 0) & (d2 <= 0))[..., None], a + 0 * p, pt)
    dist = np.sqrt(((p - pt) ** 2).sum(-1))
    k = np.argmin(dist, axis=1)
    i = np.arange(len(q))
    return pt[i, k], dist[i, k]


def test_emitted_distances_against_an_independent_routine(icpx, synth):
    """What the library emits (double-precision rows: distance and closest point) against a
    textbook closest-point-on-triangle routine over all triangles: the closest point on a
    triangle is unique, so this pins everything but the choice among tied cells and the last
    bits.  The Bunny stand-in with queries on, near, far from and exactly on vertices of it; the
    partial-overlap pair with outliers; a soup of slivers and overlapping triangles."""
    c1 = synth.config_c1()
    q1 = queries_around(synth, c1["target_xyz"], c1["target_tri"], 3000, seed=19)
    c5 = synth.config_c5(nu=96, nv=48, npts=6000)
    n = 3000
    ctr = (np.stack([synth.u01(4, 10 + k, n) for k in range(3)], -1) - 0.5) * 2.0
    size = 0.002 + 0.2 * synth.u01(4, 20, n) ** 4
    e1 = (np.stack([synth.u01(4, 30 + k, n) for k in range(3)], -1) - 0.5) * size[:, None]
    e2 = (np.stack([synth.u01(4, 40 + k, n) for k in range(3)], -1) - 0.5) * size[:, None]
    sl = synth.u01(4, 50, n) < 0.1
    e2[sl] = e1[sl] * 0.5 + e2[sl] * 1e-4
    soup = np.ascontiguousarray(np.stack([ctr, ctr + e1, ctr + e2], 1).reshape(-1, 3), np.float32)
    soup_tri = np.arange(len(soup), dtype=np.int32).reshape(-1, 3)
    qs = (ctr[:2000] + 0.05 * (np.stack([synth.u01(4, 60 + k, 2000) for k in range(3)], -1) - 0.5)
          ).astype(np.float32)
    T = synth.rigid(rx=2.0, ry=-1.0, t=(0.01, 0.0, -0.02))
    for xyz, tri, q in ((c1["target_xyz"], c1["target_tri"], q1),
                        (c5["target_xyz"], c5["target_tri"], c5["source_xyz"][::3]),
                        (soup, soup_tri, qs)):
        ctx = icpx.Context(0)
        ctx.set_target(xyz, tri)
        ctx.set_source(q)
        rows = ctx.correspond_f64(T, step=1, max_dist=1e30)
        assert len(rows["cell"]) == len(q)
        x = q.astype(np.float64) @ T[:3, :3].T + T[:3, 3]
        assert np.abs(rows["origin"] - x).max() <= 1e-15 * max(1.0, float(np.abs(x).max()))
        P = xyz[tri].astype(np.float64)
        extent = float(np.abs(P).max())
        # (degenerate triangles are skipped by the locator; none of these meshes has one)
        pt, dist = _closest_on_triangles(rows["origin"], P)
        assert np.abs(rows["dist"] - dist).max() <= 1e-12 * extent
        assert np.abs(rows["closest"] - pt).max() <= 1e-9 * extent
        # and the float rows are the roundings of the same numbers
        rf = ctx.correspond(T, step=1, max_dist=1e30)
        assert np.array_equal(rf["dist"], rows["dist"])
        assert np.array_equal(rf["closest"], rows["closest"].astype(np.float32))


@pytest.mark.parametrize("case", hand_cases(), ids=lambda c: c[0])
def test_hand_derived_answers_device(icpx, case):
    """Answers worked out by hand (an edge between faces at a dihedral that is not a right
    angle, a vertex whose incident angles differ, a query in every outside region of
    EvaluatePosition) from the device, float and double rows."""
    name, xyz, tri, q, want = case
    ctx = icpx.Context(0)
    ctx.set_target(xyz, tri)
    ctx.set_source(q)
    for width in (0, 1, 64):
        ctx.set_group_width(width)
        r64 = ctx.correspond_f64(np.eye(4), step=1, max_dist=10.0)
        assert np.allclose(r64["closest"], want["closest"], atol=1e-15), name
        assert np.abs(r64["dist"] - want["dist"]).max() <= 1e-15, name
        assert r64["cell"].tolist() == list(want["cell"]), name
        assert np.allclose(r64["normal"], want["normal"], atol=1e-15), name
        r32 = ctx.correspond(np.eye(4), step=1, max_dist=10.0)
        assert np.array_equal(r32["closest"], np.asarray(want["closest"], np.float32))
        assert np.abs(r32["normal"] - np.asarray(want["normal"], np.float32)).max() <= 1e-7
        assert r32["cell"].tolist() == list(want["cell"])


@pytest.mark.parametrize("local", [1, 3])
def test_certificate_is_below_the_true_gap(icpx, synth, local):
    """The slack a query is given never exceeds what it certifies: the distance of the
    second nearest triangle less that of the nearest (all pairs, brute force in numpy).
    local = 3: as for a cloud of millions (certificates checked by their own kernel, and a
    winner with a rival next to nothing behind kept with it as a set of two)."""
    xyz, tri = synth.bumpy_torus(48, 24)
    pts, f, fn = synth.sample_surface(xyz, tri, 40_000, seed=33)
    off = np.stack([synth.gaussian(8, 90 + 2 * k, len(pts)) for k in range(3)], -1)
    scale = 10.0 ** (-8 + 6 * synth.u01(8, 97, len(pts)))
    src = (pts.astype(np.float64) + off * scale[:, None]).astype(np.float32)
    ctx = icpx.Context(0)
    ctx.set_group_width(1)
    ctx.set_local(local)
    ctx.set_target(xyz, tri)
    ctx.set_source(src)
    T = synth.rigid(rz=0.01, t=(1e-4, 0, 0))
    for _ in range(3):
        ctx.accumulate(1, T, step=1, max_dist=10.0)  # cold, tree, then the shortcut: certificates
    T2 = synth.rigid(rx=2e-6, t=(0, 1e-7, 0)) @ T
    ctx.accumulate(1, T2, step=1, max_dist=10.0)     # a small move: slacks shrink by twice it
    cert = ctx.debug_certificates(len(src))
    order = ctx.debug_query_order(len(src))
    assert (cert > 0).mean() > 0.9
    P = xyz[tri].astype(np.float64)
    x = src.astype(np.float64) @ T2[:3, :3].T + T2[:3, 3]  # as the kernels transform: in fp64
    worst = 0.0
    isset = (cert > 0) & ((cert.view(np.uint32) & 1) == 1)  # stands for a set of contenders
    sets = ctx.debug_contender_sets(len(src))
    cell = ctx.debug_cell_ids(len(tri))
    lead = ctx.debug_leads(len(src))
    led = (cert > 0) & ((cert.view(np.uint32) & 3) == 3)    # ... whose winner's lead is on record
    nsets = nled = 0
    # a sample of the queries, in chunks, and every query that stands on a set
    chunks = [np.arange(k0 * 10, k0 * 10 + 500) for k0 in range(0, 4000, 500)]
    chunks.append(np.nonzero(isset)[0][:400])
    for pos in chunks:
        dall = _point_triangle_dist(x[order[pos]], P)
        d = np.sort(dall, axis=1)
        gap = d[:, 1] - d[:, 0]
        for i in np.nonzero(isset[pos])[0]:
            # a set certificate: the gap to the nearest triangle outside the set
            members = sets[pos[i]]
            members = members[:np.argmax(members < 0)] if (members < 0).any() else members
            assert 2 <= len(members) <= 8
            out = np.ones(len(P), bool)
            out[cell[members]] = False
            gap[i] = dall[i][out].min() - d[i, 0]
            nsets += 1
            if led[pos[i]]:
                # the lead on record: no more than the next member is behind the nearest one
                dm = np.sort(dall[i][cell[members]])
                assert 0.0 < lead[pos[i]] <= dm[1] - dm[0] + 1e-15, "a lead exceeds the true one"
                nled += 1
        assert np.all(cert[pos] <= gap + 1e-12), "a certificate exceeds the true gap"
        worst = max(worst, float((cert[pos] / np.maximum(gap, 1e-300)).max()))
    assert worst <= 1.0 + 1e-9
    assert isset.sum() > 0 and nsets > 0  # (points sampled on the surface: some lie on edges)
    assert nled > 0 or local == 1           # (pairs with a clear winner: kept where sets of two are)


def test_certificates_of_points_far_above_small_triangles(icpx, synth, oracle):
    """A noisy cloud over a graded mesh: a point dozens of triangle widths above the small
    triangles has dozens of them within a hair of the nearest.  Such a point is certified by
    its eight nearest triangles and the exact distance of the ninth; every certificate,
    these included, is held against the brute-force gap to the nearest triangle outside the
    set, rows are the oracle's before and after a small move, and the settled passes leave
    next to nothing to search."""
    xyz, tri = synth.bumpy_torus(240, 120, grade=0.985)
    # (as many points over a small triangle as over a large one: an area-uniform sample of this
    # size has next to none over the small ones)
    f = np.minimum((synth.u01(41, 1, 30_000) * len(tri)).astype(np.int64), len(tri) - 1)
    w = np.sort(np.stack([synth.u01(41, 2, len(f)), synth.u01(41, 3, len(f))], -1), axis=1)
    bary = np.stack([w[:, 0], w[:, 1] - w[:, 0], 1.0 - w[:, 1]], -1)
    pts = (xyz[tri[f]].astype(np.float64) * bary[:, :, None]).sum(1)
    noise = np.stack([synth.gaussian(9, 90 + 2 * k, len(pts)) for k in range(3)], -1) * 0.03
    src = (pts.astype(np.float64) + noise).astype(np.float32)
    mesh = oracle.Mesh(xyz, tri)
    ctx = icpx.Context(0)
    ctx.set_group_width(1)
    ctx.set_certify_min(1000)   # served as a cloud of millions: certificates by their own kernel
    ctx.set_target(xyz, tri)
    ctx.set_source(src)
    T = synth.rigid(rz=0.01, t=(1e-4, 0, 0))
    for _ in range(5):
        ctx.accumulate(1, T, step=1, max_dist=10.0)
    T2 = synth.rigid(rx=2e-8, t=(0, 1e-9, 0)) @ T
    rows = ctx.correspond(T2, step=1, max_dist=10.0)
    check_rows(rows, mesh.correspond(src, T=T2, step=1, max_dist=10.0, grid=True))
    left = ctx.debug_counters()["fallback"]
    cert = ctx.debug_certificates(len(src))
    order = ctx.debug_query_order(len(src))
    sets = ctx.debug_contender_sets(len(src))
    cell = ctx.debug_cell_ids(len(tri))
    isset = (cert > 0) & ((cert.view(np.uint32) & 1) == 1)
    size = np.where((sets < 0).any(1), np.argmax(sets < 0, 1), 8)
    full = np.nonzero(isset & (size == 8))[0]
    assert len(full) >= 5, "no point stands on a set of eight"
    assert (cert > 0).mean() > 0.98 and left < len(src) // 200, (float((cert > 0).mean()), left)
    P = xyz[tri].astype(np.float64)
    x = src.astype(np.float64) @ T2[:3, :3].T + T2[:3, 3]
    pick = np.concatenate([full[:300], np.nonzero(isset)[0][:200], np.arange(0, len(src), 97)])
    for k0 in range(0, len(pick), 250):
        pos = pick[k0:k0 + 250]
        dall = _point_triangle_dist(x[order[pos]], P)
        d = np.sort(dall, axis=1)
        gap = d[:, 1] - d[:, 0]
        for i in np.nonzero(isset[pos])[0]:
            members = sets[pos[i]]
            members = members[:np.argmax(members < 0)] if (members < 0).any() else members
            out = np.ones(len(P), bool)
            out[cell[members]] = False
            gap[i] = dall[i][out].min() - d[i, 0]
            # (the nearest triangle is a member: the winner is found among them)
            assert dall[i][cell[members]].min() == d[i, 0]
        c = cert[pos]
        assert np.all(c[c > 0] <= gap[c > 0] + 1e-12), "a certificate exceeds the true gap"
    ctx.close()


def test_tree_and_far_certificates_are_sound(icpx, synth, oracle):
    """Certificates made by the tree kernels, for a cloud that is mostly off the surface and
    partly beyond the distance threshold: a positive one never exceeds the brute-force gap
    between the nearest and the second nearest triangle, a negative one never exceeds the
    distance by which the nearest triangle is beyond the threshold; and the rows of every pass
    equal the oracle's and those of the library with certificates off."""
    xyz, tri = synth.bumpy_torus(48, 24)
    pts, f, fn = synth.sample_surface(xyz, tri, 40_000, seed=41)
    off = np.stack([synth.gaussian(9, 90 + 2 * k, len(pts)) for k in range(3)], -1)
    scale = 10.0 ** (-5 + 4.5 * synth.u01(9, 97, len(pts)))        # 1e-5 .. 0.3 off the surface
    src = (pts.astype(np.float64) + off * scale[:, None]).astype(np.float32)
    cap = 0.05
    mesh = oracle.Mesh(xyz, tri)
    ctx, plain, listed = icpx.Context(0), icpx.Context(0), icpx.Context(0)
    for c_, local in ((ctx, 1), (plain, 0), (listed, 3)):
        c_.set_group_width(1)
        c_.set_local(local)
        c_.set_target(xyz, tri)
        c_.set_source(src)
    T = synth.rigid(rz=0.01, t=(1e-4, 0, 0))
    for k, m in enumerate((0, 1e-3, 1e-5, 1e-6, 1e-7, 1e-6, 1e-7, 1e-8)):
        T = synth.rigid(rx=3 * m, t=(0, m, -m)) @ T
        rows = ctx.correspond(T, step=1, max_dist=cap)
        check_rows(rows, plain.correspond(T, step=1, max_dist=cap))
        check_rows(rows, listed.correspond(T, step=1, max_dist=cap))
        if k in (0, 3, 7):
            check_rows(rows, mesh.correspond(src, T=T, step=1, max_dist=cap, grid=True))
        ctx.accumulate(1, T, step=1, max_dist=cap)
        plain.accumulate(1, T, step=1, max_dist=cap)
        listed.accumulate(1, T, step=1, max_dist=cap)
    P = xyz[tri].astype(np.float64)
    x = src.astype(np.float64) @ T[:3, :3].T + T[:3, 3]
    for c_ in (ctx, listed):  # (the listed path's leftovers may have gone to another tree kernel)
        cert = c_.debug_certificates(len(src))
        order = c_.debug_query_order(len(src))
        assert (cert > 0).mean() > 0.3 and (cert < 0).mean() > 0.1
        for k0 in range(0, 40_000, 8000):
            pos = np.arange(k0, k0 + 400)
            d = np.sort(_point_triangle_dist(x[order[pos]], P), axis=1)
            near, far = cert[pos] > 0, cert[pos] < 0
            near &= (cert[pos].view(np.uint32) & 1) == 0  # (set certificates: tested elsewhere)
            assert np.all(cert[pos][near] <= (d[:, 1] - d[:, 0])[near] + 1e-12)
            assert np.all(-cert[pos][far] <= d[:, 0][far] - cap + 1e-12)


@pytest.mark.parametrize("metric", [0, 1, 2])
def test_finalize_builds_agree(icpx, synth, small, metric):
    """The two builds of the fused finalize (sums in registers; sums as the Gram matrix of a
    16-vector per row on the matrix core, v_mfma_f64_16x16x4_f64) on the same rows: the same
    count, sums equal to rounding, and both equal to the numpy restatement from the oracle's
    rows."""
    c, mesh = small
    sn = synth.vertex_normals(c["source_xyz"], c["target_tri"])
    T = synth.rigid(ry=2.0, t=(0.0, 0.01, 0.0))
    ref = mesh.correspond(c["source_xyz"], T=T, step=1, max_dist=0.2,
                          src_normals=sn if metric == 2 else None, grid=False)
    got = {}
    for build in (1, 2):
        ctx = icpx.Context(0)
        ctx.set_finalize(build)
        ctx.set_target(c["target_xyz"], c["target_tri"])
        ctx.set_source(c["source_xyz"], sn)
        got[build] = ctx.accumulate(metric, T, step=1, max_dist=0.2)
        again = ctx.accumulate(metric, T, step=1, max_dist=0.2)
        assert np.array_equal(got[build], again), "sums are not reproducible run to run"
        want = sums_from_rows(metric, ref["origin"], ref["closest"], ref["normal"],
                              ref["origin_normal"], ctx.origin)
        ctx.close()
    scale = np.abs(want).max()
    assert got[1][0] == got[2][0] == want[0] == len(ref["cell"]) > 0
    assert np.abs(got[1] - got[2]).max() <= 1e-12 * scale
    assert np.abs(got[1] - want).max() <= 1e-11 * scale
    assert np.abs(got[2] - want).max() <= 1e-11 * scale


def test_point_normals_on_device(icpx, synth, oracle):
    """icpx_point_normals against the numpy restatement of the filter's vtkPolyDataNormals
    (normalised sum of incident unit face normals): a closed surface, plus a point no triangle
    uses (zero normal), a degenerate triangle (adds nothing) and two faces that cancel."""
    xyz, tri = synth.bumpy_torus(120, 60)
    extra = np.array([[5, 5, 5], [6, 5, 5], [6, 6, 5], [7, 7, 7]], np.float32)
    n0 = len(xyz)
    xyz2 = np.concatenate([xyz, extra])
    tri2 = np.concatenate([tri, np.array([[n0, n0 + 1, n0 + 2], [n0, n0 + 2, n0 + 1],  # cancel
                                          [n0 + 1, n0 + 1, n0 + 2]], np.int32)])     # degenerate
    ctx = icpx.Context(0)
    got = ctx.point_normals(xyz2, tri2)
    want = oracle.point_normals(xyz2, tri2)
    assert got.shape == want.shape and got.dtype == np.float32
    assert np.abs(got - want).max() <= 1e-6   # (sums in a different order: a unit in the last place)
    assert np.all(got[n0:] == 0)
    ln = np.linalg.norm(got[:n0].astype(np.float64), axis=1)
    assert np.abs(ln - 1).max() < 1e-6
    assert np.array_equal(got, ctx.point_normals(xyz2, tri2))  # repeats bit for bit
    with pytest.raises(icpx.IcpxError):
        ctx.point_normals(xyz2, np.array([[0, 1, len(xyz2)]], np.int32))


def test_point_normals_with_inconsistent_winding(icpx, synth, oracle):
    """ConsistencyOn of the filter's vtkPolyDataNormals (ICP/vtkPolyDataCorrespondenceFilter.cxx:358):
    a closed surface with a third of its triangles wound the wrong way, plus a separate
    patch, against the restated walk of vtkPolyDataNormals::TraverseAndOrder; every piece
    ends up wound like its lowest cell, so flipping cell 0 as well turns all normals round."""
    xyz, tri = synth.bumpy_torus(120, 60)
    n0 = len(xyz)
    patch_xyz = np.array([[5, 5, 5], [6, 5, 5], [6, 6, 5], [5, 6, 5], [5.5, 5.5, 6]], np.float32)
    patch_tri = np.array([[0, 1, 4], [2, 1, 4], [2, 3, 4], [0, 3, 4]], np.int32) + n0  # mixed winding
    xyz2 = np.concatenate([xyz, patch_xyz])
    clean = np.concatenate([tri, patch_tri])
    flip = synth.u01(3, 5, len(clean)) < 0.33
    flip[0] = False
    flip[len(tri):] = False
    bad = clean.copy()
    bad[flip] = bad[flip][:, [1, 0, 2]]                       # swap two points: reversed winding
    ctx = icpx.Context(0)
    got = ctx.point_normals(xyz2, bad)
    assert ctx.debug_stats()["winding_flips"] == int(flip.sum()) + 2   # and two of the patch's four
    want = oracle.point_normals(xyz2, bad)
    assert np.abs(got - want).max() <= 1e-6
    good = ctx.point_normals(xyz2[:n0], tri)                   # the consistent surface by itself
    assert ctx.debug_stats()["winding_flips"] == 0
    assert np.abs(got[:n0] - good).max() <= 1e-6
    outward = synth.vertex_normals(xyz, tri)
    assert np.abs(good - outward).max() <= 1e-6
    bad[0] = bad[0][[1, 0, 2]]                                 # now the lowest cell is the odd one
    inv = ctx.point_normals(xyz2, bad)
    assert np.abs(inv[:n0] + good).max() <= 1e-6 and np.abs(inv[n0:] - got[n0:]).max() <= 1e-6
    assert np.abs(inv - oracle.point_normals(xyz2, bad)).max() <= 1e-6
    # without the consistency pass the flipped faces would cancel their neighbours
    assert np.abs(oracle.point_normals(xyz2, bad, consistency=False) - inv).max() > 0.5


def test_point_normals_on_a_long_strip_and_a_moebius_band(icpx, synth):
    """The consistency pass on a piece that is 200 000 cells long and one cell wide (labels
    must not need a sweep per cell to cross it), and on a band with a half twist, which
    cannot be wound consistently: that is reported."""
    n = 100_000
    i = np.arange(n + 1, dtype=np.float32)
    xyz = np.zeros((2 * (n + 1), 3), np.float32)
    xyz[0::2, 0], xyz[1::2, 0], xyz[1::2, 1] = i, i, 1.0
    k = np.arange(n, dtype=np.int32)
    tri = np.empty((2 * n, 3), np.int32)
    tri[0::2] = np.stack([2 * k, 2 * k + 2, 2 * k + 1], -1)
    tri[1::2] = np.stack([2 * k + 1, 2 * k + 2, 2 * k + 3], -1)
    flip = synth.u01(7, 3, len(tri)) < 0.4
    flip[0] = False
    bad = tri.copy()
    bad[flip] = bad[flip][:, [1, 0, 2]]
    ctx = icpx.Context(0)
    got = ctx.point_normals(xyz, bad)
    st = ctx.debug_stats()
    assert st["winding_flips"] == int(flip.sum()) and not st["unorientable"]
    assert np.array_equal(got, np.tile(np.array([0, 0, 1], np.float32), (len(xyz), 1)))
    # a band with a half twist: 64 quads, the last joined to the first upside down
    m = 64
    t = 2 * np.pi * np.arange(m) / m
    mid = np.stack([np.cos(t), np.sin(t), np.zeros(m)], -1)
    off = 0.3 * np.stack([np.cos(t / 2) * np.cos(t), np.cos(t / 2) * np.sin(t), np.sin(t / 2)], -1)
    band = np.empty((2 * m, 3), np.float32)
    band[0::2], band[1::2] = mid - off, mid + off
    q = np.arange(m, dtype=np.int32)
    a, b = 2 * q, 2 * q + 1
    c, d = (2 * (q + 1)) % (2 * m), (2 * (q + 1) + 1) % (2 * m)
    c[-1], d[-1] = 1, 0                                    # the twist: rows swapped at the seam
    btri = np.concatenate([np.stack([a, c, b], -1), np.stack([b, c, d], -1)]).astype(np.int32)
    ctx.point_normals(band, btri)
    assert ctx.debug_stats()["unorientable"]


def test_registration_demo_from_a_file(icpx, synth, tmp_path, capsys):
    """The headless registration demo end to end: a PLY file written here, read by the reader
    factory, perturbed as the reference's demo does and registered back with every metric."""
    import importlib.util
    xyz, tri = synth.bumpy_torus(64, 32)
    path = tmp_path / "surface.ply"
    with open(path, "wb") as fh:
        fh.write(("ply\nformat binary_little_endian 1.0\nelement vertex %d\nproperty float x\n"
                  "property float y\nproperty float z\nelement face %d\n"
                  "property list uchar int vertex_indices\nend_header\n" % (len(xyz), len(tri))).encode())
        fh.write(xyz.astype("<f4").tobytes())
        rec = np.zeros(len(tri), np.dtype([("n", "u1"), ("v", "<i4", 3)]))
        rec["n"], rec["v"] = 3, tri
        fh.write(rec.tobytes())
    spec = importlib.util.spec_from_file_location(
        "registration_demo", os.path.join(os.path.dirname(__file__), "..", "tools", "registration_demo.py"))
    demo = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(demo)
    for metric, tol in (("point", 5e-3), ("plane", 1e-4), ("symmetric", 1e-4)):
        argv = sys.argv
        sys.argv = ["registration_demo.py", str(path), "--metric", metric, "--landmarks", "1000000",
                    "--iterations", "60", "--max-mean-distance", "1e-7"]
        try:
            assert demo.main() == 0
        finally:
            sys.argv = argv
        out = capsys.readouterr().out
        assert "Number of iterations:" in out and "Mean distance:" in out
        worst = float(out.strip().splitlines()[-1].split(":")[1])
        assert worst < tol, (metric, out)


def _sharded_rank(rank, world, port, metric, out, landmarks=None):
    """One rank of test_sharded_device_loop_two_ranks (spawned)."""
    import torch.distributed as dist
    from iterativeclosestpoint_amd import icpx as ix, synth as sy
    from iterativeclosestpoint_amd.distributed import ShardedICP
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    xyz, tri = sy.bumpy_torus(200, 100)
    pts, f, fn = sy.sample_surface(xyz, tri, 50_001, seed=9)      # an odd count: uneven shards
    M = sy.rigid(ry=3.0, t=(0.01, 0.0, 0.02))
    src = sy.apply(M, pts)
    sn = (fn.astype(np.float64) @ M[:3, :3].T).astype(np.float32)
    sh = ShardedICP(device=0, rank=rank, world=world)
    sh.set_target(xyz, tri)
    sh.set_source(src, sn)
    p = ix.default_params()
    p.metric, p.max_iterations, p.max_landmarks, p.max_distance = metric, 12, landmarks or len(src), 10.0
    p.check_mean_distance, p.max_mean_distance = 1, 1e-7
    T, it, mean, st = sh.run_device(p)
    out.put((rank, T, it, mean, st))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("metric", (1, 2))
def test_sharded_device_loop_two_ranks(icpx, synth, metric):
    """The multi-GPU driver's loop with two ranks (two processes sharing this GPU, gloo
    carrying the device tensors): each searches its shard, the 40 sums are all-reduced on the
    stream, both solve on the device; same transform and stop as one rank with everything."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_rank, args=(r, 2, port, metric, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    xyz, tri = synth.bumpy_torus(200, 100)
    pts, f, fn = synth.sample_surface(xyz, tri, 50_001, seed=9)
    M = synth.rigid(ry=3.0, t=(0.01, 0.0, 0.02))
    src = synth.apply(M, pts)
    sn = (fn.astype(np.float64) @ M[:3, :3].T).astype(np.float32)
    one = icpx.Context(0)
    one.set_target(xyz, tri)
    one.set_source(src, sn)
    p = icpx.default_params()
    p.metric, p.max_iterations, p.max_landmarks, p.max_distance = metric, 12, len(src), 10.0
    p.check_mean_distance, p.max_mean_distance = 1, 1e-7
    T, it, mean, st = one.run(p)
    assert np.array_equal(res[0][1], res[1][1]) and res[0][2:] == res[1][2:]
    assert (res[0][2], res[0][4]) == (it, st)
    assert rel_err(res[0][1], T) <= 1e-11 and abs(res[0][3] - mean) <= 1e-12
    assert np.abs(T @ M - np.eye(4)).max() < 1e-5


def test_cxx_binding_matches_ctypes(icpx, synth, small, tmp_path):
    """The compiled C++ consumer (tests/consumer/vtkicp_shim.cpp: the call sequence documented
    for vtkICP::InternalUpdate, plain arrays in place of VTK objects) against the ctypes path:
    same library, same inputs, so the printed 4x4 must round-trip to the identical doubles."""
    import subprocess
    c, mesh = small
    cdir = os.path.join(os.path.dirname(__file__), "consumer")
    shim = os.path.join(cdir, "_build", "vtkicp_shim")
    if not os.path.exists(shim):
        subprocess.check_call(["make", "-s", "-C", cdir])
    sn = synth.vertex_normals(c["source_xyz"], c["target_tri"])
    c["target_xyz"].tofile(tmp_path / "target_xyz.f32")
    c["target_tri"].tofile(tmp_path / "target_tri.i32")
    c["source_xyz"].tofile(tmp_path / "source_xyz.f32")
    sn.tofile(tmp_path / "source_normals.f32")
    for metric, landmarks, check in ((0, 100, 0), (1, 1000, 1), (2, 500, 1)):
        run = subprocess.run([shim, str(tmp_path), str(metric), "40", str(landmarks), "10.0",
                              str(check), "0.001"], capture_output=True, text=True, timeout=120)
        assert run.returncode == 0, run.stderr
        lines = run.stdout.strip().splitlines()
        status, iters = int(lines[0].split()[1]), int(lines[1].split()[1])
        mean = float(lines[2].split()[1])
        M = np.array([[float(v) for v in ln.split()] for ln in lines[3:7]])
        ctx = icpx.Context(0)
        ctx.set_target(c["target_xyz"], c["target_tri"])
        ctx.set_source(c["source_xyz"], sn if metric == 2 else None)
        p = icpx.default_params()
        p.metric, p.max_iterations, p.max_landmarks, p.max_distance = metric, 40, landmarks, 10.0
        p.check_mean_distance, p.max_mean_distance = check, 0.001
        T, it, md, st = ctx.run(p)
        assert (status, iters) == (st, it) and np.array_equal(M, T)
        assert mean == (md if check else 0.0)
        To = mesh.icp(c["source_xyz"], metric=metric, max_iter=40, max_landmarks=landmarks,
                      check_mean=bool(check), max_mean=0.001, src_normals=sn)[0]
        assert rel_err(M, To) <= 1e-6


def test_rotated_source_normals_mode(icpx, synth, small):
    """The opt-in corrected mode (source normals rotated with the running transform; the
    reference leaves them as given, ICP/vtkPolyDataCorrespondenceFilter.cxx:284,299): rows,
    sums and the whole symmetric registration against the oracle with the same switch, and
    the default still equal to the reference's behaviour."""
    c, mesh = small
    sn = synth.vertex_normals(c["source_xyz"], c["target_tri"])
    ctx = icpx.Context(0)
    ctx.set_target(c["target_xyz"], c["target_tri"])
    ctx.set_source(c["source_xyz"], sn)
    T = synth.rigid(rx=20.0, ry=-35.0, rz=10.0, t=(0.02, -0.01, 0.03))
    for md in (0.0, 0.8):
        got = ctx.correspond(T, step=2, max_dist=10.0, min_dot=md, use_source_normals=2)
        ref = mesh.correspond(c["source_xyz"], T=T, step=2, max_dist=10.0, src_normals=sn,
                              min_dot=md, grid=False, rotate_normals=True)
        check_rows(got, ref)
        assert np.array_equal(got["origin_normal"], ref["origin_normal"])
        plain = mesh.correspond(c["source_xyz"], T=T, step=2, max_dist=10.0, src_normals=sn,
                                min_dot=md, grid=False)
        check_rows(ctx.correspond(T, step=2, max_dist=10.0, min_dot=md, use_source_normals=1), plain)
    assert len(ref["cell"]) != len(plain["cell"])  # the switch matters for the dot test
    want = sums_from_rows(2, ref["origin"], ref["closest"], ref["normal"], ref["origin_normal"],
                          ctx.origin)
    got = ctx.accumulate(2, T, step=2, max_dist=10.0, min_dot=0.8, use_source_normals=2)
    assert got[0] == want[0] and np.abs(got - want).max() <= 1e-11 * np.abs(want).max()
    p = icpx.default_params()
    p.metric, p.max_iterations, p.max_landmarks, p.max_distance = 2, 30, 2048, 10.0
    p.rotate_source_normals = 1
    Tr, it, mean, st = ctx.run(p)
    To, ito, mo, sto, n = mesh.icp(c["source_xyz"], metric=2, max_iter=30, max_landmarks=2048,
                                   src_normals=sn, rotate_normals=True)
    assert (it, st) == (ito, sto) and rel_err(Tr, To) <= 1e-6
    p.rotate_source_normals = 0
    T0, it0, _, _ = ctx.run(p)
    assert rel_err(T0, mesh.icp(c["source_xyz"], metric=2, max_iter=30, max_landmarks=2048,
                                src_normals=sn)[0]) <= 1e-6
    # (both converge to the same pose on noise-free data; their first steps differ)
    p.max_iterations = 1
    first0 = ctx.run(p)[0]
    p.rotate_source_normals = 1
    first1 = ctx.run(p)[0]
    assert rel_err(first1, mesh.icp(c["source_xyz"], metric=2, max_iter=1, max_landmarks=2048,
                                    src_normals=sn, rotate_normals=True)[0]) <= 1e-9
    assert np.array_equal(first0, first1)  # the first pass runs at the identity: nothing to rotate
    p.max_iterations = 2
    second1 = ctx.run(p)[0]
    p.rotate_source_normals = 0
    assert rel_err(second1, ctx.run(p)[0]) > 1e-6  # a 15 degree step later the modes differ
    with pytest.raises(icpx.IcpxError):
        ctx.accumulate(2, T, use_source_normals=3)


def test_in_place_edits_reach_the_device(icpx, synth, small):
    """vtkICP re-reads a mesh whose arrays were edited in place once told (Modified())."""
    from iterativeclosestpoint_amd import vtkICP, vtkPolyData
    c, mesh = small
    tgt = vtkPolyData(c["target_xyz"].copy(), c["target_tri"])
    src = vtkPolyData(c["source_xyz"].copy(), c["target_tri"])
    icp = vtkICP()
    icp.SetMaximumNumberOfLandmarks(500)
    icp.SetSource(src)
    icp.SetTarget(tgt)
    M0 = icp.GetMatrix().copy()
    src.GetPoints()[:, 0] += np.float32(0.05)
    src.Modified()
    icp.Modified()
    M1 = icp.GetMatrix().copy()
    shift = np.eye(4)
    shift[0, 3] = 0.05
    assert np.abs(M1 @ shift - M0).max() < 1e-3 and np.abs(M1 - M0).max() > 0.04
    To = mesh.icp(src.GetPoints(), metric=1, max_iter=50, max_landmarks=500)[0]
    assert rel_err(M1, To) <= 1e-6


@pytest.mark.parametrize("landmarks", (1, 7))
def test_sharded_loop_with_an_empty_shard(icpx, synth, landmarks):
    """More ranks than kept points: with MaximumNumberOfLandmarks = 1 one point is kept, rank 0's
    shard is empty and the run stops for want of rows on both ranks, as on one rank; with 7
    (eight kept points, four per rank) the two ranks register like one."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_sharded_rank, args=(r, 2, port, 1, q, landmarks)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in procs], key=lambda r: r[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    xyz, tri = synth.bumpy_torus(200, 100)
    pts, f, fn = synth.sample_surface(xyz, tri, 50_001, seed=9)
    src = synth.apply(synth.rigid(ry=3.0, t=(0.01, 0.0, 0.02)), pts)
    one = icpx.Context(0)
    one.set_target(xyz, tri)
    one.set_source(src)
    p = icpx.default_params()
    p.metric, p.max_iterations, p.max_landmarks, p.max_distance = 1, 12, landmarks, 10.0
    p.check_mean_distance, p.max_mean_distance = 1, 1e-7
    T, it, mean, st = one.run(p)
    assert np.array_equal(res[0][1], res[1][1]) and res[0][2:] == res[1][2:]
    assert (res[0][2], res[0][4]) == (it, st)
    assert rel_err(res[0][1], T) <= 1e-11
    if landmarks == 1:
        assert st == icpx.RUN_TOO_FEW and it == 0 and np.array_equal(T, np.eye(4))
    else:
        assert it >= 1 and st in (icpx.RUN_MAX_ITERATIONS, icpx.RUN_CONVERGED)


def test_line_intersection_matches_bruteforce(icpx, synth, oracle, small):
    """icpx_intersect_lines against the oracle's all-triangles restatement of
    vtkImplicitPolyDataIntersection::IntersectWithLine: hit flags, cells, kinds and every
    double bit for bit (the device runs the same operations in the same order; only the
    vertex normals go through acos and may differ in the last place).  Random chords of a
    closed surface, segments that stop short, rays through vertices and edge midpoints,
    grazing segments; then the unit cube, where edges and corners are hit exactly."""
    c, mesh = small
    xyz, tri = c["target_xyz"], c["target_tri"]
    n = 6000
    lo, hi = xyz.min(0).astype(np.float64), xyz.max(0).astype(np.float64)
    ctr, half = 0.5 * (lo + hi), 0.75 * (hi - lo)
    a = np.stack([synth.u01(5, 10 + k, n) for k in range(3)], -1) * 2 * half + (ctr - half)
    b = np.stack([synth.u01(5, 20 + k, n) for k in range(3)], -1) * 2 * half + (ctr - half)
    b[:500] = a[:500] + 0.02 * (b[:500] - a[:500])          # short ones: mostly no hit
    nrm = synth.vertex_normals(xyz, tri).astype(np.float64)
    v = xyz[:600].astype(np.float64)
    a[1000:1600], b[1000:1600] = v + 0.3 * nrm[:600], v - 0.05 * nrm[:600]   # through vertices
    mid = 0.5 * (xyz[tri[:600, 0]].astype(np.float64) + xyz[tri[:600, 1]])
    fn = synth.face_normals(xyz, tri[:600])
    a[2000:2600], b[2000:2600] = mid + 0.2 * fn, mid - 0.02 * fn             # through edges
    e = xyz[tri[:300, 1]].astype(np.float64) - xyz[tri[:300, 0]]
    cen = xyz[tri[:300]].astype(np.float64).mean(1)
    a[3000:3300], b[3000:3300] = cen - 3 * e + 1e-9 * fn[:300], cen + 3 * e  # grazing a face
    ctx = icpx.Context(0)
    ctx.set_target(xyz, tri)
    got = ctx.intersect_lines(a, b)
    ref = mesh.intersect_lines(a, b)
    assert 0.3 < ref["hit"].mean() < 0.95
    for k in ("hit", "cell", "kind", "ret", "point", "closest"):
        assert np.array_equal(got[k], ref[k]), k
    assert np.abs(got["normal"] - ref["normal"]).max() <= 1e-14
    face = ref["kind"] == 0
    assert np.array_equal(got["normal"][face], ref["normal"][face])
    # every reported hit lies on the segment and on its triangle
    h = ref["hit"] == 1
    t = -got["ret"][h] / np.linalg.norm(b[h] - a[h], axis=1)
    assert np.all((t >= 0) & (t <= 1))
    assert np.abs(a[h] + t[:, None] * (b[h] - a[h]) - got["point"][h]).max() < 1e-12
    # the cube: exact edge and corner hits, a miss, the reference's defaults
    cv = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [0, 0, 1], [1, 0, 1], [1, 1, 1],
                   [0, 1, 1]], np.float32)
    ct = np.array([[0, 2, 1], [0, 3, 2], [4, 5, 6], [4, 6, 7], [0, 1, 5], [0, 5, 4], [2, 3, 7],
                   [2, 7, 6], [1, 2, 6], [1, 6, 5], [0, 4, 7], [0, 7, 3]], np.int32)
    g = np.stack(np.meshgrid(np.linspace(0, 1, 9), np.linspace(0, 1, 9)), -1).reshape(-1, 2)
    p1 = np.concatenate([g, np.full((len(g), 1), 3.0)], 1)
    p2 = np.concatenate([g, np.full((len(g), 1), -3.0)], 1)
    p1 = np.concatenate([p1, [[5, 5, 5]]])
    p2 = np.concatenate([p2, [[6, 6, 6]]])
    cube = icpx.Context(0)
    cube.set_target(cv, ct)
    got = cube.intersect_lines(p1, p2)
    ref = oracle.Mesh(cv, ct).intersect_lines(p1, p2)
    for k in ("hit", "cell", "kind", "ret", "point", "closest"):
        assert np.array_equal(got[k], ref[k]), k
    assert np.abs(got["normal"] - ref["normal"]).max() <= 1e-15
    assert {0, 1, 2} <= set(ref["kind"].tolist())
    assert got["hit"][-1] == 0 and got["ret"][-1] == np.sqrt(3.0)
    assert got["normal"][-1].tolist() == [0, 0, 1] and got["cell"][-1] == -1
    # the class mirror
    from iterativeclosestpoint_amd import vtkImplicitPolyDataIntersection, vtkPolyData
    f = vtkImplicitPolyDataIntersection.New()
    assert f.IntersectWithLine([0.5, 0.5, 3], [0.5, 0.5, -3])[0] == 0   # no input: error, defaults
    f.SetInput(vtkPolyData(cv, ct))
    found, ret, cp, gn, typ = f.IntersectWithLine([0.25, 0.5, 3], [0.25, 0.5, -3])
    assert found == 1 and ret == -2.0 and cp.tolist() == [0.25, 0.5, 1.0] and typ == 1
    assert f.NoValue == np.sqrt(3.0)
    with pytest.raises(icpx.IcpxError):
        icpx.Context(0).intersect_lines(p1, p2)                          # no target set


def test_abi_rejects_bad_input(icpx):
    ctx = icpx.Context(0)
    v = np.zeros((3, 3), np.float32)
    with pytest.raises(icpx.IcpxError):
        ctx.set_target(v, np.array([[0, 1, 3]], np.int32))  # index out of range
    with pytest.raises(icpx.IcpxError):
        ctx.accumulate(1, np.eye(4))  # no target yet
    with pytest.raises(icpx.IcpxError):
        icpx.Context(99)
