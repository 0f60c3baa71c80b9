"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same
seeded inputs.  Bars: cell ids bit-exact (except true equidistant ties, which are
counted and must be zero on these seeds), distances and closest points bit-exact,
normals within 1e-6 (the device's acos may differ from libm's by an ulp before the
float rounding), sums within 1e-11 relative, 4x4 transforms within 1e-6 relative."""
import numpy as np
import pytest

from helpers import rel_err, sums_from_rows

pytestmark = pytest.mark.gpu

WIDTHS = (4, 8, 16, 32, 64)


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


@pytest.mark.parametrize("width", WIDTHS)
def test_closest_point_matches_bruteforce(icpx, synth, small, width):
    c, mesh = small
    q = queries_around(synth, c["target_xyz"], c["target_tri"], 3000, seed=11 + width)
    ctx = icpx.Context(0)
    ctx.set_group_width(width)
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
    for width in (8, 64, 0):
        ctx = icpx.Context(0)
        ctx.set_group_width(width)
        ctx.set_target(xyz, tri)
        ctx.set_source(src)
        check_rows(ctx.correspond(np.eye(4), max_dist=10.0), ref)
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


def test_abi_rejects_bad_input(icpx):
    ctx = icpx.Context(0)
    v = np.zeros((3, 3), np.float32)
    with pytest.raises(icpx.IcpxError):
        ctx.set_target(v, np.array([[0, 1, 3]], np.int32))  # index out of range
    with pytest.raises(icpx.IcpxError):
        ctx.accumulate(1, np.eye(4))  # no target yet
    with pytest.raises(icpx.IcpxError):
        icpx.Context(99)
