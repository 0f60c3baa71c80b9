"""CPU tests of the oracle itself: hand-computable known answers, internal consistency
(bucket grid == brute force), solver known-answer tests, and the committed golden fixtures.
PARITY UNPINNED with respect to the reference (it ships no fixtures for this path)."""
import os

import numpy as np
import pytest

from helpers import rel_err
from iterativeclosestpoint_amd import synth

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_tri_eval_regions(oracle):
    P0, P1, P2 = [0, 0, 0], [1, 0, 0], [0, 1, 0]
    st, cp, d2, w = oracle.tri_eval([0.25, 0.25, 2.0], P0, P1, P2)   # above the interior
    assert st == 1 and np.allclose(cp, [0.25, 0.25, 0]) and d2 == 4.0
    assert np.allclose(w, [0.5, 0.25, 0.25])
    st, cp, d2, w = oracle.tri_eval([0.5, -1.0, 0.0], P0, P1, P2)    # beyond edge P0-P1
    assert st == 0 and np.allclose(cp, [0.5, 0, 0]) and d2 == 1.0
    st, cp, d2, w = oracle.tri_eval([-1.0, -1.0, 0.0], P0, P1, P2)   # beyond corner P0
    assert st == 0 and np.allclose(cp, [0, 0, 0]) and d2 == 2.0
    st, cp, d2, w = oracle.tri_eval([2.0, 2.0, 1.0], P0, P1, P2)     # beyond the hypotenuse
    assert st == 0 and np.allclose(cp, [0.5, 0.5, 0]) and np.isclose(d2, 5.5)
    st, cp, d2, w = oracle.tri_eval([3.0, -1.0, 0.0], P0, P1, P2)    # beyond corner P1
    assert st == 0 and np.allclose(cp, [1, 0, 0]) and d2 == 5.0
    assert oracle.tri_eval([0, 0, 1], [0, 0, 0], [1, 1, 1], [2, 2, 2])[0] == -1  # degenerate


def test_cube_pseudonormals(oracle):
    g = np.load(os.path.join(GOLD, "cube_oracle.npz"))
    m = oracle.Mesh(g["xyz"], g["tri"])
    s3, s2 = 1 / np.sqrt(3), 1 / np.sqrt(2)
    sd, n, cp, cell, kind = m.closest_point_and_normal([2, 2, 2])
    assert kind == 2 and np.allclose(cp, [1, 1, 1]) and np.allclose(n, [s3, s3, s3])
    assert np.isclose(abs(sd), np.sqrt(3))
    sd, n, cp, cell, kind = m.closest_point_and_normal([0.5, 2, 2])
    assert kind == 1 and np.allclose(cp, [0.5, 1, 1]) and np.allclose(n, [0, s2, s2])
    sd, n, cp, cell, kind = m.closest_point_and_normal([0.25, 0.5, 2])
    assert kind == 0 and np.allclose(n, [0, 0, 1]) and cell == 3
    sd, n, cp, cell, kind = m.closest_point_and_normal([0.5, 0.5, 2])  # on the face's diagonal
    assert kind == 1 and np.allclose(n, [0, 0, 1]) and cell == 2      # smallest id of the tie 2/3
    sd_in = m.closest_point_and_normal([0.5, 0.5, 0.75])[0]
    sd_out = m.closest_point_and_normal([0.5, 0.5, 1.25])[0]
    assert sd_in < 0 < sd_out and np.isclose(-sd_in, sd_out)  # inside negative, outside positive


def test_grid_equals_bruteforce(oracle):
    xyz, tri = synth.bumpy_torus(48, 24)
    m = oracle.Mesh(xyz, tri)
    q = (synth.u01(5, 1, 3 * 1500).reshape(-1, 3) - 0.5) * 4.0
    for x in q:
        pa, ca, da = m.find_closest(x, grid=False)
        pb, cb, db = m.find_closest(x, grid=True)
        assert ca == cb and da == db and np.array_equal(pa, pb)


def test_empty_and_degenerate_mesh(oracle):
    v = np.array([[0, 0, 0], [1, 1, 1], [2, 2, 2]], np.float32)
    m = oracle.Mesh(v, np.array([[0, 1, 2]], np.int32))      # one zero-area triangle
    rows = m.correspond(np.array([[0.3, 0.1, 0.0]], np.float32), grid=False)
    assert len(rows["cell"]) == 0                            # the locator skips it
    T, it, md, st, n = m.icp(np.zeros((5, 3), np.float32), metric=1)
    assert st == 2 and it == 0 and np.array_equal(T, np.eye(4))


@pytest.mark.parametrize("metric", (0, 1, 2))
def test_solver_recovers_known_motion(oracle, metric):
    """Noise-free correspondences under a small known rigid motion."""
    xyz, tri = synth.bumpy_torus(40, 20)
    t = xyz.astype(np.float64)
    nrm = synth.vertex_normals(xyz, tri).astype(np.float64)
    M = synth.rigid(rx=0.4, ry=-0.3, rz=0.2, t=(0.003, -0.002, 0.001))
    Minv = np.linalg.inv(M)
    s = t @ Minv[:3, :3].T + Minv[:3, 3]                     # source = M^-1 target
    ns = nrm @ Minv[:3, :3].T
    got, st = oracle.solve(metric, s, t, ns, nrm)
    assert st == 0
    tol = 2e-6 if metric == 0 else 2e-4                      # P2P exact; the others linearise (~theta^2)
    assert np.abs(got - M).max() < tol
    got32, _ = oracle.solve(metric, s, t, ns, nrm, precision=32)
    assert np.abs(got32 - got).max() < 5e-5                  # the float solver's own noise


def test_too_few_rows(oracle):
    s = np.zeros((2, 3), np.float32)
    assert oracle.solve(0, s, s)[1] == 1
    assert oracle.solve(1, np.zeros((5, 3), np.float32), np.zeros((5, 3), np.float32),
                        None, np.zeros((5, 3), np.float32))[1] == 1


def test_stride_rule(oracle):
    """step = N / Max (integer), rows i % step == 0, which can exceed Max
    (vtkPolyDataCorrespondenceFilter.cxx:473-477,112)."""
    c = synth.config_c1()
    m = oracle.Mesh(c["target_xyz"], c["target_tri"])
    n = len(c["source_xyz"])                                 # 2048
    rows = m.correspond(c["source_xyz"], step=n // 100, max_dist=1e9)
    assert len(rows["cell"]) == -(-n // (n // 100)) == 103
    assert np.array_equal(rows["src_index"], np.arange(0, n, n // 100))


def test_golden_c1(oracle):
    g = np.load(os.path.join(GOLD, "c1_oracle.npz"))
    c = synth.config_c1()
    m = oracle.Mesh(c["target_xyz"], c["target_tri"])
    sn = synth.vertex_normals(c["source_xyz"], c["target_tri"])
    rows = m.correspond(c["source_xyz"], T=g["T"], step=int(g["step"]),
                        max_dist=float(g["max_dist"]), src_normals=sn, grid=True)
    for k in ("cell", "dist", "closest", "origin", "src_index", "origin_normal"):
        assert np.array_equal(rows[k], g["rows_" + k]), k
    assert np.abs(rows["normal"] - g["rows_normal"]).max() <= 1e-6
    for metric in (0, 1, 2):
        M, st = oracle.solve(metric, rows["origin"], rows["closest"], rows["origin_normal"],
                             rows["normal"])
        assert rel_err(M, g["solve_%d" % metric]) < 1e-9
        T, it, md, st, n = m.icp(
            c["source_xyz"], metric=metric, max_iter=50, max_landmarks=100,
            max_dist=10.0 if metric == 0 else 0.25 * synth.diag(c["target_xyz"]),
            src_normals=sn if metric == 2 else None)
        assert it == int(g["icp_%d_iters" % metric])
        assert rel_err(T, g["icp_%d" % metric]) < 1e-9
        assert np.abs(T @ synth.demo_perturbation() - np.eye(4)).max() < 1e-3


def test_golden_cube(oracle):
    g = np.load(os.path.join(GOLD, "cube_oracle.npz"))
    rows = oracle.Mesh(g["xyz"], g["tri"]).correspond(g["q"], max_dist=100.0, grid=True)
    for k in ("cell", "dist", "closest", "kind"):
        assert np.array_equal(rows[k], g["rows_" + k]), k
    assert np.allclose(rows["normal"], g["rows_normal"], atol=1e-7)


def test_generators_are_reproducible():
    a = synth.bumpy_torus(16, 8)
    b = synth.bumpy_torus(16, 8)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert a[1].shape == (2 * 16 * 8, 3) and a[1].max() == len(a[0]) - 1
    # closed surface: every edge is shared by exactly two triangles
    e = np.sort(np.concatenate([a[1][:, [0, 1]], a[1][:, [1, 2]], a[1][:, [2, 0]]]), axis=1)
    _, cnt = np.unique(e, axis=0, return_counts=True)
    assert np.all(cnt == 2)
    fn = synth.face_normals(*a)
    cen = a[0].astype(np.float64)[a[1]].mean(1)
    ring = np.stack([cen[:, 0], cen[:, 1], 0 * cen[:, 2]], -1)
    ring = ring / np.linalg.norm(ring, axis=1, keepdims=True)   # centre circle of the torus
    assert np.all(((cen - ring) * fn).sum(1) > 0)               # normals point outward
    u = synth.u01(7, 3, 1000)
    assert np.array_equal(u, synth.u01(7, 3, 1000)) and 0 <= u.min() and u.max() < 1
    assert np.array_equal(synth.u01(7, 3, 10, start=5), u[5:15])


def test_line_intersection_known_answers(oracle):
    """vtkImplicitPolyDataIntersection::IntersectWithLine restated, on the unit cube: hits on a
    face interior, on the diagonal edge of a face (two coplanar faces: their common normal),
    on a corner (three faces at right angles: the normalised (1,1,1)), a miss, and a segment
    that starts inside; the returned value carries the sign the reference's line :324 gives
    it (always negated), NoValue / NoGradient / zero point without a hit."""
    v = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [0, 0, 1], [1, 0, 1], [1, 1, 1],
                  [0, 1, 1]], np.float32)
    t = np.array([[0, 2, 1], [0, 3, 2], [4, 5, 6], [4, 6, 7], [0, 1, 5], [0, 5, 4], [2, 3, 7],
                  [2, 7, 6], [1, 2, 6], [1, 6, 5], [0, 4, 7], [0, 7, 3]], np.int32)
    m = oracle.Mesh(v, t)
    p1 = np.array([[0.25, 0.5, 3.0], [0.5, 0.5, 3.0], [1, 1, 3.0], [5, 5, 5], [0.3, 0.2, 0.5],
                   [3.0, 0.25, 0.75]])
    p2 = np.array([[0.25, 0.5, -3.0], [0.5, 0.5, -3.0], [1, 1, -3.0], [6, 6, 6], [0.3, 0.2, 2.0],
                   [-3.0, 0.25, 0.75]])
    r = m.intersect_lines(p1, p2)
    s3 = 1 / np.sqrt(3)
    assert r["hit"].tolist() == [1, 1, 1, 0, 1, 1]
    assert np.allclose(r["ret"], [-2, -2, -2, np.sqrt(3), -0.5, -2], atol=1e-15)
    assert r["kind"].tolist() == [0, 1, 2, -1, 0, 0]
    assert r["cell"].tolist() == [3, 2, 2, -1, 2, 9]       # ties on the edge / corner: lowest id
    assert np.allclose(r["point"], [[0.25, 0.5, 1], [0.5, 0.5, 1], [1, 1, 1], [0, 0, 0],
                                    [0.3, 0.2, 1], [1, 0.25, 0.75]], atol=1e-15)
    assert np.allclose(r["closest"], r["point"], atol=1e-15)
    assert np.allclose(r["normal"], [[0, 0, 1], [0, 0, 1], [s3, s3, s3], [0, 0, 1], [0, 0, 1],
                                     [1, 0, 0]], atol=1e-15)
    # a segment that stops short of the surface, one parallel to a face inside its plane
    r = m.intersect_lines(np.array([[0.5, 0.5, 3.0], [-1.0, 0.5, 1.0]]),
                          np.array([[0.5, 0.5, 1.5], [2.0, 0.5, 1.0]]))
    assert r["hit"][0] == 0
    # (in the plane of the top face: parallel to it, vtkPlane::IntersectWithLine says no; the
    # side faces x = 0 and x = 1 are crossed at their upper edge)
    assert r["hit"][1] == 1 and np.allclose(r["point"][1], [0, 0.5, 1]) and r["kind"][1] == 1
