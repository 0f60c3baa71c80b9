"""Shared test helpers: a numpy restatement of the raw-moment layout the GPU accumulates
(so host logic can be exercised without a GPU) and comparison utilities."""
import numpy as np

NSUMS = 40


def sums_from_rows(metric, origin_pts, closest, normal, origin_normal, o):
    """fp64 raw moments of accepted float rows about the point o (layout of icpx_solve)."""
    s = np.asarray(origin_pts, np.float32).astype(np.float64).reshape(-1, 3)
    t = np.asarray(closest, np.float32).astype(np.float64).reshape(-1, 3)
    out = np.zeros(NSUMS)
    n = len(s)
    out[0] = n
    if n == 0:
        return out
    out[1] = np.sqrt(((s - t) ** 2).sum(1)).sum()
    sp, tp = s - o, t - o
    out[2:5] = sp.sum(0)
    if metric == 0:
        out[5:8] = tp.sum(0)
        out[8:17] = (sp[:, :, None] * tp[:, None, :]).sum(0).reshape(9)
        return out
    nn = np.asarray(normal, np.float32).astype(np.float64).reshape(-1, 3)
    u, b = sp, 5
    if metric == 2:
        out[5:8] = tp.sum(0)
        nn = nn + np.asarray(origin_normal, np.float32).astype(np.float64).reshape(-1, 3)
        u, b = sp + tp, 8
    a = np.cross(u, nn)
    r = ((s - t) * nn).sum(1)
    iu = [(0, 0), (0, 1), (0, 2), (1, 1), (1, 2), (2, 2)]
    out[b:b + 6] = [(a[:, i] * a[:, j]).sum() for i, j in iu]
    out[b + 6:b + 15] = (a[:, :, None] * nn[:, None, :]).sum(0).reshape(9)
    out[b + 15:b + 21] = [(nn[:, i] * nn[:, j]).sum() for i, j in iu]
    out[b + 21:b + 24] = (a * r[:, None]).sum(0)
    out[b + 24:b + 27] = (nn * r[:, None]).sum(0)
    return out


def rel_err(A, B):
    """max |A-B| relative to max |B| (the 1e-6 gate on 4x4 transforms)."""
    A, B = np.asarray(A, np.float64), np.asarray(B, np.float64)
    return float(np.abs(A - B).max() / max(np.abs(B).max(), 1e-300))


# ---------------------------------------------------------------------------------------
# Known answers worked out by hand (coordinates are exact in float), independent of both the
# oracle and the device code.  Each case: points, triangles, queries, and per query the
# expected closest point, distance, kind (0 face, 1 edge, 2 vertex), unit normal, cell id.
def hand_cases():
    s5 = np.sqrt(5.0)
    cases = []
    # (1) two faces meeting along the x axis at a dihedral that is not a right angle: the
    # first in z = 0 (normal +z), the second hanging down at cos = 0.6, sin = 0.8 (normal
    # (0, -0.8, 0.6)).  Edge normal = their normalised sum = (0, -1, 2) / sqrt(5).
    xyz = np.array([[0, 0, 0], [1, 0, 0], [0.5, 1, 0], [0.5, -0.75, -1.0]], np.float32)
    tri = np.array([[0, 1, 2], [1, 0, 3]], np.int32)
    q = np.array([[0.5, -0.125, 0.25],     # off the edge along the bisector of the two normals
                  [0.5, 0.5, 0.75],        # above the first face
                  [0.5, -0.875, -0.125]], np.float32)   # 0.625 off the second face's interior
    want = dict(closest=[[0.5, 0, 0], [0.5, 0.5, 0], [0.5, -0.375, -0.5]],
                dist=[0.125 * s5, 0.75, 0.625], kind=[1, 0, 0],
                normal=[[0, -1 / s5, 2 / s5], [0, 0, 1], [0, -0.8, 0.6]], cell=[0, 0, 1])
    cases.append(("dihedral", xyz, tri, q, want))
    # (2) a vertex with unequal incident angles: a right angle of a face in z = 0 (normal
    # +z) and 45 degrees of a face in y = 0 (normal -y).  Angle-weighted sum:
    # pi/2 (0,0,1) + pi/4 (0,-1,0)  ->  (0, -1, 2) / sqrt(5); equal weights would give
    # (0, -1, 1) / sqrt(2).
    xyz = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0], [1, 0, -1]], np.float32)
    tri = np.array([[0, 1, 2], [1, 0, 3]], np.int32)
    q = np.array([[-0.25, -0.125, 0.25]], np.float32)   # |q| = 0.375, nearest point: the vertex
    want = dict(closest=[[0, 0, 0]], dist=[0.375], kind=[2], normal=[[0, -1 / s5, 2 / s5]],
                cell=[0])
    cases.append(("vertex angles", xyz, tri, q, want))
    # (3) one triangle, a query in each region of vtkTriangle::EvaluatePosition outside it:
    # one negative weight (three edges), two negative weights (three corners)
    xyz = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], np.float32)
    tri = np.array([[0, 1, 2]], np.int32)
    q = np.array([[1, 1, 0.5], [-0.5, 0.5, 0.5], [0.5, -0.5, 0.5],
                  [-0.5, -0.5, 0.5], [2, -0.5, 0.5], [-0.5, 2, 0.5], [0.25, 0.25, -0.5]], np.float32)
    want = dict(closest=[[0.5, 0.5, 0], [0, 0.5, 0], [0.5, 0, 0], [0, 0, 0], [1, 0, 0], [0, 1, 0],
                         [0.25, 0.25, 0]],
                dist=np.sqrt([0.75, 0.5, 0.5, 0.75, 1.5, 1.5, 0.25]), kind=[1, 1, 1, 2, 2, 2, 0],
                normal=[[0, 0, 1]] * 7, cell=[0] * 7)
    cases.append(("regions", xyz, tri, q, want))
    return cases


# ---------------------------------------------------------------------------------------
# The reference's three solvers restated with numpy's own factorisations (SVD, solve), from
# the float rows the filter hands them: a third opinion beside icpx_solve and the oracle.
def numpy_solve(metric, s, t, ns=None, nt=None):
    """One icp::*::Update step in fp64: returns the 4x4."""
    s = np.asarray(s, np.float32).astype(np.float64)
    t = np.asarray(t, np.float32).astype(np.float64)
    M = np.eye(4)
    if metric == 0:
        # point2point.cxx:25-66: weighted means, sigma = S diag(w) T^T, R = V U^T (no reflections)
        w = np.full(len(s), 1.0 / len(s))
        sm, tm = (s * w[:, None]).sum(0), (t * w[:, None]).sum(0)
        sigma = ((s - sm) * w[:, None]).T @ (t - tm)
        U, _, Vt = np.linalg.svd(sigma)
        V = Vt.T
        if np.linalg.det(U) * np.linalg.det(V) < 0:
            V = V @ np.diag([1.0, 1.0, -1.0])
        R = V @ U.T
        M[:3, :3], M[:3, 3] = R, tm - R @ sm
        return M
    nt = np.asarray(nt, np.float32).astype(np.float64)
    if metric == 1:
        # point2plane.cxx:37-110: both clouds centred on the source mean, c = s x n,
        # [c; n][c; n]^T x = -((s - t).n) [c; n], R = Rx Ry Rz, t = x[3:] + mean - R mean
        sm = s.mean(0)
        sc, tc = s - sm, t - sm
        c = np.cross(sc, nt)
        X = np.concatenate([c, nt], 1)
        rhs = -(X * ((sc - tc) * nt).sum(1)[:, None]).sum(0)
        x = np.linalg.solve(X.T @ X, rhs)
        ca, sa, cb, sb, cg, sg = np.cos(x[0]), np.sin(x[0]), np.cos(x[1]), np.sin(x[1]), \
            np.cos(x[2]), np.sin(x[2])
        Rx = np.array([[1, 0, 0], [0, ca, -sa], [0, sa, ca]])
        Ry = np.array([[cb, 0, sb], [0, 1, 0], [-sb, 0, cb]])
        Rz = np.array([[cg, -sg, 0], [sg, cg, 0], [0, 0, 1]])
        R = Rx @ Ry @ Rz
        M[:3, :3], M[:3, 3] = R, x[3:] + sm - R @ sm
        return M
    # plane2plane.cxx:37-94: P, Q centred on their own means, n = n_p + n_q,
    # x = [(p + q) x n; n], sum x x^T y = -sum ((p - q).n) x, Rodrigues half rotation twice
    ns = np.asarray(ns, np.float32).astype(np.float64)
    pm, qm = s.mean(0), t.mean(0)
    p, q, n = s - pm, t - qm, ns + nt
    X = np.concatenate([np.cross(p + q, n), n], 1)
    y = np.linalg.solve(X.T @ X, -(X * ((p - q) * n).sum(1)[:, None]).sum(0))
    at, tt = y[:3], y[3:]
    theta = np.arctan(np.linalg.norm(at))
    a = at / np.linalg.norm(at)
    W = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    Ri = np.eye(3) + np.sin(theta) * W + (1 - np.cos(theta)) * (W @ W)
    M[:3, :3] = Ri @ Ri
    M[:3, 3] = Ri @ Ri @ (-pm) + Ri @ (np.cos(theta) * tt) + qm
    return M
