"""Synthetic meshes and scans for tests and benchmarks.

All randomness comes from a counter-based generator, splitmix64(seed ^ index), so
any process regenerates bit-identical data from (seed, sizes) alone and no large
fixture has to be stored.  The shapes stand in for the reference's data/Bunny.vtp
(a git-LFS pointer in the reference tree) and follow the perturbation used by
ICP/Testing/Python/RegistrationDemo.py:98-100 (translate (0,0,0.3), rotate Y 15 deg).
"""
import numpy as np

SEED = 20261111
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x):
    """Vectorised splitmix64 finaliser over uint64 counters."""
    with np.errstate(over="ignore"):
        x = (np.asarray(x, np.uint64) + np.uint64(0x9E3779B97F4A7C15)) & _M64
        x = ((x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        x = ((x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return x ^ (x >> np.uint64(31))


def u01(seed, stream, n, start=0):
    """n uniform doubles in [0,1) for (seed, stream), counters start..start+n-1."""
    idx = np.arange(start, start + n, dtype=np.uint64)
    key = splitmix64(np.uint64(seed) ^ (np.uint64(stream) << np.uint64(40)))
    bits = splitmix64(idx ^ key)
    return (bits >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def gaussian(seed, stream, n):
    """n standard normals by Box-Muller on two counter streams."""
    a = u01(seed, stream, n)
    b = u01(seed, stream + 1, n)
    return np.sqrt(-2.0 * np.log(1.0 - a)) * np.cos(2.0 * np.pi * b)


def rigid(rx=0.0, ry=0.0, rz=0.0, t=(0.0, 0.0, 0.0)):
    """Row-major 4x4 of p -> Rz*Ry*Rx*p + t; angles in degrees."""
    ax, ay, az = np.radians([rx, ry, rz])
    cx, sx, cy, sy, cz, sz = np.cos(ax), np.sin(ax), np.cos(ay), np.sin(ay), np.cos(az), np.sin(az)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    M = np.eye(4)
    M[:3, :3] = Rz @ Ry @ Rx
    M[:3, 3] = t
    return M


def demo_perturbation(scale=1.0):
    """vtkTransform.Translate(0,0,0.3); RotateY(15) (pre-multiply): p -> Tr * Ry * p."""
    return rigid(ry=15.0, t=(0.0, 0.0, 0.3 * scale))


def apply(M, xyz):
    """fp64 transform of points, rounded to fp32 (what vtkTransformPolyDataFilter stores)."""
    p = np.asarray(xyz, np.float64)
    return (p @ M[:3, :3].T + M[:3, 3]).astype(np.float32)


def bumpy_torus(nu, nv, seed=SEED, R=1.0, r=0.4, amp=0.08, grade=0.0):
    """Closed torus with a smooth seeded radial displacement.

    Returns (xyz float32 [nu*nv,3], tri int32 [2*nu*nv,3]); outward orientation.
    grade in [0, 1): the grid lines are drawn together on one side of the torus and apart on
    the other in both directions (parameter s -> s + grade sin s), so that edge lengths vary
    by (1 + grade) / (1 - grade) and the quads' shapes from squares to slivers, as the
    triangles of a decimated scan do.
    """
    k = 8
    a = u01(seed, 1, k) * (amp / k)
    m = 1 + np.floor(u01(seed, 2, k) * 6).astype(np.int64)
    n = 1 + np.floor(u01(seed, 3, k) * 4).astype(np.int64)
    ph = u01(seed, 4, k) * 2 * np.pi
    ps = u01(seed, 5, k) * 2 * np.pi
    u = (np.arange(nu, dtype=np.float64) * (2 * np.pi / nu))[:, None]
    v = (np.arange(nv, dtype=np.float64) * (2 * np.pi / nv))[None, :]
    if grade:
        u = u + grade * np.sin(u)
        v = v + grade * np.sin(v + 1.0)
    d = np.zeros((nu, nv))
    for i in range(k):
        d += a[i] * np.sin(m[i] * u + ph[i]) * np.sin(n[i] * v + ps[i])
    rr = r + d
    x = (R + rr * np.cos(v)) * np.cos(u)
    y = (R + rr * np.cos(v)) * np.sin(u)
    z = rr * np.sin(v) + 0.0 * u
    xyz = np.stack([x, y, z], -1).reshape(-1, 3).astype(np.float32)
    i = np.arange(nu, dtype=np.int64)[:, None]
    j = np.arange(nv, dtype=np.int64)[None, :]
    p00 = i * nv + j
    p10 = ((i + 1) % nu) * nv + j
    p11 = ((i + 1) % nu) * nv + (j + 1) % nv
    p01 = i * nv + (j + 1) % nv
    t0 = np.stack([p00, p10, p11], -1).reshape(-1, 3)
    t1 = np.stack([p00, p11, p01], -1).reshape(-1, 3)
    tri = np.empty((2 * nu * nv, 3), np.int32)
    tri[0::2] = t0
    tri[1::2] = t1
    return xyz, tri


def face_normals(xyz, tri):
    p = np.asarray(xyz, np.float64)
    a, b, c = p[tri[:, 0]], p[tri[:, 1]], p[tri[:, 2]]
    n = np.cross(b - a, c - a)
    ln = np.linalg.norm(n, axis=1, keepdims=True)
    return n / np.where(ln > 0, ln, 1.0)


def vertex_normals(xyz, tri):
    """Point normals as vtkPolyDataNormals builds them with splitting off:
    the normalised sum of the unit normals of the faces around each point."""
    fn = face_normals(xyz, tri)
    vn = np.zeros((len(xyz), 3))
    for k in range(3):
        np.add.at(vn, tri[:, k], fn)
    ln = np.linalg.norm(vn, axis=1, keepdims=True)
    return (vn / np.where(ln > 0, ln, 1.0)).astype(np.float32)


def sample_surface(xyz, tri, n, seed=SEED, stream=16):
    """n area-uniform surface samples: (points float32, face index int64, face normals float32)."""
    p = np.asarray(xyz, np.float64)
    a, b, c = p[tri[:, 0]], p[tri[:, 1]], p[tri[:, 2]]
    area = 0.5 * np.linalg.norm(np.cross(b - a, c - a), axis=1)
    cdf = np.cumsum(area)
    f = np.searchsorted(cdf, u01(seed, stream, n) * cdf[-1], side="right")
    f = np.minimum(f, len(tri) - 1)
    s1 = np.sqrt(u01(seed, stream + 1, n))
    r2 = u01(seed, stream + 2, n)
    w0, w1, w2 = (1 - s1)[:, None], (s1 * (1 - r2))[:, None], (s1 * r2)[:, None]
    pts = w0 * a[f] + w1 * b[f] + w2 * c[f]
    fn = face_normals(xyz, tri[f])
    return pts.astype(np.float32), f.astype(np.int64), fn.astype(np.float32)


def diag(xyz):
    p = np.asarray(xyz, np.float64)
    return float(np.linalg.norm(p.max(0) - p.min(0)))


# --------------------------------------------------------------------------- configurations
def config_c1(landmarks=100):
    """Bunny stand-in (4096 triangles) against a rigidly perturbed copy, point-to-point."""
    xyz, tri = bumpy_torus(64, 32)
    src = apply(demo_perturbation(), xyz)
    return dict(name="c1", target_xyz=xyz, target_tri=tri, source_xyz=src, source_normals=None,
                metric=0, max_landmarks=landmarks, max_dist=10.0, max_iter=50)


def config_c2(landmarks=100):
    """Same pair, point-to-plane with a distance threshold of 25 % of the diagonal."""
    c = config_c1(landmarks)
    c.update(name="c2", metric=1, max_dist=0.25 * diag(c["target_xyz"]))
    return c


def config_c3(nu=1000, nv=500, npts=1_000_000):
    """1M-triangle mesh, 1M area-uniform source samples moved by rotY 5 deg + 1 % of the
    diagonal, symmetric metric; source normals are the sampled faces' normals rotated with
    the points."""
    xyz, tri = bumpy_torus(nu, nv)
    pts, f, fn = sample_surface(xyz, tri, npts)
    d = diag(xyz)
    M = rigid(ry=5.0, t=(0.01 * d / np.sqrt(3),) * 3)
    src = apply(M, pts)
    sn = (fn.astype(np.float64) @ M[:3, :3].T).astype(np.float32)
    return dict(name="c3", target_xyz=xyz, target_tri=tri, source_xyz=src, source_normals=sn,
                metric=2, max_landmarks=npts, max_dist=10.0, max_iter=50, truth=M)


def config_c4(nu=2500, nv=2000, npts=16_000_000):
    """10M-triangle mesh, 16M noisy source samples (sigma = 0.1 % of the diagonal),
    point-to-plane, 50 iterations, every point used."""
    xyz, tri = bumpy_torus(nu, nv)
    pts, f, fn = sample_surface(xyz, tri, npts)
    d = diag(xyz)
    noise = np.stack([gaussian(SEED, 32 + 2 * k, npts) for k in range(3)], -1) * (1e-3 * d)
    M = rigid(ry=5.0, t=(0.01 * d / np.sqrt(3),) * 3)
    src = apply(M, pts.astype(np.float64) + noise)
    return dict(name="c4", target_xyz=xyz, target_tri=tri, source_xyz=src, source_normals=None,
                metric=1, max_landmarks=npts, max_dist=10.0, max_iter=50, truth=M)


def config_c5(nu=256, nv=128, npts=40_000, outlier_frac=0.2):
    """Partial overlap: the target keeps triangles whose centroid has x > -0.25, the scan
    keeps samples with x < +0.25 and adds uniform outliers in 1.5x the bounding box;
    threshold 2 % of the diagonal."""
    xyz, tri = bumpy_torus(nu, nv)
    cen = np.asarray(xyz, np.float64)[tri].mean(1)
    ttri = np.ascontiguousarray(tri[cen[:, 0] > -0.25])
    pts, f, fn = sample_surface(xyz, tri, npts)
    keep = pts[:, 0] < 0.25
    pts = pts[keep]
    nout = int(outlier_frac * len(pts))
    lo, hi = xyz.min(0).astype(np.float64), xyz.max(0).astype(np.float64)
    c, h = 0.5 * (lo + hi), 0.75 * (hi - lo)
    out = np.stack([u01(SEED, 48 + k, nout) for k in range(3)], -1) * 2 * h + (c - h)
    M = rigid(ry=2.0, t=(0.0, 0.0, 0.01))
    src = apply(M, np.concatenate([pts.astype(np.float64), out], 0))
    return dict(name="c5", target_xyz=xyz, target_tri=ttri, source_xyz=src, source_normals=None,
                metric=1, max_landmarks=len(src), max_dist=0.02 * diag(xyz), max_iter=30,
                truth=M)


def config_c6(nu=1000, nv=500, npts=1_000_000, grade=0.97):
    """A graded mesh: the torus of c3 with its grid lines drawn together on one side (edge
    lengths over a range of 65, areas of 4000, squares to slivers), area-uniform samples moved
    as in c3, point-to-plane."""
    xyz, tri = bumpy_torus(nu, nv, grade=grade)
    pts, f, fn = sample_surface(xyz, tri, npts)
    d = diag(xyz)
    M = rigid(ry=5.0, t=(0.01 * d / np.sqrt(3),) * 3)
    src = apply(M, pts)
    return dict(name="c6", target_xyz=xyz, target_tri=tri, source_xyz=src, source_normals=None,
                metric=1, max_landmarks=npts, max_dist=10.0, max_iter=50, truth=M)
