"""ctypes front end of the CPU oracle (oracle/icp_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, by __graft_entry__.smoke() and by
bench.py's cpu_baseline leg, never by the iterativeclosestpoint_amd package.

PARITY UNPINNED: the reference ships no golden vectors for this path and its
arithmetic lives in VTK >= 9.3, which is not installed; see icp_oracle.c.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libicp_oracle.so")

_f32p = np.ctypeslib.ndpointer(np.float32, flags="C_CONTIGUOUS")
_f64p = np.ctypeslib.ndpointer(np.float64, flags="C_CONTIGUOUS")
_i32p = np.ctypeslib.ndpointer(np.int32, flags="C_CONTIGUOUS")
_i64p = np.ctypeslib.ndpointer(np.int64, flags="C_CONTIGUOUS")


def build():
    """Compile the oracle with gcc (a few seconds)."""
    subprocess.check_call(["make", "-s", "-C", _HERE])


def _load():
    if not os.path.exists(_SO):
        build()
    lib = C.CDLL(_SO)
    lib.orc_mesh_create.restype = C.c_void_p
    lib.orc_mesh_create.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]
    lib.orc_mesh_destroy.argtypes = [C.c_void_p]
    lib.orc_tri_eval.restype = C.c_int
    lib.orc_tri_eval.argtypes = [_f64p, _f64p, _f64p, _f64p, _f64p, _f64p, _f64p]
    for name in ("orc_find_closest_brute", "orc_find_closest_grid"):
        f = getattr(lib, name)
        f.restype = None
        f.argtypes = [C.c_void_p, _f64p, _f64p, _i64p, _f64p]
    lib.orc_closest_point_and_normal.restype = C.c_double
    lib.orc_closest_point_and_normal.argtypes = [
        C.c_void_p, _f64p, C.c_int, _f64p, _f64p, _i64p, _i32p]
    lib.orc_correspond.restype = C.c_int64
    lib.orc_correspond.argtypes = [
        C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, _f64p, C.c_int64, C.c_double, C.c_double,
        C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
        C.c_void_p, C.c_void_p, C.c_int]
    lib.orc_correspond_f64.restype = C.c_int64
    lib.orc_correspond_f64.argtypes = [
        C.c_void_p, C.c_void_p, C.c_int64, _f64p, C.c_int64, C.c_double, C.c_int, C.c_int,
        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    lib.orc_intersect_lines.restype = None
    lib.orc_intersect_lines.argtypes = [C.c_void_p, _f64p, _f64p, C.c_int64, C.c_int, _i32p, _f64p,
                                        _f64p, _f64p, _f64p, _i32p, _i32p]
    lib.orc_solve.restype = C.c_int
    lib.orc_solve.argtypes = [C.c_int, C.c_int, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                              C.c_void_p, _f64p]
    lib.orc_icp_run.restype = C.c_int
    lib.orc_icp_run.argtypes = [
        C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_int64, C.c_double,
        C.c_int, C.c_double, C.c_double, C.c_int, C.c_int, C.c_int, _f64p,
        C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_int64), C.c_int]
    lib.orc_num_threads.restype = C.c_int
    return lib


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load()
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def num_threads():
    return int(lib().orc_num_threads())


def tri_eval(x, p0, p1, p2):
    """vtkTriangle::EvaluatePosition restated: returns (status, closest, dist2, weights)."""
    x, p0, p1, p2 = (np.ascontiguousarray(a, np.float64) for a in (x, p0, p1, p2))
    cp = np.zeros(3)
    d2 = np.zeros(1)
    w = np.zeros(3)
    st = lib().orc_tri_eval(x, p0, p1, p2, cp, d2, w)
    return st, cp, float(d2[0]), w


class Mesh:
    """Target mesh as the locator sees it (triangles only, float points)."""

    def __init__(self, xyz, tri, cell_normals=None):
        self.xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
        self.tri = np.ascontiguousarray(tri, np.int32).reshape(-1, 3)
        self.cn = None if cell_normals is None else np.ascontiguousarray(
            cell_normals, np.float32).reshape(-1, 3)
        self.h = lib().orc_mesh_create(_ptr(self.xyz), len(self.xyz), _ptr(self.tri),
                                       len(self.tri), _ptr(self.cn))

    def __del__(self):
        # (at interpreter shutdown the module's globals may be gone before the object)
        if getattr(self, "h", None) and callable(lib):
            lib().orc_mesh_destroy(self.h)
            self.h = None

    def find_closest(self, x, grid=False):
        x = np.ascontiguousarray(x, np.float64)
        p = np.zeros(3)
        c = np.zeros(1, np.int64)
        d2 = np.zeros(1)
        f = lib().orc_find_closest_grid if grid else lib().orc_find_closest_brute
        f(self.h, x, p, c, d2)
        return p, int(c[0]), float(d2[0])

    def closest_point_and_normal(self, x, grid=False):
        x = np.ascontiguousarray(x, np.float64)
        n = np.zeros(3)
        cp = np.zeros(3)
        c = np.zeros(1, np.int64)
        k = np.zeros(1, np.int32)
        sd = lib().orc_closest_point_and_normal(self.h, x, int(grid), n, cp, c, k)
        return sd, n, cp, int(c[0]), int(k[0])

    def intersect_lines(self, p1, p2, threads=None):
        """vtkImplicitPolyDataIntersection::IntersectWithLine restated, for n segments."""
        p1 = np.ascontiguousarray(p1, np.float64).reshape(-1, 3)
        p2 = np.ascontiguousarray(p2, np.float64).reshape(-1, 3)
        n = len(p1)
        out = dict(hit=np.zeros(n, np.int32), ret=np.zeros(n), closest=np.zeros((n, 3)),
                   normal=np.zeros((n, 3)), point=np.zeros((n, 3)), cell=np.zeros(n, np.int32),
                   kind=np.zeros(n, np.int32))
        lib().orc_intersect_lines(self.h, p1.reshape(-1), p2.reshape(-1), n,
                                  int(threads or num_threads()), out["hit"], out["ret"],
                                  out["closest"].reshape(-1), out["normal"].reshape(-1),
                                  out["point"].reshape(-1), out["cell"], out["kind"])
        return out

    def correspond(self, src, T=None, step=1, max_dist=10.0, src_normals=None, min_dot=0.0,
                   grid=True, threads=None, rotate_normals=False):
        """vtkPolyDataCorrespondenceFilter restated; rows in index order.  rotate_normals: the
        opt-in corrected mode (source normals rotated by T; the reference does not)."""
        src = np.ascontiguousarray(src, np.float32).reshape(-1, 3)
        sn = None if src_normals is None else np.ascontiguousarray(
            src_normals, np.float32).reshape(-1, 3)
        T = np.eye(4) if T is None else np.ascontiguousarray(T, np.float64).reshape(4, 4)
        step = max(1, int(step))
        nk = (len(src) + step - 1) // step
        out = dict(
            origin=np.zeros((nk, 3), np.float32), origin_normal=np.zeros((nk, 3), np.float32),
            closest=np.zeros((nk, 3), np.float32), normal=np.zeros((nk, 3), np.float32),
            cell=np.zeros(nk, np.int32), dist=np.zeros(nk, np.float64),
            src_index=np.zeros(nk, np.int64), kind=np.zeros(nk, np.int32))
        n = lib().orc_correspond(
            self.h, _ptr(src), len(src), _ptr(sn), T.reshape(-1), step, float(max_dist),
            float(min_dot), int(grid), int(threads or num_threads()), _ptr(out["origin"]),
            _ptr(out["origin_normal"]), _ptr(out["closest"]), _ptr(out["normal"]),
            _ptr(out["cell"]), _ptr(out["dist"]), _ptr(out["src_index"]), _ptr(out["kind"]),
            int(bool(rotate_normals)))
        return {k: v[:n] for k, v in out.items()}

    def correspond_f64(self, src, T=None, step=1, max_dist=10.0, grid=True, threads=None):
        """The filter with OutputPrecision = DOUBLE_PRECISION; rows in index order."""
        src = np.ascontiguousarray(src, np.float32).reshape(-1, 3)
        T = np.eye(4) if T is None else np.ascontiguousarray(T, np.float64).reshape(4, 4)
        step = max(1, int(step))
        nk = (len(src) + step - 1) // step
        out = dict(origin=np.zeros((nk, 3), np.float64), closest=np.zeros((nk, 3), np.float64),
                   normal=np.zeros((nk, 3), np.float64), cell=np.zeros(nk, np.int32),
                   dist=np.zeros(nk, np.float64), src_index=np.zeros(nk, np.int64))
        n = lib().orc_correspond_f64(
            self.h, _ptr(src), len(src), T.reshape(-1), step, float(max_dist), int(grid),
            int(threads or num_threads()), _ptr(out["origin"]), _ptr(out["closest"]),
            _ptr(out["normal"]), _ptr(out["cell"]), _ptr(out["dist"]), _ptr(out["src_index"]))
        return {k: v[:n] for k, v in out.items()}

    def icp(self, src, metric=1, max_iter=50, max_landmarks=200, max_dist=10.0,
            check_mean=False, max_mean=0.01, src_normals=None, min_dot=0.0, precision=64,
            grid=True, threads=None, rotate_normals=False):
        """vtkICP::InternalUpdate restated. Returns (T, iterations, mean_distance, status, n)."""
        src = np.ascontiguousarray(src, np.float32).reshape(-1, 3)
        sn = None if src_normals is None else np.ascontiguousarray(
            src_normals, np.float32).reshape(-1, 3)
        T = np.zeros(16)
        it = C.c_int(0)
        md = C.c_double(0)
        ln = C.c_int64(0)
        st = lib().orc_icp_run(
            self.h, _ptr(src), len(src), _ptr(sn), int(metric), int(max_iter),
            int(max_landmarks), float(max_dist), int(check_mean), float(max_mean),
            float(min_dot), int(precision), int(grid), int(threads or num_threads()), T,
            C.byref(it), C.byref(md), C.byref(ln), int(bool(rotate_normals)))
        return T.reshape(4, 4), it.value, md.value, st, ln.value


def consistent_polys(polys):
    """vtkPolyDataNormals' ConsistencyOn pass (ICP/vtkPolyDataCorrespondenceFilter.cxx:358),
    restated from VTK 9.3's vtkPolyDataNormals::TraverseAndOrder: from every cell not yet
    visited, lowest id first, a wave spreads across shared edges; a neighbour that runs along
    the shared edge in the same direction is reversed (vtkPolyData::ReverseCell).  Only edges
    with exactly one other cell are walked across.  Returns (polys, number of reversals)."""
    polys = np.array(polys, np.int64).reshape(-1, 3)
    nt = len(polys)
    cells_of = {}
    for c in range(nt):
        for v in set(polys[c].tolist()):
            cells_of.setdefault(v, []).append(c)
    visited = np.zeros(nt, bool)
    flips = 0
    for seed in range(nt):
        if visited[seed]:
            continue
        visited[seed] = True
        wave = [seed]
        while wave:
            wave2 = []
            for c in wave:
                pts = polys[c]
                for j in range(3):
                    a, b = int(pts[j]), int(pts[(j + 1) % 3])
                    if a == b:
                        continue
                    nei = [o for o in cells_of[a] if o != c and b in polys[o]]
                    if len(nei) != 1 or visited[nei[0]]:
                        continue
                    o = nei[0]
                    npts = polys[o].tolist()
                    pos = npts.index(b)
                    if npts[(pos + 1) % 3] != a:
                        polys[o] = polys[o][::-1].copy()
                        flips += 1
                    visited[o] = True
                    wave2.append(o)
            wave = wave2
    return polys, flips


def point_normals(points, polys, consistency=True):
    """Point normals as the filter's own vtkPolyDataNormals yields them with splitting off and
    consistency on (ICP/vtkPolyDataCorrespondenceFilter.cxx:356-359, run at :457-464): the
    normalised sum of the unit normals of the incident faces, one term per corner; zero where
    nothing adds up."""
    p = np.asarray(points, np.float32).astype(np.float64)
    polys = np.asarray(polys).reshape(-1, 3)
    if consistency:
        polys = consistent_polys(polys)[0]
    a, b, c = p[polys[:, 0]], p[polys[:, 1]], p[polys[:, 2]]
    fn = np.cross(b - a, c - a)
    ln = np.linalg.norm(fn, axis=1, keepdims=True)
    fn = fn / np.where(ln > 0, ln, 1.0)
    vn = np.zeros_like(p)
    for k in range(3):
        np.add.at(vn, polys[:, k], fn)
    ln = np.linalg.norm(vn, axis=1, keepdims=True)
    return (vn / np.where(ln > 0, ln, 1.0)).astype(np.float32)


def solve(metric, s, t, ns=None, nt=None, precision=64):
    """One solver step on fp32 rows; returns (M 4x4, status)."""
    s = np.ascontiguousarray(s, np.float32).reshape(-1, 3)
    t = np.ascontiguousarray(t, np.float32).reshape(-1, 3)
    ns = None if ns is None else np.ascontiguousarray(ns, np.float32).reshape(-1, 3)
    nt = None if nt is None else np.ascontiguousarray(nt, np.float32).reshape(-1, 3)
    M = np.zeros(16)
    st = lib().orc_solve(int(metric), int(precision), len(s), _ptr(s), _ptr(t), _ptr(ns),
                         _ptr(nt), M)
    return M.reshape(4, 4), st
