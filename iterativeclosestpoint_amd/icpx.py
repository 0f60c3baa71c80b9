"""ctypes binding of libicpx (include/icpx.h).

The library is the product; this module only marshals numpy arrays into the C
ABI.  There is no CPU fallback: a missing library or a missing GPU raises.
"""
import ctypes as C
import os

import numpy as np

NSUMS = 40
METRIC_POINT_TO_POINT = 0
METRIC_POINT_TO_PLANE = 1
METRIC_PLANE_TO_PLANE = 2

E_INVALID, E_HIP, E_NOMEM, E_FEW = -1, -2, -3, -4
RUN_MAX_ITERATIONS, RUN_CONVERGED, RUN_TOO_FEW, RUN_DIVERGED = 0, 1, 2, 3

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ICPX_LIB") or os.path.join(_HERE, "libicpx.so")  # (ICPX_LIB: development builds)

# entry points declared in include/icpx.h (the boundary a binding of the reference's classes uses)
SYMBOLS = (
    "icpx_last_error", "icpx_device_count", "icpx_create", "icpx_destroy", "icpx_set_stream",
    "icpx_synchronize", "icpx_set_target", "icpx_set_source", "icpx_set_shard", "icpx_get_origin",
    "icpx_accumulate", "icpx_accumulate_static", "icpx_correspond", "icpx_correspond_f64", "icpx_solve", "icpx_default_params",
    "icpx_set_allreduce", "icpx_run",
    "icpx_run_begin", "icpx_run_enqueue", "icpx_run_advance", "icpx_run_update", "icpx_run_end",
    "icpx_point_normals", "icpx_intersect_lines",
)
# ... and in include/icpx_dev.h (kernel variants, diagnostics, timing)
DEV_SYMBOLS = (
    "icpx_last_build_seconds", "icpx_set_group_width", "icpx_reset_hints", "icpx_set_timing",
    "icpx_get_timing", "icpx_get_pass_timings", "icpx_set_exhaustive", "icpx_set_leaf_size",
    "icpx_set_persistent", "icpx_debug_dump", "icpx_set_local", "icpx_set_refill",
    "icpx_set_finalize", "icpx_set_lean", "icpx_set_anchor", "icpx_set_shard_order",
    "icpx_set_certify_min", "icpx_set_wave_blocks", "icpx_set_query_curve", "icpx_set_noisy_policy", "icpx_set_seed",
    "icpx_set_weigh", "icpx_set_bail", "icpx_set_step_budget", "icpx_set_defer",
)


class IcpxError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("icpx error %d: %s" % (code, msg))
        self.code = code


class Params(C.Structure):
    _fields_ = [("metric", C.c_int), ("max_iterations", C.c_int), ("max_landmarks", C.c_int64),
                ("max_distance", C.c_double), ("check_mean_distance", C.c_int),
                ("max_mean_distance", C.c_double), ("min_normal_dot", C.c_double),
                ("rotate_source_normals", C.c_int)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.POINTER(C.c_double), C.c_int, C.c_void_p)

_lib = None


def load():
    """Load libicpx.so from the package directory (built by __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("libicpx.so is not built: run `make -C iterativeclosestpoint_amd/csrc` "
                          "(or __graft_entry__.build()); there is no CPU fallback")
    lib = C.CDLL(LIB_PATH)
    vp, i64, dbl, ci = C.c_void_p, C.c_int64, C.c_double, C.c_int
    lib.icpx_last_error.restype = C.c_char_p
    lib.icpx_device_count.restype = ci
    lib.icpx_create.argtypes = [C.POINTER(vp), ci]
    lib.icpx_destroy.argtypes = [vp]
    lib.icpx_destroy.restype = None
    lib.icpx_set_stream.argtypes = [vp, vp]
    lib.icpx_synchronize.argtypes = [vp]
    lib.icpx_set_target.argtypes = [vp, vp, i64, vp, i64, vp]
    lib.icpx_set_source.argtypes = [vp, vp, i64, vp]
    lib.icpx_point_normals.argtypes = [vp, vp, i64, vp, i64, vp]
    lib.icpx_set_shard.argtypes = [vp, ci, ci]
    lib.icpx_get_origin.argtypes = [vp, vp]
    lib.icpx_accumulate.argtypes = [vp, ci, vp, i64, dbl, dbl, ci, vp, vp]
    lib.icpx_accumulate_static.argtypes = [vp, ci, vp, i64, dbl, dbl, ci, vp]
    lib.icpx_correspond.argtypes = [vp, vp, i64, dbl, dbl, ci, vp, vp, vp, vp, vp, vp, vp,
                                    C.POINTER(i64)]
    lib.icpx_correspond_f64.argtypes = [vp, vp, i64, dbl, vp, vp, vp, vp, vp, vp, C.POINTER(i64)]
    lib.icpx_intersect_lines.argtypes = [vp, vp, vp, i64, vp, vp, vp, vp, vp, vp, vp]
    lib.icpx_solve.argtypes = [ci, vp, vp, vp, C.POINTER(i64), C.POINTER(dbl)]
    lib.icpx_default_params.argtypes = [C.POINTER(Params)]
    lib.icpx_default_params.restype = None
    lib.icpx_set_allreduce.argtypes = [vp, ALLREDUCE_FN, vp]
    lib.icpx_run.argtypes = [vp, C.POINTER(Params), vp, C.POINTER(ci), C.POINTER(dbl),
                             C.POINTER(ci)]
    lib.icpx_run_begin.argtypes = [vp, C.POINTER(Params)]
    lib.icpx_run_enqueue.argtypes = [vp, vp, C.POINTER(ci)]
    lib.icpx_run_advance.argtypes = [vp, vp, C.POINTER(ci)]
    lib.icpx_run_update.argtypes = [vp, vp]
    lib.icpx_set_shard_order.argtypes = [vp, ci]
    lib.icpx_set_certify_min.argtypes = [vp, i64]
    lib.icpx_set_wave_blocks.argtypes = [vp, ci]
    lib.icpx_set_query_curve.argtypes = [vp, ci]
    lib.icpx_set_noisy_policy.argtypes = [vp, ci]
    lib.icpx_set_seed.argtypes = [vp, ci]
    lib.icpx_set_weigh.argtypes = [vp, ci]
    lib.icpx_run_end.argtypes = [vp, vp, C.POINTER(ci), C.POINTER(dbl), C.POINTER(ci)]
    lib.icpx_last_build_seconds.argtypes = [vp]
    lib.icpx_last_build_seconds.restype = dbl
    lib.icpx_set_group_width.argtypes = [vp, ci]
    lib.icpx_reset_hints.argtypes = [vp]
    lib.icpx_set_exhaustive.argtypes = [vp, ci]
    lib.icpx_set_leaf_size.argtypes = [vp, ci]
    lib.icpx_set_persistent.argtypes = [vp, ci]
    lib.icpx_set_refill.argtypes = [vp, ci]
    lib.icpx_set_bail.argtypes = [vp, ci]
    lib.icpx_set_step_budget.argtypes = [vp, ci]
    lib.icpx_set_defer.argtypes = [vp, ci]
    lib.icpx_set_finalize.argtypes = [vp, ci]
    lib.icpx_set_lean.argtypes = [vp, ci]
    lib.icpx_set_anchor.argtypes = [vp, ci]
    lib.icpx_set_local.argtypes = [vp, ci]
    lib.icpx_debug_dump.argtypes = [vp, ci, vp, i64]
    lib.icpx_set_timing.argtypes = [vp, ci]
    lib.icpx_get_timing.argtypes = [vp, C.POINTER(dbl), C.POINTER(i64), C.POINTER(dbl),
                                    C.POINTER(i64)]
    lib.icpx_get_pass_timings.argtypes = [vp, vp, vp, vp, i64, C.POINTER(i64)]
    _lib = lib
    return lib


def _chk(rc):
    if rc != 0:
        raise IcpxError(rc, load().icpx_last_error().decode())


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def device_count():
    return int(load().icpx_device_count())


def default_params():
    p = Params()
    load().icpx_default_params(C.byref(p))
    return p


def solve(metric, sums, origin):
    """Host solve of one iteration: returns (M 4x4, n, mean_distance, rc) with rc 0 or E_FEW."""
    sums = np.ascontiguousarray(sums, np.float64)
    origin = np.ascontiguousarray(origin, np.float64)
    assert sums.size == NSUMS and origin.size == 3
    M = np.zeros(16)
    n = C.c_int64(0)
    md = C.c_double(0)
    rc = load().icpx_solve(int(metric), _p(sums), _p(origin), _p(M), C.byref(n), C.byref(md))
    if rc not in (0, E_FEW):
        _chk(rc)
    return M.reshape(4, 4), n.value, md.value, rc


class Context:
    """One GPU's share of a registration (icpx_ctx)."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        self._cb = None
        _chk(load().icpx_create(C.byref(self._h), int(device)))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            load().icpx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def set_stream(self, stream_ptr):
        _chk(load().icpx_set_stream(self._h, C.c_void_p(stream_ptr or 0)))

    def synchronize(self):
        _chk(load().icpx_synchronize(self._h))

    def set_local(self, enable=True):
        """True/1: the shortcut with certificates; 2: without certificates; 3: as 1, the
        certificates always checked by their own kernel; False/0: tree only."""
        _chk(load().icpx_set_local(self._h, int(enable)))

    def debug_stats(self):
        """{shortcut passes, tree-only passes, neighbour-list builds, list radius}."""
        v = np.zeros(16, np.int64)
        _chk(load().icpx_debug_dump(self._h, 4, _p(v), 16))
        return dict(local_passes=int(v[0]), tree_passes=int(v[1]), nbr_builds=int(v[2]),
                    nbr_eps=float(v[3]) * 1e-9, nbr_total=int(v[4]), certify_passes=int(v[5]),
                    split_passes=int(v[6]), split_wave=int(v[7]), split_pool=int(v[8]),
                    winding_flips=int(v[9]), direct_passes=int(v[10]),
                    lean_passes=int(v[11]), lean_failures=int(v[12]),
                    unorientable=bool(v[13]), mesh_grading=float(v[14]) * 1e-3,
                    bail_passes=int(v[15]))

    def debug_counters(self):
        """Counters of the last lane-per-query pass: {ties, overflows, pool position, fallbacks}."""
        v = np.zeros(4, np.uint32)
        _chk(load().icpx_debug_dump(self._h, 2, _p(v), 4))
        return dict(ties=int(v[0]), overflows=int(v[1]), pool=int(v[2]), fallback=int(v[3]))

    def debug_certificates(self, n):
        """The shortcut's certificates of the first n queries (in the library's query order)."""
        v = np.zeros(int(n), np.float32)
        _chk(load().icpx_debug_dump(self._h, 5, _p(v), int(n)))
        return v

    def debug_leads(self, n):
        """Per query on a set: its winner's lead over the set's other members (valid where the
        certificate's two low mantissa bits are set)."""
        v = np.zeros(int(n), np.float32)
        _chk(load().icpx_debug_dump(self._h, 9, _p(v), int(n)))
        return v

    def debug_query_order(self, n):
        """Position in the library's query order -> kept index within the shard."""
        v = np.zeros(int(n), np.int32)
        _chk(load().icpx_debug_dump(self._h, 6, _p(v), int(n)))
        return v

    def debug_cell_ids(self, nt):
        """Morton position of a triangle -> its cell id (-1 - id for a degenerate one)."""
        v = np.zeros(int(nt), np.int32)
        _chk(load().icpx_debug_dump(self._h, 1, _p(v), int(nt)))
        return v

    def debug_contender_sets(self, n):
        """(n, 8) Morton positions of the contenders kept for the first n queries (in the
        library's order); -1 ends a shorter set."""
        v = np.zeros((int(n), 8), np.int32)
        _chk(load().icpx_debug_dump(self._h, 7, _p(v), int(n) * 8))
        return v

    def set_persistent(self, mode):
        """Tree kernel: 1 pooled lanes, 3 one walk per wavefront, 4 by regime (default)."""
        _chk(load().icpx_set_persistent(self._h, int(mode)))

    def set_refill(self, lanes):
        _chk(load().icpx_set_refill(self._h, int(lanes)))

    def set_bail(self, microseconds):
        """Microseconds after which a whole pass of the wavefront walk gives up on a batch:
        1 by the cloud's size (default), 0 never, else that many."""
        _chk(load().icpx_set_bail(self._h, int(microseconds)))

    def set_step_budget(self, steps):
        """Steps after which the pooled lanes give up on a query (0: never; default 512)."""
        _chk(load().icpx_set_step_budget(self._h, int(steps)))

    def set_defer(self, edges):
        """Edge lengths beyond which the wavefront walk sets a query aside that is far out
        where its batch is not (0: never, the default)."""
        _chk(load().icpx_set_defer(self._h, int(edges)))

    def set_finalize(self, build):
        """0: by size; 1: sums in registers; 2: sums on the matrix core."""
        _chk(load().icpx_set_finalize(self._h, int(build)))

    def set_anchor(self, enable=True):
        """Settled passes of the device-resident loop leave the certificates as they are
        (held against the transform of the last pass that brought them up to date); default
        on.  Same results."""
        _chk(load().icpx_set_anchor(self._h, int(bool(enable))))

    def set_lean(self, enable=True):
        """Settled passes of icpx_run on one rank as two launches (certificate check inside the
        fused finalize); default on; 2: tried at every chance, 3: and every third made to fail (tests).  Same results."""
        _chk(load().icpx_set_lean(self._h, int(enable)))

    def set_leaf_size(self, triangles):
        _chk(load().icpx_set_leaf_size(self._h, int(triangles)))

    def set_exhaustive(self, enable=True):
        _chk(load().icpx_set_exhaustive(self._h, int(bool(enable))))

    def reset_hints(self):
        _chk(load().icpx_reset_hints(self._h))

    def set_timing(self, enable=True):
        """True/1: events around the search and the finalize kernels; 2: the search only."""
        _chk(load().icpx_set_timing(self._h, int(enable)))

    def get_timing(self):
        """(search_ms, search_launches, finalize_ms, finalize_launches) since set_timing."""
        a, b = C.c_double(0), C.c_double(0)
        na, nb = C.c_int64(0), C.c_int64(0)
        _chk(load().icpx_get_timing(self._h, C.byref(a), C.byref(na), C.byref(b), C.byref(nb)))
        return a.value, na.value, b.value, nb.value

    def get_pass_timings(self):
        """Search-kernel milliseconds of every pass timed since set_timing, oldest first:
        (ms float64[n], kind int32[n]: 1 = the hinted shortcut first, 2 = a lean pass,
        hinted int32[n])."""
        n = C.c_int64(0)
        _chk(load().icpx_get_pass_timings(self._h, None, None, None, 0, C.byref(n)))
        ms = np.zeros(n.value, np.float64)
        kind = np.zeros(n.value, np.int32)
        hinted = np.zeros(n.value, np.int32)
        _chk(load().icpx_get_pass_timings(self._h, _p(ms), _p(kind), _p(hinted), n.value,
                                          C.byref(n)))
        return ms, kind, hinted

    def set_group_width(self, lanes):
        _chk(load().icpx_set_group_width(self._h, int(lanes)))

    def set_target(self, xyz, tri, cell_normals=None):
        xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
        tri = np.ascontiguousarray(tri, np.int32).reshape(-1, 3)
        cn = None
        if cell_normals is not None:
            cn = np.ascontiguousarray(cell_normals, np.float32).reshape(-1, 3)
            if len(cn) != len(tri):
                raise ValueError("cell_normals must have one row per triangle")
        _chk(load().icpx_set_target(self._h, _p(xyz), len(xyz), _p(tri), len(tri), _p(cn)))

    def point_normals(self, xyz, tri):
        """Per-point normals of a triangle mesh (normalised sum of the unit normals of the faces
        at each point, after cells wound against their piece's lowest cell have been reversed),
        computed on the device: what the filter's own vtkPolyDataNormals yields with splitting
        off and consistency on (ICP/vtkPolyDataCorrespondenceFilter.cxx:356-359,457-464)."""
        xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
        tri = np.ascontiguousarray(tri, np.int32).reshape(-1, 3)
        out = np.zeros_like(xyz)
        _chk(load().icpx_point_normals(self._h, _p(xyz), len(xyz), _p(tri), len(tri), _p(out)))
        return out

    def set_source(self, xyz, normals=None):
        xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
        sn = None
        if normals is not None:
            sn = np.ascontiguousarray(normals, np.float32).reshape(-1, 3)
            if len(sn) != len(xyz):
                raise ValueError("normals must have one row per point")
        self._np = len(xyz)
        _chk(load().icpx_set_source(self._h, _p(xyz), len(xyz), _p(sn)))

    def set_shard(self, rank, world):
        _chk(load().icpx_set_shard(self._h, int(rank), int(world)))

    def set_shard_order(self, order):
        """1 (default): shards are ranges of the kept points' order along the space-filling
        curve; 0: blocks of consecutive kept indices."""
        _chk(load().icpx_set_shard_order(self._h, int(order)))

    def set_wave_blocks(self, blocks_per_cu):
        _chk(load().icpx_set_wave_blocks(self._h, int(blocks_per_cu)))

    def set_seed(self, mode):
        """Cold seeds of the wavefront walk: 0 Morton curve, 1 grid and curve (default), 2 grid."""
        _chk(load().icpx_set_seed(self._h, int(mode)))

    def set_weigh(self, mode):
        """Certified sets: bit 0, weighed by the matrix-core finalize instead of k_contested;
        bit 1, the winners' leads are kept and clear winners not weighed at all.  Default 3."""
        _chk(load().icpx_set_weigh(self._h, int(mode)))

    def set_noisy_policy(self, enable):
        _chk(load().icpx_set_noisy_policy(self._h, int(enable)))

    def set_query_curve(self, hilbert):
        _chk(load().icpx_set_query_curve(self._h, int(hilbert)))

    def set_certify_min(self, queries):
        _chk(load().icpx_set_certify_min(self._h, int(queries)))

    @property
    def origin(self):
        o = np.zeros(3)
        _chk(load().icpx_get_origin(self._h, _p(o)))
        return o

    @property
    def build_seconds(self):
        return float(load().icpx_last_build_seconds(self._h))

    def accumulate(self, metric, T, step=1, max_dist=10.0, min_dot=0.0, use_source_normals=None,
                   device_ptr=None):
        """Fused search + partial sums.  Returns the sums (numpy) unless device_ptr is given,
        in which case they are left at that device address, stream-ordered.
        use_source_normals: False/0, True/1 (as given), 2 (rotated by T: the corrected mode)."""
        T = np.ascontiguousarray(T, np.float64).reshape(16)
        if use_source_normals is None:
            use_source_normals = metric == METRIC_PLANE_TO_PLANE
        if device_ptr:
            _chk(load().icpx_accumulate(self._h, int(metric), _p(T), int(step), float(max_dist),
                                        float(min_dot), int(use_source_normals), None,
                                        C.c_void_p(device_ptr)))
            return None
        sums = np.zeros(NSUMS)
        _chk(load().icpx_accumulate(self._h, int(metric), _p(T), int(step), float(max_dist),
                                    float(min_dot), int(use_source_normals), _p(sums), None))
        return sums

    def accumulate_static(self, metric, T, step, max_dist, min_dot, use_source_normals,
                          device_ptr):
        """icpx_accumulate_static: the same sums left at `device_ptr`, with no allocation (the
        scratch is a static device array in the library; at most 32768 kept points)."""
        T = np.ascontiguousarray(T, np.float64).reshape(16)
        _chk(load().icpx_accumulate_static(self._h, int(metric), _p(T), int(step),
                                           float(max_dist), float(min_dot),
                                           int(use_source_normals), C.c_void_p(device_ptr)))

    def correspond(self, T=None, step=1, max_dist=10.0, min_dot=0.0, use_source_normals=False):
        """The correspondence filter's outputs: accepted rows in ascending source index."""
        T = np.eye(4) if T is None else np.asarray(T, np.float64)
        T = np.ascontiguousarray(T).reshape(16)
        step = max(1, int(step))
        nk = (self._np + step - 1) // step
        out = dict(origin=np.zeros((nk, 3), np.float32),
                   origin_normal=np.zeros((nk, 3), np.float32),
                   closest=np.zeros((nk, 3), np.float32), normal=np.zeros((nk, 3), np.float32),
                   cell=np.zeros(nk, np.int32), dist=np.zeros(nk, np.float64),
                   src_index=np.zeros(nk, np.int64))
        n = C.c_int64(0)
        _chk(load().icpx_correspond(
            self._h, _p(T), step, float(max_dist), float(min_dot), int(use_source_normals),
            _p(out["origin"]), _p(out["origin_normal"]), _p(out["closest"]), _p(out["normal"]),
            _p(out["cell"]), _p(out["dist"]), _p(out["src_index"]), C.byref(n)))
        return {k: v[:n.value] for k, v in out.items()}

    def correspond_f64(self, T=None, step=1, max_dist=10.0):
        """The filter's outputs in double precision (OutputPrecision = DOUBLE_PRECISION)."""
        T = np.eye(4) if T is None else np.asarray(T, np.float64)
        T = np.ascontiguousarray(T).reshape(16)
        step = max(1, int(step))
        nk = (self._np + step - 1) // step
        out = dict(origin=np.zeros((nk, 3), np.float64), closest=np.zeros((nk, 3), np.float64),
                   normal=np.zeros((nk, 3), np.float64), cell=np.zeros(nk, np.int32),
                   dist=np.zeros(nk, np.float64), src_index=np.zeros(nk, np.int64))
        n = C.c_int64(0)
        _chk(load().icpx_correspond_f64(
            self._h, _p(T), step, float(max_dist), _p(out["origin"]), _p(out["closest"]),
            _p(out["normal"]), _p(out["cell"]), _p(out["dist"]), _p(out["src_index"]),
            C.byref(n)))
        return {k: v[:n.value] for k, v in out.items()}

    def intersect_lines(self, p1, p2):
        """First intersection of the segments p1[i] -> p2[i] with the target
        (vtkImplicitPolyDataIntersection::IntersectWithLine, one call for all of them).
        Returns a dict of arrays: hit, ret, closest, normal, point, cell, kind."""
        p1 = np.ascontiguousarray(p1, np.float64).reshape(-1, 3)
        p2 = np.ascontiguousarray(p2, np.float64).reshape(-1, 3)
        if len(p1) != len(p2):
            raise ValueError("p1 and p2 must have the same number of rows")
        n = len(p1)
        out = dict(hit=np.zeros(n, np.int32), ret=np.zeros(n), closest=np.zeros((n, 3)),
                   normal=np.zeros((n, 3)), point=np.zeros((n, 3)), cell=np.zeros(n, np.int32),
                   kind=np.zeros(n, np.int32))
        _chk(load().icpx_intersect_lines(self._h, _p(p1), _p(p2), n, _p(out["hit"]),
                                         _p(out["ret"]), _p(out["closest"]), _p(out["normal"]),
                                         _p(out["point"]), _p(out["cell"]), _p(out["kind"])))
        return out

    def set_allreduce(self, fn):
        """fn(numpy view of the NSUMS sums) -> None; called once per iteration inside run()."""
        if fn is None:
            self._cb = ALLREDUCE_FN()
        else:
            def tramp(ptr, n, _user):
                try:
                    fn(np.ctypeslib.as_array(ptr, shape=(n,)))
                    return 0
                except Exception:  # surfaced as an icpx error by the library
                    return 1
            self._cb = ALLREDUCE_FN(tramp)
        _chk(load().icpx_set_allreduce(self._h, self._cb, None))

    def run_begin(self, params):
        """Start a device-resident registration (see run_enqueue / run_update / run_end)."""
        _chk(load().icpx_run_begin(self._h, C.byref(params)))

    def run_enqueue(self, device_ptr=None):
        """Enqueue one iteration's search + sums (into the device buffer at `device_ptr`, or
        an internal one).  Returns True once the run is known to have stopped."""
        done = C.c_int(0)
        _chk(load().icpx_run_enqueue(self._h, C.c_void_p(device_ptr) if device_ptr else None,
                                     C.byref(done)))
        return bool(done.value)

    def run_advance(self, device_ptr):
        """Enqueue the next pass of a loop whose position the library keeps (several ranks:
        all-reduce the sums at `device_ptr`, then run_update).  False when the run is over."""
        more = C.c_int(0)
        _chk(load().icpx_run_advance(self._h, C.c_void_p(device_ptr), C.byref(more)))
        return bool(more.value)

    def run_update(self, device_ptr=None):
        """Enqueue the solve + transform update of the iteration on the (reduced) sums."""
        _chk(load().icpx_run_update(self._h, C.c_void_p(device_ptr) if device_ptr else None))

    def run_end(self):
        """Wait for the run; returns (T 4x4, iterations, mean_distance, status)."""
        T = np.zeros(16)
        it, st, md = C.c_int(0), C.c_int(0), C.c_double(0)
        _chk(load().icpx_run_end(self._h, _p(T), C.byref(it), C.byref(md), C.byref(st)))
        return T.reshape(4, 4), it.value, md.value, st.value

    def run(self, params):
        """Whole vtkICP loop on this context. Returns (T 4x4, iterations, mean_distance, status)."""
        T = np.zeros(16)
        it, st, md = C.c_int(0), C.c_int(0), C.c_double(0)
        _chk(load().icpx_run(self._h, C.byref(params), _p(T), C.byref(it), C.byref(md),
                             C.byref(st)))
        return T.reshape(4, 4), it.value, md.value, st.value
