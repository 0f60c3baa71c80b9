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
