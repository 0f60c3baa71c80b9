"""icpx_accumulate_static (the entry point that allocates nothing: its scratch is a static
device array) against icpx_accumulate, through the C ABI: bit-identical sums on calls in a
row over the shared scratch, the second with a smaller input; limits and preconditions."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _static(icpx, ctx, metric, T, step, max_dist, min_dot, use_sn, out):
    ctx.accumulate_static(metric, T, step, max_dist, min_dot, use_sn, out.data_ptr())


def test_static_scratch_accumulate_matches_accumulate():
    from iterativeclosestpoint_amd import icpx, synth
    xyz, tri = synth.bumpy_torus(96, 48)
    pts, f, fn = synth.sample_surface(xyz, tri, 30_000, seed=77)
    M = synth.rigid(ry=4.0, t=(0.02, 0.0, -0.01))
    src = synth.apply(M, pts)
    sn = (fn.astype(np.float64) @ M[:3, :3].T).astype(np.float32)
    T = synth.rigid(rx=1.0, t=(0.0, 0.005, 0.0))
    ctx = icpx.Context(0)
    ctx.set_target(xyz, tri)
    ctx.set_source(src, sn)
    out = torch.zeros(icpx.NSUMS, dtype=torch.float64, device="cuda:0")
    torch.cuda.synchronize()
    with pytest.raises(icpx.IcpxError, match="not built"):
        _static(icpx, ctx, 1, T, 1, 0.2, 0.0, 0, out)          # no tree yet: it never builds one
    # the existing entry point first: it builds the trees (allocating) and gives the answers
    cases = [(metric, step) for metric in (0, 1, 2) for step in (1, 4, 11, 300)]
    want = {}
    for metric, step in cases:
        want[metric, step] = ctx.accumulate(metric, T, step=step, max_dist=0.2, min_dot=0.3)
    # then calls in a row over the shared scratch: large, smaller, smaller still, and back up
    for metric in (0, 1, 2):
        for step in (1, 4, 11, 300, 4, 1):                      # 30000, 7500, 2728, 100 queries
            out.fill_(-1.0)
            torch.cuda.synchronize()
            _static(icpx, ctx, metric, T, step, 0.2, 0.3, 1 if metric == 2 else 0, out)
            ctx.synchronize()
            got = out.cpu().numpy()
            assert got[0] == want[metric, step][0] > 0
            assert np.array_equal(got, want[metric, step]), (metric, step)
    # the corrected normals mode goes through as well
    w2 = ctx.accumulate(2, T, step=4, max_dist=0.2, min_dot=0.3, use_source_normals=2)
    _static(icpx, ctx, 2, T, 4, 0.2, 0.3, 2, out)
    ctx.synchronize()
    assert np.array_equal(out.cpu().numpy(), w2)
    # a second context on the same GPU shares the array; used one after the other, both work
    other = icpx.Context(0)
    other.set_target(xyz, tri)
    other.set_source(src[::2].copy())
    w_other = other.accumulate(1, np.eye(4), step=1, max_dist=0.5)
    _static(icpx, other, 1, np.eye(4), 1, 0.5, 0.0, 0, out)
    other.synchronize()
    assert np.array_equal(out.cpu().numpy(), w_other)
    _static(icpx, ctx, 1, T, 11, 0.2, 0.3, 0, out)
    ctx.synchronize()
    assert np.array_equal(out.cpu().numpy(), want[1, 11])
    # limits: more kept points than the scratch holds, bad arguments
    big = icpx.Context(0)
    big.set_target(xyz, tri)
    big.set_source(np.repeat(src, 2, axis=0))
    with pytest.raises(icpx.IcpxError, match="at most"):
        _static(icpx, big, 1, T, 1, 0.2, 0.0, 0, out)
    with pytest.raises(icpx.IcpxError):
        _static(icpx, ctx, 7, T, 1, 0.2, 0.0, 0, out)
    lib = icpx.load()
    assert lib.icpx_accumulate_static(ctx._h, 1, np.eye(4).ctypes.data_as(C.c_void_p), 1, 0.2,
                                      0.0, 0, None) == icpx.E_INVALID
