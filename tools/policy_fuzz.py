"""Every policy, the same rows: mid-size scenes served under random settings of the library's
policy knobs (which kernel walks, lists or none, certificates, batches given up on, who weighs
the sets, seeds, query order, lean and anchored passes), each compared bit for bit with the
plainest setting (pooled lanes over the tree, nothing kept between passes).

    python tools/policy_fuzz.py [combos per scene] [seed] [big|huge|odd]

big: a million triangles and three times the points; huge: ten million triangles and sixteen
million points, the two noisy scenes only; odd: clouds of one repeated point, along a line,
nowhere near the target with and without a threshold in reach, and in a plane through it.

Prints one line per scene; exits 1 at the first difference, naming the knobs.
"""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from iterativeclosestpoint_amd import icpx, synth as S  # noqa: E402

KNOBS = dict(
    local=[0, 1, 2, 3], persistent=[1, 3, 4], bail=[0, 1, 2, 20, 200], certify_min=[1000, 100_000, 4_000_000],
    noisy_policy=[0, 1], weigh=[0, 1, 2, 3], seed=[0, 1, 2], query_curve=[0, 1], finalize=[0, 1, 2],
    lean=[0, 1, 2, 3], anchor=[0, 1], wave_blocks=[1, 2, 8], step_budget=[0, 48, 4096], defer=[0, 1, 8])
PLAIN = dict(local=0, persistent=1, bail=0, lean=0, anchor=0, step_budget=0, defer=0)


def odd_scenes():
    xyz, tri = S.bumpy_torus(400, 200)
    n = 300_000
    t = S.u01(51, 1, n)[:, None]
    yield "one point", xyz, tri, np.repeat(np.float32([[1.41, 0.02, 0.03]]), n, 0), 10.0
    yield "a line", xyz, tri, (np.float64([-2.0, 0.1, 0.05]) + t * np.float64([4.0, 0.0, 0.1])).astype(np.float32), 10.0
    pts, f, fn = S.sample_surface(xyz, tri, n, seed=53)
    yield "out of reach", xyz, tri, (pts.astype(np.float64) + 1e4).astype(np.float32), 10.0
    yield "far, no threshold", xyz, tri, (pts.astype(np.float64) + np.float64([50.0, -20.0, 10.0])).astype(np.float32), 1e9
    plane = np.stack([4 * S.u01(55, 1, n) - 2, 4 * S.u01(55, 2, n) - 2, 0 * t[:, 0]], -1)
    yield "a plane through it", xyz, tri, plane.astype(np.float32), 0.05


def scenes(big, only=None):
    if big == "odd":
        yield from odd_scenes()
        return
    nu, nv, mul = (2500, 2000, 40) if big == "huge" else (1000, 500, 3) if big else (400, 200, 1)
    for name, grade, noise, npts in (("clean", 0.0, 0.0, 400_000), ("noisy", 0.0, 2e-3, 400_000),
                                     ("graded", 0.97, 0.0, 400_000), ("graded noisy", 0.97, 4e-3, 300_000),
                                     ("sparse", 0.0, 1e-3, 3_000), ("two sheets", 0.0, 1e-4, 300_000)):
        if only and name not in only:
            continue
        xyz, tri = S.bumpy_torus(nu, nv, grade=grade)
        if name == "two sheets":
            tri = np.concatenate([tri, tri + len(xyz)]).astype(np.int32)
            xyz = np.concatenate([xyz, xyz + np.float32([1e-5, 1e-5, 1e-5])])
        pts, f, fn = S.sample_surface(xyz, tri, npts * mul, seed=41)
        cloud = pts.astype(np.float64)
        if noise:
            cloud = cloud + np.stack([S.gaussian(43, 2 * k, npts * mul) for k in range(3)], -1) * noise
        M = S.rigid(ry=3.0, t=(0.01, 0.01, -0.01))
        yield name, xyz, tri, S.apply(M, cloud), 10.0
    if not only or "strays" in only:
        # one point in fifty anywhere in the target's box, no threshold
        xyz, tri = S.bumpy_torus(nu, nv)
        n = 300_000 * mul
        pts, f, fn = S.sample_surface(xyz, tri, n, seed=45)
        lo, hi = xyz.min(0).astype(np.float64), xyz.max(0).astype(np.float64)
        cloud = pts.astype(np.float64)
        cloud[::50] = np.stack([S.u01(47, k, len(cloud[::50])) for k in range(3)], -1) * (hi - lo) + lo
        yield "strays", xyz, tri, S.apply(S.rigid(ry=3.0, t=(0.01, 0.01, -0.01)), cloud), 1e9
    if only and "partial" not in only:
        return
    # partial overlap, a fifth of the cloud outliers, a threshold of 2 % of the diagonal
    c = S.config_c5(nu=nu, nv=nv, npts=300_000 * mul)
    yield "partial", c["target_xyz"], c["target_tri"], c["source_xyz"], c["max_dist"]


def serve(xyz, tri, src, max_dist, knobs, moves, shard=None, metric=1, stride=1, rows=True):
    ctx = icpx.Context(0)
    for k, v in knobs.items():
        getattr(ctx, "set_" + k)(v)
    ctx.set_target(xyz, tri)
    # (the symmetric metric wants source normals: any unit vectors will do for a comparison)
    ctx.set_source(src, np.tile(np.float32([[0.6, 0.0, 0.8]]), (len(src), 1)) if metric == 2 else None)
    if shard:
        ctx.set_shard_order(shard[2])
        ctx.set_shard(shard[0], shard[1])
    out = []
    T = np.eye(4)
    for i, m in enumerate(moves if rows else ()):
        T = S.rigid(rz=20 * m, t=(m, -m, 0.5 * m)) @ T
        # (the threshold changes from pass to pass: certificates of "nothing within reach" hold
        # for the threshold they were made for, or a smaller one)
        got = ctx.correspond(T, step=1, max_dist=max_dist * (1.0, 1.0, 0.5, 2.0, 1.0, 0.25, 3.0, 1.0)[i % 8])
        out.append((got["cell"].copy(), got["dist"].copy(), got["closest"].copy(), got["src_index"].copy()))
    p = icpx.default_params()
    p.metric, p.max_iterations, p.max_landmarks, p.max_distance = metric, 30, len(src) // stride, max_dist
    Tr, it, mean, st = ctx.run(p)
    ctx.close()
    return out, (Tr.copy(), it, mean, st)


def fuzz(ncombo, seed, big=False, only=None):
    """None if every setting agrees, else a description of the first difference."""
    rng = np.random.default_rng(seed)
    moves = (0, 0, 1e-7, 1e-5, 1e-5, 1e-3, 1e-6, 3e-2, 1e-6, 1e-6, 1e-4, 0)
    for name, xyz, tri, src, max_dist in scenes(big, only):
        t0 = tick = time.perf_counter()
        rows0, run0 = serve(xyz, tri, src, max_dist, PLAIN, moves)
        plain_runs = {(1, 1): run0}
        for c in range(ncombo):
            knobs = {k: int(rng.choice(v)) for k, v in KNOBS.items() if rng.random() < 0.6}
            # (one setting in four serves one rank's share of a job of 2, 3 or 8: its rows are
            # the whole cloud's rows of its points)
            shard = None
            if rng.random() < 0.25:
                world = int(rng.choice([2, 3, 8]))
                shard = (int(rng.integers(world)), world, int(rng.integers(2)))
                knobs = dict(knobs, shard=shard)
            # (the registration by each metric, of every point or every third)
            metric, stride = int(rng.integers(3)), int(rng.choice([1, 3]))
            if not shard and (metric, stride) not in plain_runs:
                plain_runs[(metric, stride)] = serve(xyz, tri, src, max_dist, PLAIN, moves, None, metric, stride, False)[1]
            if not shard:
                run0 = plain_runs[(metric, stride)]
                knobs = dict(knobs, metric=metric, stride=stride)
            rows, run = serve(xyz, tri, src, max_dist,
                              {k: v for k, v in knobs.items() if k not in ("shard", "metric", "stride")}, moves, shard,
                              metric, stride)
            if time.perf_counter() - tick > 30.0:   # (a sign of life for whoever watches a long run)
                tick = time.perf_counter()
                print("  ... %s: %d of %d settings" % (name, c + 1, ncombo), file=sys.stderr, flush=True)
            for i, (a, b) in enumerate(zip(rows0, rows)):
                if shard:
                    at = np.searchsorted(a[3], b[3])
                    if len(b[3]) and (at.max() >= len(a[3]) or not np.array_equal(a[3][at], b[3])):
                        return "scene %r pass %d: the shard accepts points the whole cloud does not, under %r" % (name, i, knobs)
                    a = tuple(v[at] for v in a)
                for what, x, y in zip(("cell", "dist", "closest", "src_index"), a, b):
                    if x.shape != y.shape:
                        return "scene %r pass %d: %d rows against %d under %r" % (name, i, len(y), len(x), knobs)
                    if not np.array_equal(x, y):
                        bad = np.flatnonzero((x != y).reshape(len(x), -1).any(1))
                        return ("scene %r pass %d %s at %d queries (first %s) under %r" %
                                (name, i, what, len(bad), bad[:5], knobs))
            # (the policy changes nothing: the same transform to the last bit.  The sums' order
            # follows the query order and the finalize build: with those set the loops agree to
            # rounding, and to what rounding becomes where a stray's nearest triangle flips, 1e-8)
            # (the odd clouds make singular systems, whose answers follow the sums' last digits)
            same_sums = "finalize" not in knobs and "query_curve" not in knobs
            if shard or (big == "odd" and not same_sums):
                ok = True
            elif same_sums:
                ok = np.array_equal(run0[0], run[0]) and run0[1:] == run[1:]
            else:
                ok = (np.abs(run0[0] - run[0]).max() < 1e-6 and run0[1] == run[1] and run0[3] == run[3]
                      and abs(run0[2] - run[2]) <= 1e-6 * abs(run0[2]) + 1e-9)
            if not ok:
                return "scene %r registration under %r: %r against %r" % (name, knobs, run, run0)
        print("%-14s %d triangles %d points: %d settings agree with the plain one on %d passes and the registration (%.1f s)"
              % (name, len(tri), len(src), ncombo, len(moves), time.perf_counter() - t0), flush=True)
    return None


if __name__ == "__main__":
    bad = fuzz(int(sys.argv[1]) if len(sys.argv) > 1 else 24, int(sys.argv[2]) if len(sys.argv) > 2 else 7,
               sys.argv[3] if len(sys.argv) > 3 else False,
               only=("noisy", "graded noisy") if len(sys.argv) > 3 and sys.argv[3] == "huge" else None)
    if bad:
        print("DIFFERENT:", bad)
        sys.exit(1)
