"""Development aid: registrations of shapes the benchmark's workloads do not cover (a far start,
a sparse and a dense cloud, an elongated scene, heavy noise), pass by pass, to find where the
host's policy or a kernel falls off a cliff."""
import os
import sys
import time
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402
from iterativeclosestpoint_amd import icpx, synth  # noqa: E402


def refine(xyz, tri, sel, m):
    """Triangles `sel` cut into m x m coplanar ones each (the surface stays what it was)."""
    P = xyz.astype(np.float64)
    newv, newt = [], []
    base = len(xyz)
    idx = {}
    k = 0
    for i in range(m + 1):
        for j in range(m + 1 - i):
            idx[(i, j)] = k
            k += 1
    per = k
    ij = np.array(sorted(idx, key=idx.get), np.float64) / m
    cells = []
    for i in range(m):
        for j in range(m - i):
            cells.append((idx[(i, j)], idx[(i + 1, j)], idx[(i, j + 1)]))
            if i + j < m - 1:
                cells.append((idx[(i + 1, j)], idx[(i + 1, j + 1)], idx[(i, j + 1)]))
    cells = np.array(cells, np.int64)
    for n, t in enumerate(sel):
        a, b, c = P[tri[t, 0]], P[tri[t, 1]], P[tri[t, 2]]
        newv.append(a + ij[:, :1] * (b - a) + ij[:, 1:] * (c - a))
        newt.append(cells + base + n * per)
    keep = np.ones(len(tri), bool)
    keep[sel] = False
    return (np.concatenate([xyz, np.concatenate(newv).astype(np.float32)]),
            np.concatenate([tri[keep], np.concatenate(newt).astype(np.int32)]))


def case(name):
    S = synth
    if name == "far":      # C3's pair from 15 % of the diagonal and 25 degrees off
        c = S.config_c3()
        xyz, tri = c["target_xyz"], c["target_tri"]
        pts, f, fn = S.sample_surface(xyz, tri, 1_000_000)
        M = S.rigid(ry=25.0, t=(0.15 * S.diag(xyz) / np.sqrt(3),) * 3)
        c.update(source_xyz=S.apply(M, pts), source_normals=None, metric=1, truth=M)
    elif name == "sparse":  # 100 000 points on C3's million triangles
        c = S.config_c3(npts=100_000)
    elif name.startswith("sparse"):  # sparse<points in thousands>k[x<triangles in millions>m]
        a, _, b = name[6:].partition("x")
        nu, nv = (2500, 2000) if b == "10m" else (1000, 500)
        c = S.config_c3(nu=nu, nv=nv, npts=1000 * int(a.rstrip("k")))
    elif name == "dense":   # 4 million points on 20 000 triangles
        c = S.config_c3(nu=100, nv=100, npts=4_000_000)
    elif name == "long":    # C3's scene stretched twentyfold along x
        c = S.config_c3()
        sc = np.array([20.0, 1.0, 1.0])
        xyz = (c["target_xyz"].astype(np.float64) * sc).astype(np.float32)
        pts, f, fn = S.sample_surface(xyz, c["target_tri"], 1_000_000)
        M = S.rigid(ry=1.0, t=(0.01 * S.diag(xyz) / np.sqrt(3),) * 3)
        c.update(target_xyz=xyz, source_xyz=S.apply(M, pts), source_normals=None, metric=1, truth=M)
    elif name == "noisy":   # C3's pair with noise of five edge lengths
        c = S.config_c3()
        xyz, tri = c["target_xyz"], c["target_tri"]
        pts, f, fn = S.sample_surface(xyz, tri, 1_000_000)
        noise = np.stack([S.gaussian(S.SEED, 32 + 2 * k, len(pts)) for k in range(3)], -1) * 0.03
        c.update(source_xyz=S.apply(c["truth"], pts.astype(np.float64) + noise), source_normals=None, metric=1)
    elif name == "c4clean":  # C4's pair without the noise: every foot point inside a triangle
        c = S.config_c4()
        pts, f, fn = S.sample_surface(c["target_xyz"], c["target_tri"], 16_000_000)
        c.update(source_xyz=S.apply(c["truth"], pts))
    elif name.startswith("offset"):  # C3's scene moved away from the origin by offset<distance>
        c = S.config_c3()
        sh = np.array([float(name[6:]), 0.0, 0.0])
        xyz = (c["target_xyz"].astype(np.float64) + sh).astype(np.float32)
        pts, f, fn = S.sample_surface(xyz, c["target_tri"], 1_000_000)
        ctr = 0.5 * (xyz.max(0).astype(np.float64) + xyz.min(0))
        M0 = S.rigid(ry=5.0, t=(0.01 * S.diag(xyz) / np.sqrt(3),) * 3)
        M = np.eye(4)
        M[:3, :3] = M0[:3, :3]
        M[:3, 3] = M0[:3, 3] + ctr - M0[:3, :3] @ ctr   # (the same motion about the scene's centre)
        c.update(target_xyz=xyz, source_xyz=S.apply(M, pts), source_normals=None, metric=1, truth=M)
    elif name.startswith("patch"):  # C3's torus with patch<n> of its triangles cut into 64 x 64 each
        c = S.config_c3()
        xyz, tri = c["target_xyz"], c["target_tri"]
        n = int(name[5:] or 3)
        cen = xyz[tri].mean(1)
        sel = np.argsort(np.linalg.norm(cen - cen[12345], axis=1))[:n]
        xyz2, tri2 = refine(xyz, tri, sel, 64)
        c.update(target_xyz=xyz2, target_tri=tri2, source_normals=None, metric=1)
    elif name == "flat":    # a flat square of a million triangles (a singular system for the solvers)
        n = 708
        g = np.linspace(0.0, 1.0, n, dtype=np.float64)
        X, Y = np.meshgrid(g, g, indexing="ij")
        xyz = np.stack([X, Y, 0.0 * X], -1).reshape(-1, 3).astype(np.float32)
        i, j = np.meshgrid(np.arange(n - 1), np.arange(n - 1), indexing="ij")
        p00 = (i * n + j).ravel()
        tri = np.concatenate([np.stack([p00, p00 + n, p00 + n + 1], -1), np.stack([p00, p00 + n + 1, p00 + 1], -1)]).astype(np.int32)
        pts, f, fn = S.sample_surface(xyz, tri, 1_000_000)
        M = S.rigid(rx=2.0, t=(0.002, -0.001, 0.01))
        c = dict(target_xyz=xyz, target_tri=tri, source_xyz=S.apply(M, pts), source_normals=None, metric=1,
                 max_dist=10.0, truth=M)
    elif name == "apart":   # two of C3's tori a hundred units apart, the cloud on one of them
        c = S.config_c3(nu=700, nv=350)
        xyz, tri = c["target_xyz"], c["target_tri"]
        xyz2 = np.concatenate([xyz, xyz + np.float32([100.0, 0, 0])])
        tri2 = np.concatenate([tri, tri + len(xyz)]).astype(np.int32)
        c.update(target_xyz=xyz2, target_tri=tri2, source_normals=None, metric=1)
    elif name == "sliver":  # every triangle a sliver: 4000 x 125 grid lines, sides 32 to 1
        c = S.config_c3(nu=4000, nv=125)
        c.update(source_normals=None, metric=1)
    elif name == "fewpts":  # a thousand points on ten million triangles
        c = S.config_c4(npts=1000)
    elif name == "twosheet":  # two copies of C3's torus 1e-5 apart: two contenders for every point
        c = S.config_c3(nu=700, nv=350)
        xyz, tri = c["target_xyz"], c["target_tri"]
        xyz2 = np.concatenate([xyz, xyz + np.float32([1e-5, 1e-5, 1e-5])])
        tri2 = np.concatenate([tri, tri + len(xyz)]).astype(np.int32)
        c.update(target_xyz=xyz2, target_tri=tri2, source_normals=None, metric=1)
    elif name.startswith("away"):  # C3's cloud away<distance> units off its place (the target is 3 across), no threshold
        d = float(name[4:])
        c = S.config_c3()
        M = S.rigid(ry=5.0, t=(0.8 * d, -0.5 * d, 0.33 * d))
        pts = S.apply(np.linalg.inv(c["truth"]), c["source_xyz"])
        c.update(source_xyz=S.apply(M, pts), source_normals=None, metric=1, max_dist=1e9, truth=M)
    elif name.startswith("strays"):  # C3's cloud with strays<n> of its points anywhere in the target's box, no threshold
        n = int(name[6:])
        c = S.config_c3()
        lo, hi = c["target_xyz"].min(0).astype(np.float64), c["target_xyz"].max(0).astype(np.float64)
        out = np.stack([S.u01(61, k, n) for k in range(3)], -1) * (hi - lo) + lo
        src = c["source_xyz"].copy()
        src[:: len(src) // n][:n] = S.apply(c["truth"], out)
        c.update(source_xyz=src, source_normals=None, metric=1, max_dist=1e9)
    elif name == "cube":    # a million points on the twelve triangles of a cube
        v = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [0, 0, 1], [1, 0, 1], [1, 1, 1], [0, 1, 1]], np.float32)
        t = np.array([[0, 2, 1], [0, 3, 2], [4, 5, 6], [4, 6, 7], [0, 1, 5], [0, 5, 4], [1, 2, 6], [1, 6, 5],
                      [2, 3, 7], [2, 7, 6], [3, 0, 4], [3, 4, 7]], np.int32)
        pts, f, fn = S.sample_surface(v, t, 1_000_000)
        M = S.rigid(ry=3.0, t=(0.01, 0.02, 0.015))
        c = dict(target_xyz=v, target_tri=t, source_xyz=S.apply(M, pts), source_normals=None, metric=1,
                 max_dist=10.0, truth=M)
    elif name == "c3":
        c = S.config_c3()
    elif name == "c6":      # the graded torus
        c = S.config_c6()
    elif name == "c6noisy":  # ... with noise of two mean edge lengths (many small ones)
        c = S.config_c6()
        pts, f, fn = S.sample_surface(c["target_xyz"], c["target_tri"], 1_000_000)
        noise = np.stack([S.gaussian(S.SEED, 32 + 2 * k, len(pts)) for k in range(3)], -1) * 0.012
        c.update(source_xyz=S.apply(c["truth"], pts.astype(np.float64) + noise))
    elif name == "c6big":    # ten million graded triangles, sixteen million noisy points
        c = S.config_c6(nu=2500, nv=2000, npts=16_000_000)
        pts, f, fn = S.sample_surface(c["target_xyz"], c["target_tri"], 16_000_000)
        noise = np.stack([S.gaussian(S.SEED, 32 + 2 * k, len(pts)) for k in range(3)], -1) * (1e-3 * S.diag(c["target_xyz"]))
        c.update(source_xyz=S.apply(c["truth"], pts.astype(np.float64) + noise))
    else:
        raise SystemExit("unknown case " + name)
    return c


for spec in (sys.argv[1:] or ["c3", "far", "sparse", "dense", "long", "noisy"]):
    # name[:knob=value,...] with the knobs of icpx.Context (persistent, bail, local, ...)
    name, _, knobs = spec.partition(":")
    c = case(name)
    ctx = icpx.Context(0)
    for kv in filter(None, knobs.split(",")):
        k, v = kv.split("=")
        getattr(ctx, "set_" + k)(int(v))
    ctx.set_target(c["target_xyz"], c["target_tri"])
    ctx.set_source(c["source_xyz"], c["source_normals"])
    p = icpx.default_params()
    p.metric, p.max_iterations, p.max_landmarks, p.max_distance = c["metric"], 50, len(c["source_xyz"]), c["max_dist"]
    ctx.run(p)
    t = []
    for _ in range(3):
        t0 = time.perf_counter()
        T, it, mean, st = ctx.run(p)
        t.append(1e3 * (time.perf_counter() - t0))
    ctx.set_timing(2)
    ctx.run(p)
    ms, kind, hinted = ctx.get_pass_timings()
    ctx.set_timing(0)
    stt = ctx.debug_stats()
    print("%-20s %d tri %d pts: registration %.2f ms (%d iterations, status %d, mean %.2e, off truth %.1e)" %
          (spec, len(c["target_tri"]), len(c["source_xyz"]), np.median(t), it, st, mean,
           np.abs(T @ c["truth"] - np.eye(4)).max()))
    print("        passes ms", [round(float(v), 3) for v in ms[:10]], "... settled %.3f" % float(np.median(ms[25:])),
          "kinds", [int(k) for k in kind[:10]])
    print("        ", {k: v for k, v in stt.items() if v})
    ctx.close()
