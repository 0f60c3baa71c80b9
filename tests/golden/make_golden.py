"""Regenerates tests/golden/*.npz from the CPU oracle.

The reference holds no golden vectors for this path and cannot be built or imported here
(VTK >= 9.3 and Eigen are absent), so these fixtures are outputs of oracle/icp_oracle.c on
seeded synthetic inputs: they pin the oracle against accidental change and give the GPU
tests a fixed answer that does not depend on the oracle being rebuilt.  PARITY UNPINNED
with respect to the reference itself.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from iterativeclosestpoint_amd import synth  # noqa: E402
from oracle import oracle as orc  # noqa: E402


def main():
    c = synth.config_c1()
    mesh = orc.Mesh(c["target_xyz"], c["target_tri"])
    sn = synth.vertex_normals(c["source_xyz"], c["target_tri"])
    T = synth.rigid(rx=3, ry=-4, rz=2, t=(0.02, -0.01, 0.03))
    rows = mesh.correspond(c["source_xyz"], T=T, step=16, max_dist=0.2, src_normals=sn,
                           min_dot=0.0, grid=False)
    out = {"T": T, "step": 16, "max_dist": 0.2}
    out.update({"rows_" + k: v for k, v in rows.items()})
    for metric in (0, 1, 2):
        M, st = orc.solve(metric, rows["origin"], rows["closest"], rows["origin_normal"],
                          rows["normal"])
        out["solve_%d" % metric] = M
        Ti, it, md, st, n = mesh.icp(
            c["source_xyz"], metric=metric, max_iter=50, max_landmarks=100,
            max_dist=10.0 if metric == 0 else 0.25 * synth.diag(c["target_xyz"]),
            src_normals=sn if metric == 2 else None)
        out["icp_%d" % metric] = Ti
        out["icp_%d_iters" % metric] = it
    np.savez_compressed(os.path.join(HERE, "c1_oracle.npz"), **out)

    # closest-point known answers on a unit cube (hand-checkable)
    v = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0], [0, 0, 1], [1, 0, 1], [1, 1, 1],
                  [0, 1, 1]], np.float32)
    t = np.array([[0, 2, 1], [0, 3, 2], [4, 5, 6], [4, 6, 7], [0, 1, 5], [0, 5, 4], [2, 3, 7],
                  [2, 7, 6], [1, 2, 6], [1, 6, 5], [0, 4, 7], [0, 7, 3]], np.int32)
    q = np.array([[2, 2, 2], [0.5, 0.5, 2], [0.5, 2, 2], [0.25, 0.5, 0.9], [-1, -1, 0.5],
                  [0.5, 0.5, 0.5], [3, 0.25, 0.25]], np.float32)
    cube = orc.Mesh(v, t).correspond(q, max_dist=100.0, grid=False)
    np.savez_compressed(os.path.join(HERE, "cube_oracle.npz"), xyz=v, tri=t, q=q,
                        **{"rows_" + k: val for k, val in cube.items()})
    print("wrote", sorted(os.listdir(HERE)))


if __name__ == "__main__":
    main()
