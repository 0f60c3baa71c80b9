"""Golden vectors of the large configuration (10M triangles, 16M noisy points, point-to-plane)
from the CPU oracle: the accumulated 4x4 after one and after two iterations of the loop of
vtkICP::InternalUpdate (ICP/vtkICP.cxx:189-246) and every 3200th correspondence row of both
passes.  Tens of minutes per iteration on eight cores; run once in the build container:

    python tools/make_c4_golden.py          # writes tests/golden/c4_two_iterations.npz

The inputs are regenerated from the seed by synth.config_c4(), so only the outputs are stored.
The cloud is searched in blocks whose rows are parked under /tmp, so a run that is cut short
resumes where it stopped.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from iterativeclosestpoint_amd import synth  # noqa: E402
from oracle import oracle  # noqa: E402

SAMPLE_STRIDE = 3200
BLOCK = 400_000
PARK = "/tmp/c4_golden_blocks"


def main():
    out = os.path.join(ROOT, "tests", "golden", "c4_two_iterations.npz")
    os.makedirs(PARK, exist_ok=True)
    c = synth.config_c4()
    src = c["source_xyz"]
    n = len(src)
    t0 = time.perf_counter()
    mesh = oracle.Mesh(c["target_xyz"], c["target_tri"])
    print("oracle mesh: %.1f s" % (time.perf_counter() - t0), flush=True)
    T = np.eye(4)
    keep = {}
    for it in range(2):
        tpath = os.path.join(PARK, "T%d.npy" % (it + 1))
        parts = []
        for a in range(0, n, BLOCK):
            path = os.path.join(PARK, "it%d_%08d.npz" % (it, a))
            if not os.path.exists(path):
                t0 = time.perf_counter()
                rows = mesh.correspond(src[a:a + BLOCK], T=T, step=1, max_dist=c["max_dist"])
                rows["src_index"] = rows["src_index"] + a
                np.savez(path + ".tmp.npz", **rows)
                os.replace(path + ".tmp.npz", path)
                print("iteration %d block %d: %.1f s" % (it + 1, a // BLOCK,
                                                         time.perf_counter() - t0), flush=True)
            parts.append(path)
        rows = {}
        loaded = [np.load(p) for p in parts]
        for k in ("origin", "closest", "normal", "src_index", "cell", "dist", "kind"):
            rows[k] = np.concatenate([ld[k] for ld in loaded])
        del loaded
        M, st = oracle.solve(1, rows["origin"], rows["closest"], None, rows["normal"])
        assert st == 0
        pick = rows["src_index"] % SAMPLE_STRIDE == 0
        keep["pose%d" % it] = T.copy()
        keep["n%d" % it] = np.int64(len(rows["cell"]))
        keep["mean%d" % it] = np.float64(np.linalg.norm(
            rows["origin"].astype(np.float64) - rows["closest"], axis=1).mean())
        for k in ("src_index", "cell", "dist", "closest", "normal", "kind"):
            keep["%s%d" % (k, it)] = rows[k][pick]
        T = M @ T  # T <- M * T (vtkICP.cxx:240)
        keep["T%d" % (it + 1)] = T.copy()
        np.save(tpath, T)
        print("iteration %d done: %d rows" % (it + 1, len(rows["cell"])), flush=True)
        del rows
    np.savez_compressed(out, sample_stride=np.int64(SAMPLE_STRIDE), **keep)
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    main()
