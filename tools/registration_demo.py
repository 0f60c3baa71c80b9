"""Headless equivalent of the reference's registration demo
(ICP/Testing/Python/RegistrationDemo.py:78-106,188-204): load a mesh (any format of
meshio.vtkPolyDataReaderFactory; a synthetic closed surface when no file is given), copy it,
move the copy by Translate(0,0,0.3) + RotateY(15), register the copy back onto the original
with vtkICP, and print what the demo prints plus the recovered matrix and how far it is from
the inverse of the perturbation.  No rendering, no animation."""
import argparse
import os
import sys
from timeit import default_timer as timer

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from iterativeclosestpoint_amd import synth, vtkICP, vtkPolyData  # noqa: E402
from iterativeclosestpoint_amd.meshio import vtkPolyDataReaderFactory  # noqa: E402


def main():
    ap = argparse.ArgumentParser(description="vtkICP registration demo, headless")
    ap.add_argument("src_fn", nargs="?", help="mesh file (.ply .vtp .stl .vtk .g .obj)")
    ap.add_argument("--metric", choices=["point", "plane", "symmetric"], default="symmetric",
                    help="the demo ends up with SetMetricToPlaneToPlane()")
    ap.add_argument("--landmarks", type=int, default=100)
    ap.add_argument("--iterations", type=int, default=100)
    ap.add_argument("--max-mean-distance", type=float, default=0.001)
    ap.add_argument("--max-distance", type=float, default=10.0)
    ap.add_argument("--device", type=int, default=0)
    a = ap.parse_args()

    if a.src_fn:
        print("Loading source:", a.src_fn)
        reader = vtkPolyDataReaderFactory.CreateReader(a.src_fn)
        reader.Update()
        target = reader.GetOutput()
    else:
        print("Loading source: synthetic closed surface (no file given)")
        xyz, tri = synth.bumpy_torus(64, 32)
        target = vtkPolyData(xyz, tri)
    print("  %d points, %d triangles" % (target.GetNumberOfPoints(), target.GetNumberOfCells()))

    # vtkTransform is pre-multiplying: Translate then RotateY gives M = Trans . RotY
    P = synth.rigid(t=(0.0, 0.0, 0.3)) @ synth.rigid(ry=15.0)
    pts = np.asarray(target.GetPoints(), np.float64)
    moved = (pts @ P[:3, :3].T + P[:3, 3]).astype(np.float32)
    source = vtkPolyData(moved, target.GetPolys())
    if a.metric == "symmetric":
        # vtkICP only runs this metric on a source that carries a "Normals" array (the demo's
        # file does); the values it uses are the filter's own (ICP/vtkICP.cxx:182)
        from iterativeclosestpoint_amd import icpx
        source.SetPointNormals(icpx.Context(a.device).point_normals(moved, target.GetPolys()))

    icpr = vtkICP(a.device)
    icpr.CheckMeanDistanceOn()
    icpr.SetMaximumMeanDistance(a.max_mean_distance)
    {"point": icpr.SetMetricToPointToPoint, "plane": icpr.SetMetricToPointToPlane,
     "symmetric": icpr.SetMetricToPlaneToPlane}[a.metric]()
    icpr.SetMaximumNumberOfIterations(a.iterations)
    icpr.SetMaximumNumberOfLandmarks(a.landmarks)
    icpr.SetMaximumDistance(a.max_distance)
    icpr.SetSource(source)
    icpr.SetTarget(target)
    start = timer()
    icpr.Update()
    elapsed = timer() - start
    print("Elapsed seconds: %f" % elapsed)
    print("Number of iterations: %d" % icpr.GetNumberOfIterations())
    print("Mean distance: %f" % icpr.GetMeanDistance())
    M = np.asarray(icpr.GetMatrix(), np.float64).reshape(4, 4)
    print("Matrix (%s):" % icpr.GetMetricModeAsString())
    for row in M:
        print("  % .9f % .9f % .9f % .9f" % tuple(row))
    back = moved.astype(np.float64) @ M[:3, :3].T + M[:3, 3]
    print("Largest distance of a registered point from where it came from: %.3e"
          % float(np.linalg.norm(back - pts, axis=1).max()))
    return 0


if __name__ == "__main__":
    sys.exit(main())
