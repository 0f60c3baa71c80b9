"""The reference's class surface over libicpx.

`vtkICP` here has the methods, defaults and error behaviour of the reference's
vtkICP (ICP/vtkICP.h:20-155, ICP/vtkICP.cxx:27-247) so that scripts written
against the reference's Python wrapping (Wrapping/Python/icp.py) read the same.
VTK itself is not a dependency: `vtkPolyData` below is a small holder for the
arrays vtkICP actually touches (float points, triangle connectivity, optional
"Normals" point data and cell normals).  All per-point work happens in the HIP
library; `icp_loop` is the iterate loop shared by the single-GPU and the
sharded driver.
"""
import sys

import numpy as np

from . import icpx


class vtkPolyData:
    """Points (float32 N x 3), triangles (int32 M x 3), optional normals."""

    def __init__(self, points=None, polys=None):
        self._mtime = 0  # bumped by every setter and by Modified(), like vtkObject's MTime
        self._points = None
        self._polys = np.zeros((0, 3), np.int32)
        self._point_normals = None
        self._cell_normals = None
        self._point_dtype = np.float32
        if points is not None:
            self.SetPoints(points)
        if polys is not None:
            self.SetPolys(polys)

    def Modified(self):
        """Mark the data as changed (call after editing an array in place): consumers
        re-read it on their next Update(), as VTK's pipeline does on a newer MTime."""
        self._mtime += 1

    def GetMTime(self):
        return self._mtime

    def SetPoints(self, pts):
        self._mtime += 1
        pts = np.asarray(pts)
        self._point_dtype = pts.dtype  # vtkICP refuses anything but float (vtkICP.cxx:145-150)
        self._points = np.ascontiguousarray(pts).reshape(-1, 3)

    def GetPoints(self):
        return self._points

    def GetNumberOfPoints(self):
        return 0 if self._points is None else len(self._points)

    def SetPolys(self, tri):
        self._mtime += 1
        self._polys = np.ascontiguousarray(tri, np.int32).reshape(-1, 3)

    def GetPolys(self):
        return self._polys

    def GetNumberOfCells(self):
        return len(self._polys)

    def SetPointNormals(self, n):
        self._mtime += 1
        # float, or double where the producer ran with double output precision
        dt = np.float64 if n is not None and np.asarray(n).dtype == np.float64 else np.float32
        self._point_normals = None if n is None else np.ascontiguousarray(n, dt).reshape(-1, 3)

    def GetPointNormals(self):
        return self._point_normals

    def SetCellNormals(self, n):
        self._mtime += 1
        self._cell_normals = None if n is None else np.ascontiguousarray(
            n, np.float32).reshape(-1, 3)

    def GetCellNormals(self):
        return self._cell_normals

    def DeepCopy(self, other):
        self._mtime += 1
        self._points = None if other._points is None else other._points.copy()
        self._point_dtype = other._point_dtype
        self._polys = other._polys.copy()
        self._point_normals = None if other._point_normals is None else other._point_normals.copy()
        self._cell_normals = None if other._cell_normals is None else other._cell_normals.copy()


def icp_loop(accumulate, allreduce, origin, metric, max_iterations, check_mean_distance,
             max_mean_distance):
    """The iterate loop of vtkICP::InternalUpdate (ICP/vtkICP.cxx:189-246).

    accumulate(T) -> the NSUMS partial sums of this rank for transform T;
    allreduce(sums) -> sums added over ranks (None for a single rank).
    Returns (T, iterations, mean_distance, status)."""
    T = np.eye(4)
    iters, status, mean = 0, icpx.RUN_MAX_ITERATIONS, 0.0
    for _ in range(int(max_iterations)):
        sums = accumulate(T)
        if allreduce is not None:
            sums = allreduce(sums)
        sums = np.asarray(sums, np.float64)
        M, n, mean, rc = icpx.solve(metric, sums, origin)
        if check_mean_distance and n > 0 and mean <= max_mean_distance:
            status = icpx.RUN_CONVERGED
            break
        if rc == icpx.E_FEW:
            status = icpx.RUN_TOO_FEW
            break
        Tn = M @ T  # vtkTransform PostMultiply + Concatenate (vtkICP.cxx:169,240)
        # (a transform the library does not take, icpx.h: the loop ends with the last good one)
        if not (np.all(np.abs(Tn[:3, :3]) <= 1e6) and np.all(np.abs(Tn[:3, 3]) <= 1e18)):
            status = icpx.RUN_DIVERGED
            break
        iters += 1
        T = Tn
    return T, iters, mean, status


class vtkICP:
    """vtkLinearTransform-like registration object (ICP/vtkICP.h)."""

    MetricPointToPoint = 0
    MetricPointToPlane = 1
    MetricPlaneToPlane = 2

    def __init__(self, device=0):
        # defaults: ICP/vtkICP.cxx:27-40; landmarks 200 as documented in vtkICP.h:52-59
        self.NumberOfIterations = 0
        self.MaximumNumberOfIterations = 50
        self.CheckMeanDistance = False
        self.MeanDistance = 0.0
        self.MaximumMeanDistance = 0.01
        self.MaximumDistance = 10.0
        self.MaximumNumberOfLandmarks = 200
        self.Metric = vtkICP.MetricPointToPlane
        self.MinNormalDot = 0.0
        self.RotateSourceNormals = False  # opt-in corrected mode; the reference never rotates them
        self.Source = None
        self.Target = None
        self.Matrix = np.eye(4)
        self.Status = icpx.RUN_MAX_ITERATIONS
        self._device = device
        self._ctx = None
        self._target_key = None
        self._source_key = None
        self._modified = True

    @staticmethod
    def New():
        return vtkICP()

    # --- setters / getters (vtkSetMacro / vtkGetMacro of vtkICP.h) ---------------------
    def SetSource(self, pd):
        self.Source = pd
        self._modified = True

    def SetTarget(self, pd):
        self.Target = pd
        self._modified = True

    def GetSource(self):
        return self.Source

    def GetTarget(self):
        return self.Target

    def SetMaximumNumberOfIterations(self, n):
        self.MaximumNumberOfIterations = int(n)
        self._modified = True

    def GetMaximumNumberOfIterations(self):
        return self.MaximumNumberOfIterations

    def SetMaximumNumberOfLandmarks(self, n):
        self.MaximumNumberOfLandmarks = int(n)
        self._modified = True

    def GetMaximumNumberOfLandmarks(self):
        return self.MaximumNumberOfLandmarks

    def SetMetric(self, m):
        self.Metric = min(max(int(m), 0), 2)  # vtkSetClampMacro
        self._modified = True

    def GetMetric(self):
        return self.Metric

    def SetMetricToPointToPoint(self):
        self.SetMetric(vtkICP.MetricPointToPoint)

    def SetMetricToPointToPlane(self):
        self.SetMetric(vtkICP.MetricPointToPlane)

    def SetMetricToPlaneToPlane(self):
        self.SetMetric(vtkICP.MetricPlaneToPlane)

    def GetMetricModeAsString(self):
        return ("Point-to-point", "Point-to-plane", "Plane-to-plane")[self.Metric]

    def GetNumberOfIterations(self):
        return self.NumberOfIterations

    def SetCheckMeanDistance(self, v):
        self.CheckMeanDistance = bool(v)
        self._modified = True

    def GetCheckMeanDistance(self):
        return self.CheckMeanDistance

    def CheckMeanDistanceOn(self):
        self.SetCheckMeanDistance(True)

    def CheckMeanDistanceOff(self):
        self.SetCheckMeanDistance(False)

    def SetMaximumMeanDistance(self, v):
        self.MaximumMeanDistance = float(v)
        self._modified = True

    def GetMaximumMeanDistance(self):
        return self.MaximumMeanDistance

    def SetMaximumDistance(self, v):
        self.MaximumDistance = float(v)
        self._modified = True

    def GetMaximumDistance(self):
        return self.MaximumDistance

    def GetMeanDistance(self):
        return self.MeanDistance

    def SetRotateSourceNormals(self, v):
        """Not in the reference: with the plane-to-plane metric, rotate the source normals
        with the running transform (the reference uses them as given in every iteration,
        ICP/vtkPolyDataCorrespondenceFilter.cxx:284,299).  Off by default."""
        self.RotateSourceNormals = bool(v)
        self._modified = True

    def GetRotateSourceNormals(self):
        return self.RotateSourceNormals

    def Inverse(self):
        self.Source, self.Target = self.Target, self.Source
        self._modified = True

    def MakeTransform(self):
        return vtkICP(self._device)

    def DeepCopy(self, t):
        self.SetSource(t.GetSource())
        self.SetTarget(t.GetTarget())
        self.SetMaximumNumberOfIterations(t.GetMaximumNumberOfIterations())
        self.SetCheckMeanDistance(t.GetCheckMeanDistance())
        self.SetMaximumMeanDistance(t.GetMaximumMeanDistance())
        self.SetMaximumNumberOfLandmarks(t.GetMaximumNumberOfLandmarks())
        self.SetMaximumDistance(t.GetMaximumDistance())
        self.SetMetric(t.GetMetric())
        self.Matrix = t.Matrix.copy()

    def Modified(self):
        """An explicit Modified() also drops the device copies of both meshes: the next
        Update() reads the current arrays again, as the reference does on every Update()
        (arrays edited in place are picked up without a Modified() on the poly data)."""
        self._modified = True
        self._target_key = None
        self._source_key = None

    def PrintSelf(self, os=sys.stdout, indent=""):
        for k in ("Source", "Target"):
            v = getattr(self, k)
            os.write("%s%s: %s\n" % (indent, k, "(none)" if v is None else hex(id(v))))
        for k in ("MaximumNumberOfIterations", "CheckMeanDistance", "MaximumMeanDistance",
                  "MaximumDistance", "NumberOfIterations", "MeanDistance"):
            os.write("%s%s: %s\n" % (indent, k, getattr(self, k)))
        os.write("%sMetric: %s\n" % (indent, self.GetMetricModeAsString()))

    # --- vtkAbstractTransform surface ------------------------------------------------
    def Update(self):
        if self._modified:
            self.InternalUpdate()
            self._modified = False

    def GetMatrix(self):
        self.Update()
        return self.Matrix

    def TransformPoint(self, p):
        M = self.GetMatrix()
        return M[:3, :3] @ np.asarray(p, np.float64) + M[:3, 3]

    def _error(self, msg):
        sys.stderr.write("ERROR: vtkICP: %s\n" % msg)  # vtkErrorMacro: report, leave Matrix alone

    def _source_normals(self):
        src = self.Source
        if src.GetNumberOfCells() > 0:
            # the filter's own vtkPolyDataNormals (ICP/vtkPolyDataCorrespondenceFilter.cxx:457-464)
            return self._ctx.point_normals(src.GetPoints(), src.GetPolys())
        return src.GetPointNormals()

    def _upload(self):
        if self._ctx is None:
            self._ctx = icpx.Context(self._device)
        tgt, src = self.Target, self.Source
        # Device copies are keyed on (object, its modification count): the object itself is
        # kept in the key, so its identity cannot be reused by another one while it is there.
        if self._target_key is None or self._target_key[0] is not tgt or \
                self._target_key[1] != tgt.GetMTime():
            self._ctx.set_target(tgt.GetPoints(), tgt.GetPolys(), tgt.GetCellNormals())
            self._target_key = (tgt, tgt.GetMTime())
        sym = self.Metric == vtkICP.MetricPlaneToPlane
        if self._source_key is None or self._source_key[0] is not src or \
                self._source_key[1:] != (src.GetMTime(), sym):
            sn = self._source_normals() if sym else None
            self._ctx.set_source(src.GetPoints(), sn)
            self._source_key = (src, src.GetMTime(), sym)

    def _params(self):
        p = icpx.default_params()
        p.metric = self.Metric
        p.max_iterations = self.MaximumNumberOfIterations
        p.max_landmarks = self.MaximumNumberOfLandmarks
        p.max_distance = self.MaximumDistance
        p.check_mean_distance = int(self.CheckMeanDistance)
        p.max_mean_distance = self.MaximumMeanDistance
        p.min_normal_dot = self.MinNormalDot
        p.rotate_source_normals = int(self.RotateSourceNormals)
        return p

    def InternalUpdate(self):
        if self.Source is None or not self.Source.GetNumberOfPoints():
            return self._error("Can't execute with nullptr or empty input")
        if self.Target is None or not self.Target.GetNumberOfPoints():
            return self._error("Can't execute with nullptr or empty target")
        if self.Target._point_dtype != np.float32 or self.Source._point_dtype != np.float32:
            return self._error("We only support single-precision points for now")
        if self.Metric == vtkICP.MetricPlaneToPlane and self.Source.GetPointNormals() is None:
            return self._error("Plane to plane metric requires source normals")
        if self.Target.GetNumberOfCells() == 0:
            return self._error("No polygons to evaluate function!")
        self._upload()
        T, it, md, st = self._ctx.run(self._params())
        self.NumberOfIterations = it
        if self.CheckMeanDistance:
            self.MeanDistance = md
        self.Status = st
        self.Matrix = T
