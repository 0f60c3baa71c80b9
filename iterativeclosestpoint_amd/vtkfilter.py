"""The reference's correspondence filter as a class over libicpx.

`vtkPolyDataCorrespondenceFilter` here has the methods, defaults and failure behaviour
of the reference's filter (ICP/vtkPolyDataCorrespondenceFilter.h:11-107,
ICP/vtkPolyDataCorrespondenceFilter.cxx:349-547): two inputs (port 0 source, port 1
target), two outputs (port 0: the accepted source points after the transform, with
"Normals" when IncludeOriginNormals is on; port 1: their closest points on the other
mesh with that mesh's pseudo-normals), a transform applied to the searching cloud, the
stride rule for MaximumNumberOfLandmarks, MaximumDistance, MinNormalDot and the search
direction.  The search itself is `icpx_correspond`.

Two things differ from the reference as written, both where the reference reads memory it
never wrote: landmark normals are the locator's (the reference self-assigns,
.cxx:126-132) and MinNormalDot defaults to 0.0 and is tested against the source normal
(.cxx:286, .h:66-71).  With double output precision the rows come from
`icpx_correspond_f64`; as in the reference (.cxx:519-531) that path has neither origin
normals nor the normal test.
"""
import sys

import numpy as np

from . import icpx
from .vtkicp import vtkPolyData

VTK_INT_MAX = 2147483647


class vtkPolyDataCorrespondenceFilter:
    SourceToTarget = 0
    TargetToSource = 1
    # vtkAlgorithm::DesiredOutputPrecision
    SINGLE_PRECISION = 0
    DOUBLE_PRECISION = 1
    DEFAULT_PRECISION = 2

    def __init__(self, device=0):
        # defaults: ICP/vtkPolyDataCorrespondenceFilter.cxx:349-366
        self.MaximumDistance = 10.0
        self.MaximumNumberOfLandmarks = VTK_INT_MAX
        self.OutputPrecision = self.DEFAULT_PRECISION
        self.MinNormalDot = 0.0
        self.IncludeOriginNormals = False
        self.SearchDirection = self.SourceToTarget
        self.Transform = np.eye(4)
        self._inputs = [None, None]
        self._outputs = [vtkPolyData(), vtkPolyData()]
        self._device = device
        self._ctx = None
        self._target_key = None
        self._source_key = None
        self._modified = True

    @staticmethod
    def New():
        return vtkPolyDataCorrespondenceFilter()

    # --- pipeline ---------------------------------------------------------------------
    def SetInputData(self, port, pd=None):
        if pd is None:  # SetInputData(pd) means port 0
            port, pd = 0, port
        if port not in (0, 1):
            raise IndexError("vtkPolyDataCorrespondenceFilter has input ports 0 and 1")
        self._inputs[port] = pd
        self._modified = True

    def GetInput(self, port=0):
        return self._inputs[port]

    def GetOutput(self, port=0):
        return self._outputs[port]

    def GetNumberOfInputPorts(self):
        return 2

    def GetNumberOfOutputPorts(self):
        return 2

    def Modified(self):
        """Re-execute on the next Update() and read both inputs' current arrays again."""
        self._modified = True
        self._target_key = None
        self._source_key = None

    # --- parameters -------------------------------------------------------------------
    def SetTransform(self, t):
        """A 4x4 (row-major) or anything with GetMatrix(); None means identity."""
        if t is None:
            M = np.eye(4)
        elif hasattr(t, "GetMatrix"):
            M = np.asarray(t.GetMatrix(), np.float64)
        else:
            M = np.asarray(t, np.float64)
        self.Transform = np.ascontiguousarray(M, np.float64).reshape(4, 4)
        self._modified = True

    def GetTransform(self):
        return self.Transform

    def SetMaximumNumberOfLandmarks(self, n):
        self.MaximumNumberOfLandmarks = int(n)
        self._modified = True

    def GetMaximumNumberOfLandmarks(self):
        return self.MaximumNumberOfLandmarks

    def SetOutputPrecision(self, p):
        self.OutputPrecision = int(p)
        self._modified = True

    def GetOutputPrecision(self):
        return self.OutputPrecision

    def SetMaximumDistance(self, d):
        self.MaximumDistance = float(d)
        self._modified = True

    def GetMaximumDistance(self):
        return self.MaximumDistance

    def SetMinNormalDot(self, d):
        self.MinNormalDot = float(d)
        self._modified = True

    def GetMinNormalDot(self):
        return self.MinNormalDot

    def SetSearchDirection(self, d):
        self.SearchDirection = min(max(int(d), self.SourceToTarget), self.TargetToSource)
        self._modified = True

    def GetSearchDirection(self):
        return self.SearchDirection

    def SetSourceToTarget(self):
        self.SetSearchDirection(self.SourceToTarget)

    def SetTargetToSource(self):
        self.SetSearchDirection(self.TargetToSource)

    def SetIncludeOriginNormals(self, v):
        self.IncludeOriginNormals = bool(v)
        self._modified = True

    def GetIncludeOriginNormals(self):
        return self.IncludeOriginNormals

    def IncludeOriginNormalsOn(self):
        self.SetIncludeOriginNormals(True)

    def IncludeOriginNormalsOff(self):
        self.SetIncludeOriginNormals(False)

    def PrintSelf(self, os=sys.stdout, indent=""):
        for k in ("MaximumDistance", "MinNormalDot", "MaximumNumberOfLandmarks", "SearchDirection",
                  "OutputPrecision", "IncludeOriginNormals"):
            os.write("%s%s: %s\n" % (indent, k, getattr(self, k)))

    # --- execution --------------------------------------------------------------------
    def Update(self):
        if self._modified:
            if self.RequestData():
                self._modified = False

    def _fail(self, msg):
        # the reference's vtkDebugMacro + return 0: outputs stay as they were
        sys.stderr.write("vtkPolyDataCorrespondenceFilter: %s\n" % msg)
        return 0

    def RequestData(self):
        """ICP/vtkPolyDataCorrespondenceFilter.cxx:397-547.  Returns 1 on success, 0 otherwise."""
        in0, in1 = self._inputs
        if in0 is None or in1 is None:
            return 0
        if in0.GetPoints() is None or in1.GetPoints() is None:
            return 0
        if in0._point_dtype != in1._point_dtype:
            return self._fail("Input points must be of the same type")
        if self.SearchDirection == self.SourceToTarget:
            source, target = in0, in1
        else:
            source, target = in1, in0
        if source._point_dtype != np.float32:
            return self._fail("We only support float input data")
        double = self.OutputPrecision == self.DOUBLE_PRECISION
        if target.GetNumberOfCells() == 0:
            return self._fail("No polygons to evaluate function!")
        sn = None
        own_normals = self.IncludeOriginNormals and source.GetNumberOfCells() > 0
        if self.IncludeOriginNormals and not own_normals:
            sn = source.GetPointNormals()
            if sn is None or len(sn) != source.GetNumberOfPoints():
                return self._fail("Number of vertex normals does not match the number of points")

        if self._ctx is None:
            self._ctx = icpx.Context(self._device)
        # device copies are keyed on (object, its modification count); see vtkICP._upload
        if self._target_key is None or self._target_key[0] is not target or \
                self._target_key[1] != target.GetMTime():
            self._ctx.set_target(target.GetPoints(), target.GetPolys(), target.GetCellNormals())
            self._target_key = (target, target.GetMTime())
        if self._source_key is None or self._source_key[0] is not source or \
                self._source_key[1:] != (source.GetMTime(), self.IncludeOriginNormals):
            if own_normals:
                # the filter's own vtkPolyDataNormals (.cxx:356-359,457-464), on the device
                sn = self._ctx.point_normals(source.GetPoints(), source.GetPolys())
            self._ctx.set_source(source.GetPoints(), sn)
            self._source_key = (source, source.GetMTime(), self.IncludeOriginNormals)

        n = source.GetNumberOfPoints()
        step = n // self.MaximumNumberOfLandmarks if n > self.MaximumNumberOfLandmarks else 1
        if double:
            # Functor with double outputs (.cxx:525): no origin normals, no normal test
            rows = self._ctx.correspond_f64(self.Transform, step=step,
                                            max_dist=self.MaximumDistance)
        else:
            rows = self._ctx.correspond(self.Transform, step=step, max_dist=self.MaximumDistance,
                                        min_dot=self.MinNormalDot,
                                        use_source_normals=self.IncludeOriginNormals)
        out0 = vtkPolyData(rows["origin"])
        if self.IncludeOriginNormals and not double:
            out0.SetPointNormals(rows["origin_normal"])
        out1 = vtkPolyData(rows["closest"])
        out1.SetPointNormals(rows["normal"])
        self._outputs = [out0, out1]
        self.Cells, self.Distances, self.SourceIndices = rows["cell"], rows["dist"], rows["src_index"]
        return 1
