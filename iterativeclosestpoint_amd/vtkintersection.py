"""The reference's line/mesh intersection class over libicpx.

`vtkImplicitPolyDataIntersection` here has the methods and defaults of the reference's class
(ICP/vtkImplicitPolyDataIntersection.h:31-150, ICP/vtkImplicitPolyDataIntersection.cxx:29-337):
SetInput(polydata), IntersectWithLine(p1, p2) and the NoValue / NoGradient /
NoClosestPoint / Tolerance members.  IntersectWithLine returns the first three for a segment
that misses, whatever the caller has set them to; Tolerance is kept for the interface's sake:
the library's intersection test runs at the reference's default of 1e-12 (.cxx:148) and a
changed value is not passed on.  Among several cells hit at the same parameter the library
takes the smallest cell id, where vtkCellLocator's order of buckets decides in the reference.  The search is `icpx_intersect_lines`; `IntersectWithLines`
(plural) takes many segments in one call, which is how a GPU wants them.
"""
import sys

import numpy as np

from . import icpx

# vtkDataObject::AttributeTypes as the six-argument overload reports them
CELL, EDGE, VERTEX, NUMBER_OF_ATTRIBUTE_TYPES = 1, 5, 4, 7


class vtkImplicitPolyDataIntersection:
    def __init__(self, device=0):
        # defaults: ICP/vtkImplicitPolyDataIntersection.cxx:29-44
        self.NoClosestPoint = np.zeros(3)
        self.NoGradient = np.array([0.0, 0.0, 1.0])
        self.NoValue = 0.0
        self.Tolerance = 1e-12
        self.Input = None
        self._device = device
        self._ctx = None
        self._key = None

    @staticmethod
    def New():
        return vtkImplicitPolyDataIntersection()

    def SetInput(self, pd):
        """The triangles of `pd` become the mesh to intersect (.cxx:47-74); NoValue becomes
        the length of its bounds."""
        self.Input = pd
        if pd is not None and pd.GetNumberOfPoints() and pd.GetNumberOfCells():
            p = np.asarray(pd.GetPoints(), np.float64)
            self.NoValue = float(np.linalg.norm(p.max(0) - p.min(0)))

    def GetInput(self):
        return self.Input

    def _upload(self):
        pd = self.Input
        if self._ctx is None:
            self._ctx = icpx.Context(self._device)
        if self._key is None or self._key[0] is not pd or self._key[1] != pd.GetMTime():
            self._ctx.set_target(pd.GetPoints(), pd.GetPolys(), pd.GetCellNormals())
            self._key = (pd, pd.GetMTime())

    def IntersectWithLines(self, p1, p2):
        """Many segments at once: dict of arrays hit, ret, closest, normal, point, cell, kind
        (see icpx_intersect_lines), or None after the reference's error for a missing mesh."""
        if self.Input is None or self.Input.GetNumberOfCells() == 0:
            sys.stderr.write("ERROR: vtkImplicitPolyDataIntersection: No polygons to intersect!\n")
            return None
        self._upload()
        return self._ctx.intersect_lines(p1, p2)

    def IntersectWithLine(self, p1, p2):
        """(found, ret, closestPoint, g, type) for one segment, as the reference's
        IntersectWithLine fills its arguments (.cxx:110-337)."""
        r = self.IntersectWithLines(np.asarray(p1, np.float64).reshape(1, 3),
                                    np.asarray(p2, np.float64).reshape(1, 3))
        if r is None:
            return 0, self.NoValue, self.NoClosestPoint.copy(), self.NoGradient.copy(), \
                NUMBER_OF_ATTRIBUTE_TYPES
        if int(r["hit"][0]) == 0:
            # no intersection: the instance's own members, as the reference returns them
            # (.cxx:114-124); a caller may have changed them
            return 0, self.NoValue, np.array(self.NoClosestPoint, np.float64), \
                np.array(self.NoGradient, np.float64), NUMBER_OF_ATTRIBUTE_TYPES
        kind = {0: CELL, 1: EDGE, 2: VERTEX}.get(int(r["kind"][0]), NUMBER_OF_ATTRIBUTE_TYPES)
        return int(r["hit"][0]), float(r["ret"][0]), r["closest"][0], r["normal"][0], kind
