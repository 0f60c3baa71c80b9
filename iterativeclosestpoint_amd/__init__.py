"""MI355X-native ICP correspondence-and-alignment hot path (libicpx + the vtkICP surface)."""
from . import icpx, synth  # noqa: F401
from .vtkicp import vtkICP, vtkPolyData, icp_loop  # noqa: F401
from .vtkfilter import vtkPolyDataCorrespondenceFilter  # noqa: F401
from .vtkintersection import vtkImplicitPolyDataIntersection  # noqa: F401

__all__ = ["icpx", "synth", "vtkICP", "vtkPolyData", "icp_loop", "vtkPolyDataCorrespondenceFilter",
           "vtkImplicitPolyDataIntersection"]
