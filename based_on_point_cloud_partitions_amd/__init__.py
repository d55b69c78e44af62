"""MI355X-native Gaussian rasterizer + kNN init behind the reference's operator surface.

Public names mirror the two import sites of the reference:
  gaussian_renderer/__init__.py:14  from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer
  scene/gaussian_model.py:20        from simple_knn._C import distCUDA2
(put `based_on_point_cloud_partitions_amd/dropin` on PYTHONPATH to get those exact module names).
"""
from ._lib import GsrError, SO_PATH, load as load_library  # noqa: F401
from .knn import distCUDA2  # noqa: F401
from .rasterizer import (GaussianRasterizationSettings, GaussianRasterizer, mark_visible, rasterize_gaussians,  # noqa: F401
                         rasterize_gaussians_backward, rasterize_gaussians_forward)
