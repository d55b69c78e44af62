"""Drop-in for the reference's `diff_gaussian_rasterization` package (same public names,
submodules/diff-gaussian-rasterization/diff_gaussian_rasterization/__init__.py), backed by the MI355X HIP library."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from based_on_point_cloud_partitions_amd.rasterizer import (  # noqa: E402,F401
    _C, GaussianRasterizationSettings, GaussianRasterizer, _RasterizeGaussians, cpu_deep_copy_tuple, rasterize_gaussians)
