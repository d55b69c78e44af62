"""Drop-in for `simple_knn._C` (submodules/simple-knn/ext.cpp:15-17): exports distCUDA2."""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from based_on_point_cloud_partitions_amd.knn import distCUDA2  # noqa: E402,F401
