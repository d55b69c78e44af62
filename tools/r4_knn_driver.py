#!/usr/bin/env python3
"""distCUDA2 on cfg2's points, 20 calls: the program to put behind `rocprofv3 --kernel-trace --stats --` for a per-kernel breakdown."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from based_on_point_cloud_partitions_amd import distCUDA2  # noqa: E402
from based_on_point_cloud_partitions_amd import synthetic as syn  # noqa: E402

pts = syn.make_gaussians(int(sys.argv[1]) if len(sys.argv) > 1 else 500_000, seed=0).xyz.cuda()
for _ in range(20):
    distCUDA2(pts)
torch.cuda.synchronize()
