#!/usr/bin/env python3
"""kNN init: the wave-autonomous kernel (GSR_KNN_WAVE=1, default) against the workgroup-cooperative one, uniform and clustered points.
usage (GPU box): python tools/r4_knn.py"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from based_on_point_cloud_partitions_amd import _lib, distCUDA2  # noqa: E402
from based_on_point_cloud_partitions_amd import synthetic as syn  # noqa: E402


def run(name, pts):
    for wave in (1, 0, 1, 0):
        _lib.set_variant("KNN_WAVE", wave)
        cnt = torch.zeros(1, dtype=torch.int64, device="cuda")
        _lib.knn_counter(cnt)
        distCUDA2(pts)
        torch.cuda.synchronize()
        _lib.knn_counter(None)
        t0 = time.perf_counter()
        for _ in range(10):
            distCUDA2(pts)
        torch.cuda.synchronize()
        print(f"{name} P={pts.shape[0]} KNN_WAVE={wave}: {1e2 * (time.perf_counter() - t0):.3f} ms, {int(cnt.item()) / pts.shape[0]:.0f} candidate distances per point")


rng = np.random.default_rng(0)
run("cfg2 uniform box", syn.make_gaussians(500_000, seed=0).xyz.cuda())
run("cfg4 uniform box", syn.make_gaussians(2_000_000, seed=0, box_min=(-8.0, -8.0, -1.0), box_max=(8.0, 8.0, 1.0)).xyz.cuda())
P = 500_000
blobs = (rng.standard_normal((P, 3)) * 0.05 + rng.integers(0, 50, (P, 1)) * 1.0).astype(np.float32)
run("clustered (50 blobs)", torch.from_numpy(blobs).cuda())
surf = rng.uniform(-4, 4, (P, 3)).astype(np.float32)
surf[:, 2] = 0.1 * np.sin(surf[:, 0]) + 0.001 * rng.standard_normal(P).astype(np.float32)
run("surface (2-D manifold)", torch.from_numpy(surf).cuda())
