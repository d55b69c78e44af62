#!/usr/bin/env python3
"""Step time against the camera's distance: the further away, the fewer tiles the scene covers and the longer their lists (a distant
dense region seen from a partition's camera).  Lists beyond 8192 entries take the in-place bitonic path of tile_sort_mid_kernel.
python tools/r4_far_camera.py [P] [radius,radius,...] [WxH]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from based_on_point_cloud_partitions_amd import synthetic as syn  # noqa: E402
from based_on_point_cloud_partitions_amd import _lib  # noqa: E402
from based_on_point_cloud_partitions_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 500_000
dev = torch.device("cuda:0")
W, H = (int(v) for v in sys.argv[3].split("x")) if len(sys.argv) > 3 else (1920, 1080)
focal = 1200.0 * W / 1920.0
g = syn.make_gaussians(P, seed=0)
xyz, scales, rots, opac, shs = [t.to(dev).requires_grad_(True) for t in (g.xyz, g.scales, g.rotations, g.opacities, g.shs)]
dL = torch.randn(3, H, W, generator=torch.Generator().manual_seed(1)).to(dev)
bg = torch.zeros(3, device=dev)
RADII = [float(v) for v in sys.argv[2].split(",")] if len(sys.argv) > 2 else [6.0, 12.0, 25.0, 50.0, 100.0, 200.0, 400.0]
for radius in RADII:
    c = syn.ring_cameras(4, W, H, focal, radius=radius, height=radius / 3.0)[0]
    r = GaussianRasterizer(GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=c.tanfovx, tanfovy=c.tanfovy, bg=bg,
        scale_modifier=1.0, viewmatrix=c.world_view_transform.to(dev), projmatrix=c.full_proj_transform.to(dev), sh_degree=3,
        campos=c.camera_center.to(dev), prefiltered=False, debug=False))

    def step():
        ss = torch.zeros_like(xyz, requires_grad=True)
        image, radii = r(means3D=xyz, means2D=ss, opacities=opac, shs=shs, scales=scales, rotations=rots)
        image.backward(dL)
        for p in (xyz, scales, rots, opac, shs):
            p.grad = None
        return radii

    for _ in range(3):
        radii = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / n
    _lib.profile_enable(True)
    step()
    torch.cuda.synchronize()
    _lib.profile_enable(False)
    st = _lib.profile_collect()
    vis = int((radii > 0).sum())
    print(f"radius {radius:6.1f}: {ms:8.3f} ms/step  visible {vis}  stages(ms) " + " ".join(f"{k}={v[0] / max(v[1], 1):.3f}" for k, v in st.items() if v[1]), flush=True)
