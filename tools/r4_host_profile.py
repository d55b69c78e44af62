#!/usr/bin/env python3
"""Where the HOST time of a rasterizer step goes (cfg1: 10k Gaussians at 256x256, where a slow host core makes the step host-bound):
cProfile over N steps of forward + backward through the autograd Function, sorted by own time.  python tools/r4_host_profile.py [steps] [config]"""
import cProfile
import os
import pstats
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from based_on_point_cloud_partitions_amd import synthetic as syn  # noqa: E402
from based_on_point_cloud_partitions_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
cfg = syn.CONFIGS[sys.argv[2] if len(sys.argv) > 2 else "cfg1"]
dev = torch.device("cuda:0")
P, W, H = cfg["P"], cfg["W"], cfg["H"]
g = syn.make_gaussians(P, seed=0)
cams = syn.ring_cameras(cfg["cameras"], W, H, cfg["focal"])
dL = torch.randn(3, H, W, generator=torch.Generator().manual_seed(1)).to(dev)
bg = torch.zeros(3, device=dev)
xyz, scales, rots, opac, shs = [t.to(dev).requires_grad_(True) for t in (g.xyz, g.scales, g.rotations, g.opacities, g.shs)]
params = [xyz, scales, rots, opac, shs]
rs = [GaussianRasterizer(GaussianRasterizationSettings(image_height=H, image_width=W, tanfovx=c.tanfovx, tanfovy=c.tanfovy, bg=bg,
      scale_modifier=1.0, viewmatrix=c.world_view_transform.to(dev), projmatrix=c.full_proj_transform.to(dev), sh_degree=3,
      campos=c.camera_center.to(dev), prefiltered=False, debug=False)) for c in cams]


def step(i):
    ss = torch.zeros_like(xyz, requires_grad=True)
    image, radii = rs[i % len(rs)](means3D=xyz, means2D=ss, opacities=opac, shs=shs, scales=scales, rotations=rots)
    image.backward(dL)
    for p in params:
        p.grad = None


for i in range(200):
    step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(steps):
    step(i)
torch.cuda.synchronize()
print("plain loop: %.1f us per step" % (1e6 * (time.perf_counter() - t0) / steps))
pr = cProfile.Profile()
pr.enable()
for i in range(steps):
    step(i)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(22)
