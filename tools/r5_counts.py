"""Counts of the blend backward's log walk (a library built with -DGSR_LOG_PROFILE=2, selected through GSR_LIBRARY): entries walked per
wave, how many with both / one / none of the wave's quadrants, blended pixels, staged entries.
usage: GSR_LIBRARY=tools/scratch/ab_<tag>_<n>/.../libgsrast_mi355x.so python tools/r5_counts.py [config] [GSR_BWD_NQ]"""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from based_on_point_cloud_partitions_amd import GaussianRasterizationSettings, GaussianRasterizer, _lib  # noqa: E402
from based_on_point_cloud_partitions_amd import synthetic as syn  # noqa: E402

cfg = syn.CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "cfg2"]
dev = torch.device("cuda", 0)
g = syn.make_gaussians(cfg["P"], seed=0).to(dev)
cams = syn.ring_cameras(cfg["cameras"], cfg["W"], cfg["H"], cfg["focal"])
dL = torch.randn(3, cfg["H"], cfg["W"], generator=torch.Generator().manual_seed(1)).to(dev)
bg = torch.zeros(3, device=dev)
params = [t.clone().requires_grad_(True) for t in (g.xyz, g.scales, g.rotations, g.opacities, g.shs)]
lib = _lib.load()
buf = (ctypes.c_uint64 * 16)()


def step(c):
    r = GaussianRasterizer(GaussianRasterizationSettings(
        image_height=cfg["H"], image_width=cfg["W"], tanfovx=c.tanfovx, tanfovy=c.tanfovy, bg=bg, scale_modifier=1.0,
        viewmatrix=c.world_view_transform.to(dev), projmatrix=c.full_proj_transform.to(dev), sh_degree=3, campos=c.camera_center.to(dev),
        prefiltered=False, debug=False))
    img, _ = r(means3D=params[0], means2D=torch.zeros_like(params[0], requires_grad=True), opacities=params[3], shs=params[4],
               scales=params[1], rotations=params[2])
    img.backward(dL)


step(cams[0])
lib.gsr_debug_blend_profile(buf)
n = 4
for c in cams[:n]:
    step(c)
built = lib.gsr_debug_blend_profile(buf)
v = [int(x) / n for x in buf]
names = ["-", "entries walked (per wave)", "both quadrants", "one quadrant", "none", "blended pixels", "staged entries with a quadrant bit", "quadrant bits staged"]
print(json.dumps({"built": built, "config": sys.argv[1] if len(sys.argv) > 1 else "cfg2", "per_launch": dict(zip(names[1:], v[1:8]))}))
