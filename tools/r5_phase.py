"""Phase profile of the blend backward's log walk (a library built with -DGSR_LOG_PROFILE=1, selected through GSR_LIBRARY):
usage: GSR_LIBRARY=tools/scratch/ab_<tag>_<n>/.../libgsrast_mi355x.so python tools/r5_phase.py [config]"""
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


for c in cams[:4]:
    step(c)
lib.gsr_debug_blend_profile(buf)
for c in cams:
    step(c)
built = lib.gsr_debug_blend_profile(buf)
v = [int(x) for x in buf]
names = ["whole wave", "prologue", "staging loads + tests", "wait: previous batch flushed + LDS writes + barrier", "zero rows + end", "walk",
         "wait: end-of-walk barrier", "flush"]
fn = ["whole wave", "staging loads + reach tests", "barriers + LDS writes", "walk + window stores", "prologue + epilogue"]
print(json.dumps({"kernel": "render_forward", "wave_cycles": dict(zip(fn, v[8:13])),
                  "share_of_wave_time": {n: round(x / max(v[8], 1), 4) for n, x in zip(fn[1:], v[9:13])}}))
print(json.dumps({"built_with_profile": built, "launches": len(cams), "wave_cycles": dict(zip(names, v)),
                  "share_of_wave_time": {n: round(x / max(v[0], 1), 4) for n, x in zip(names[1:], v[1:])}}))
