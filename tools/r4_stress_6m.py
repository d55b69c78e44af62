#!/usr/bin/env python3
"""One-off stress run at the reference's densification ceiling (train.py:146: 6,000,000 Gaussians), 1080p, one camera: full oracle parity
(forward bit-exact in the default list mode, two-stage gradient check) + step time.  usage (GPU box): python tools/r4_stress_6m.py [P] [box half-width]"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import assert_forward_bit_exact, hip_backward_raw, hip_forward_raw, oracle_forward, two_stage_check  # noqa: E402

from based_on_point_cloud_partitions_amd import synthetic as syn  # noqa: E402

P = int(sys.argv[1]) if len(sys.argv) > 1 else 6_000_000
W, H = 1920, 1080
HALF = float(sys.argv[2]) if len(sys.argv) > 2 else 12.0  # half-width of the box: 12 = sparse (short lists), 4 = config 2's box (every second list beyond 8192 entries)
g = syn.make_gaussians(P, seed=0, box_min=(-HALF, -HALF, -1.0), box_max=(HALF, HALF, 1.0))
cam = syn.ring_cameras(16, W, H, 1200.0)[3]
t0 = time.time()
st = oracle_forward(g, cam)
ln = st["ranges"][:, 1] - st["ranges"][:, 0]
print(f"oracle forward {time.time() - t0:.1f} s: R = {st['num_rendered']}, visible {int((st['radii'] > 0).sum())}, tile lists beyond 8192 entries: "
      f"{int((ln > 8192).sum())} of {ln.size}, longest {int(ln.max())}", flush=True)
ctx = hip_forward_raw(g, cam)
assert_forward_bit_exact(ctx, st)
print("forward bit-exact", flush=True)
dL = np.random.default_rng(1).standard_normal((3, H, W)).astype(np.float32)
stats = two_stage_check(st, hip_backward_raw(ctx, dL, capture=True), dL)
print("gradients:", {k: round(v["worst_ratio"], 3) for k, v in stats.items()}, flush=True)
# timing through the operator
from based_on_point_cloud_partitions_amd import GaussianRasterizationSettings, GaussianRasterizer  # noqa: E402

dev = "cuda"
r = GaussianRasterizer(GaussianRasterizationSettings(
    image_height=H, image_width=W, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, bg=torch.zeros(3, device=dev), scale_modifier=1.0,
    viewmatrix=cam.world_view_transform.to(dev), projmatrix=cam.full_proj_transform.to(dev), sh_degree=3, campos=cam.camera_center.to(dev),
    prefiltered=False, debug=False))
leaf = lambda t: t.to(dev).requires_grad_(True)
xyz, sc, rot, op, sh = leaf(g.xyz), leaf(g.scales), leaf(g.rotations), leaf(g.opacities), leaf(g.shs)
dLd = torch.from_numpy(dL).to(dev)


def step():
    ss = torch.zeros_like(xyz, requires_grad=True)
    img, _ = r(means3D=xyz, means2D=ss, opacities=op, shs=sh, scales=sc, rotations=rot)
    img.backward(dLd)
    for p in (xyz, sc, rot, op, sh):
        p.grad = None


for _ in range(5):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    step()
torch.cuda.synchronize()
print(f"P = {P}: {1e3 * (time.perf_counter() - t0) / 20:.3f} ms per forward + backward; peak device memory {torch.cuda.max_memory_allocated() / 2**30:.2f} GiB")
