#!/usr/bin/env python3
"""Probe (not a benchmark of the product): how much faster do raster steps get when kernels of DIFFERENT steps may overlap on the GPU?
Two host threads, each with its own HIP stream, run independent forward + backward steps of cfg2 (different cameras); compared with one
thread doing the same number of steps.  An upper bound for what overlapping the small latency-bound binning kernels of one step with the
blend kernels could buy (a training loop cannot pipeline steps: the optimizer sits between them)."""
import os
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from based_on_point_cloud_partitions_amd import GaussianRasterizationSettings, GaussianRasterizer  # noqa: E402
from based_on_point_cloud_partitions_amd import synthetic as syn  # noqa: E402

dev = torch.device("cuda", 0)
cfg = syn.CONFIGS["cfg2"]
g = syn.make_gaussians(cfg["P"], seed=0)
cams = syn.ring_cameras(16, cfg["W"], cfg["H"], cfg["focal"])
dL = torch.randn(3, cfg["H"], cfg["W"], generator=torch.Generator().manual_seed(1)).to(dev)
bg = torch.zeros(3, device=dev)


FWD_ONLY = len(sys.argv) > 1 and sys.argv[1] == "fwd"


def make(worker):
    leaf = lambda t: t.to(dev).clone().requires_grad_(True)
    ps = [leaf(g.xyz), leaf(g.scales), leaf(g.rotations), leaf(g.opacities), leaf(g.shs)]
    rs = [GaussianRasterizer(GaussianRasterizationSettings(
        image_height=cfg["H"], image_width=cfg["W"], tanfovx=c.tanfovx, tanfovy=c.tanfovy, bg=bg, scale_modifier=1.0,
        viewmatrix=c.world_view_transform.to(dev), projmatrix=c.full_proj_transform.to(dev), sh_degree=3, campos=c.camera_center.to(dev),
        prefiltered=False, debug=False)) for c in cams]

    def step(i):
        xyz, sc, rot, op, sh = ps
        ss = torch.zeros_like(xyz, requires_grad=True)
        img, _ = rs[(i + 8 * worker) % 16](means3D=xyz, means2D=ss, opacities=op, shs=sh, scales=sc, rotations=rot)
        if not FWD_ONLY:
            img.backward(dL)
            for p in ps:
                p.grad = None
    return step


def run(nthreads, steps):
    fns = [make(w) for w in range(nthreads)]
    streams = [torch.cuda.Stream(device=dev) for _ in range(nthreads)]

    def work(w):
        with torch.cuda.stream(streams[w]):
            for i in range(steps):
                fns[w](i)
    for w in range(nthreads):  # warm
        with torch.cuda.stream(streams[w]):
            for i in range(8):
                fns[w](i)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ts = [threading.Thread(target=work, args=(w,)) for w in range(nthreads)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"{nthreads} stream(s): {1e3 * dt / (steps * nthreads):.4f} ms per step ({steps * nthreads / dt:.0f} it/s)")


for n in (1, 2, 1, 2, 3):
    run(n, 400)
