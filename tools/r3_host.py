"""Host-side time of one rasterizer step (cfg2): how long Python needs from the forward's return to the end of backward's enqueue,
and how long the whole loop body takes on the host.  python tools/r3_host.py [steps]"""
import sys, time
import numpy as np, torch
sys.path.insert(0, ".")
from based_on_point_cloud_partitions_amd import synthetic as syn
from based_on_point_cloud_partitions_amd.rasterizer import GaussianRasterizationSettings, GaussianRasterizer

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
dev = torch.device("cuda:0")
cfg = syn.CONFIGS["cfg2"]
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
tf, tb, tl = [], [], []
for i in range(steps + 50):
    t0 = time.perf_counter()
    ss = torch.zeros_like(xyz, requires_grad=True)
    image, radii = rs[i % len(rs)](means3D=xyz, means2D=ss, opacities=opac, shs=shs, scales=scales, rotations=rots)
    t1 = time.perf_counter()
    image.backward(dL)
    for p in params:
        p.grad = None
    t2 = time.perf_counter()
    if i >= 50:
        tf.append(t1 - t0); tb.append(t2 - t1); tl.append(t2 - t0)
torch.cuda.synchronize()
f = lambda a: "median %.0f  p10 %.0f  p90 %.0f us" % (1e6 * np.median(a), 1e6 * np.percentile(a, 10), 1e6 * np.percentile(a, 90))
print("forward call (includes the wait for num_rendered):", f(tf))
print("backward call (host only):", f(tb))
print("loop body:", f(tl))
