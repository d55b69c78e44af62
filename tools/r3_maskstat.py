import sys, numpy as np, torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from based_on_point_cloud_partitions_amd import synthetic as syn
from helpers import oracle_forward
from oracle import oracle
P, W, H, focal = 150000, 1920, 1080, 1200.0   # cfg2's camera, 30 % of its Gaussians (lists shorter, geometry per splat identical)
g = syn.make_gaussians(P, seed=0)
cam = syn.ring_cameras(16, W, H, focal)[0]
st = oracle_forward(g, cam)
R = st["num_rendered"]; gx = (W + 15) // 16
keys = st["keys_sorted"]; ids = st["point_list"].astype(np.int64); tiles = (keys >> np.uint64(32)).astype(np.int64)
flags = oracle.block_flags(st)[:R]
rng = np.random.default_rng(0); sel = rng.choice(R, size=min(R, 300000), replace=False)
ids, tiles, flags = ids[sel], tiles[sel], flags[sel]
m2 = st["means2D"][ids].astype(np.float64); co = st["conic_opacity"][ids].astype(np.float64)
mx, my, cx, cy, cz, op = m2[:, 0], m2[:, 1], co[:, 0], co[:, 1], co[:, 2], co[:, 3]
tx0 = (tiles % gx) * 16.0; ty0 = (tiles // gx) * 16.0
exact = np.zeros((len(ids), 4), bool); cont = np.zeros((len(ids), 4), bool)
lx, ly = np.meshgrid(np.arange(8.0), np.arange(8.0))
t0 = 2.0 * np.log(255.0 * op)
for q in range(4):
    x0 = tx0 + 8 * (q & 1); y0 = ty0 + 8 * (q >> 1)
    px = x0[:, None, None] + lx[None]; py = y0[:, None, None] + ly[None]
    dx = mx[:, None, None] - px; dy = my[:, None, None] - py
    power = -0.5 * (cx[:, None, None] * dx * dx + cz[:, None, None] * dy * dy) - cy[:, None, None] * dx * dy
    alpha = np.minimum(0.99, op[:, None, None] * np.exp(power))
    exact[:, q] = ((alpha >= 1.0 / 255.0) & (power <= 0)).any(axis=(1, 2))
    # continuous test: min of the quadratic form over the box [x0,x0+7]x[y0,y0+7] <= t0  (convex QP over a box: check via dense sampling 57x57)
    sx = np.linspace(0, 7, 29); gxs, gys = np.meshgrid(sx, sx)
    dxc = mx[:, None, None] - (x0[:, None, None] + gxs[None]); dyc = my[:, None, None] - (y0[:, None, None] + gys[None])
    qf = cx[:, None, None] * dxc * dxc + cz[:, None, None] * dyc * dyc + 2 * cy[:, None, None] * dxc * dyc
    cont[:, q] = (qf.min(axis=(1, 2)) <= t0 + 1e-2)
blended = np.stack([(flags >> q) & 1 for q in range(4)], 1).astype(bool)
print("instances sampled", len(ids), " per instance: blended quadrants %.3f, exact(alpha>=1/255 at a pixel) %.3f, continuous box test %.3f" % (blended.sum(1).mean(), exact.sum(1).mean(), cont.sum(1).mean()))
print("instances with no blended quadrant: %.3f, with no exact quadrant %.3f" % ((blended.sum(1) == 0).mean(), (exact.sum(1) == 0).mean()))
