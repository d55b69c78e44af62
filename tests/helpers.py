"""Shared scene builders for the parity tests (oracle side = numpy, product side = torch)."""
import numpy as np
import torch

from based_on_point_cloud_partitions_amd import synthetic as syn


def scene(P, W, H, focal, seed=0, cam_index=0, n_cams=4, scale_lo=0.004, scale_hi=0.04, **kw):
    g = syn.make_gaussians(P, seed=seed, scale_lo=scale_lo, scale_hi=scale_hi, **kw)
    cam = syn.ring_cameras(n_cams, W, H, focal)[cam_index]
    return g, cam


def oracle_forward(g, cam, bg=(0.0, 0.0, 0.0), sh_degree=3, scale_modifier=1.0, colors_precomp=None, cov3D_precomp=None):
    from oracle import oracle

    kw = {}
    if colors_precomp is None:
        kw["shs"] = g.shs.numpy()
    else:
        kw["colors_precomp"] = np.asarray(colors_precomp)
    if cov3D_precomp is None:
        kw["scales"] = g.scales.numpy()
        kw["rotations"] = g.rotations.numpy()
    else:
        kw["cov3D_precomp"] = np.asarray(cov3D_precomp)
    return oracle.forward(g.xyz.numpy(), g.opacities.numpy(), viewmatrix=cam.world_view_transform.numpy(),
                          projmatrix=cam.full_proj_transform.numpy(), campos=cam.camera_center.numpy(),
                          bg=np.asarray(bg, np.float32), W=cam.image_width, H=cam.image_height, tanfovx=cam.tanfovx,
                          tanfovy=cam.tanfovy, sh_degree=sh_degree, scale_modifier=scale_modifier, **kw)


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12)) if a.size else 0.0
