"""Pins the oracle's backward (restating backward.cu) against float64 autograd of an independent model."""
import numpy as np
import pytest
import torch

from helpers import oracle_forward, rel_err, scene
from oracle import oracle
import ref_autograd


def _run(P, W, H, focal, seed, bg, deg, scale_lo, scale_hi, use_precomp_color=False, use_precomp_cov=False, scale_modifier=1.0):
    g, cam = scene(P, W, H, focal, seed=seed, scale_lo=scale_lo, scale_hi=scale_hi)
    colors = None
    cov = None
    rng = np.random.default_rng(seed + 7)
    if use_precomp_color:
        colors = rng.uniform(0, 1, (P, 3)).astype(np.float32)
    if use_precomp_cov:
        A = rng.normal(0, 0.15, (P, 3, 3))
        S = A @ A.transpose(0, 2, 1) + 0.002 * np.eye(3)
        cov = np.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], 1).astype(np.float32)
    st = oracle_forward(g, cam, bg=bg, sh_degree=deg, scale_modifier=scale_modifier, colors_precomp=colors, cov3D_precomp=cov)
    dL = rng.standard_normal((3, H, W)).astype(np.float32)
    gr = oracle.backward(st, dL)

    f64 = lambda t: torch.tensor(np.asarray(t), dtype=torch.float64)
    xyz = f64(g.xyz).requires_grad_(True)
    sc = f64(g.scales).requires_grad_(True)
    rt = f64(g.rotations).requires_grad_(True)
    op = f64(g.opacities).requires_grad_(True)
    sh = f64(g.shs).requires_grad_(True)
    off = torch.zeros(P, 2, dtype=torch.float64, requires_grad=True)
    cp = f64(colors).requires_grad_(True) if colors is not None else None
    cv = f64(cov).requires_grad_(True) if cov is not None else None
    img = ref_autograd.render(xyz, sc, rt, op, sh, off, viewmatrix=f64(cam.world_view_transform),
                              projmatrix=f64(cam.full_proj_transform), campos=f64(cam.camera_center), bg=f64(bg), W=W, H=H,
                              tanfovx=cam.tanfovx, tanfovy=cam.tanfovy, deg=deg, scale_modifier=scale_modifier,
                              ranges=st["ranges"], point_list=st["point_list"], n_contrib=st["n_contrib"],
                              colors_precomp=cp, cov3D_precomp=cv)
    assert np.abs(img.detach().numpy() - st["out_color"]).max() < 2e-5
    (img * f64(dL)).sum().backward()
    tol = 3e-4
    assert rel_err(gr["dL_dmeans3D"], xyz.grad) < tol
    assert rel_err(gr["dL_dmeans2D"][:, :2], off.grad) < tol
    assert np.all(gr["dL_dmeans2D"][:, 2] == 0)
    assert rel_err(gr["dL_dopacity"], op.grad) < tol
    if colors is None:
        assert rel_err(gr["dL_dsh"], sh.grad) < tol
    else:
        assert rel_err(gr["dL_dcolors"], cp.grad) < tol
        assert np.all(gr["dL_dsh"] == 0)
    if cov is None:
        # reference quirk kept by the oracle: backward.cu:316-319 returns dL/d(mod*scale), not dL/d(scale)
        assert rel_err(gr["dL_dscales"] * scale_modifier, sc.grad) < tol
        assert rel_err(gr["dL_drotations"], rt.grad) < tol
    else:
        # the reference returns dL/d(cov3D) with the off-diagonal gradients doubled (backward.cu:215-221),
        # i.e. exactly the derivative w.r.t. the 6 stored entries
        assert rel_err(gr["dL_dcov3D"], cv.grad) < tol
    return st


@pytest.mark.parametrize("deg", [0, 1, 2, 3])
def test_backward_matches_autograd_sh_degrees(deg):
    st = _run(P=60, W=48, H=40, focal=30.0, seed=3 + deg, bg=(0.3, 0.5, 0.7), deg=deg, scale_lo=0.1, scale_hi=0.5)
    assert st["num_rendered"] > 100 and st["n_contrib"].max() > 5


def test_backward_matches_autograd_saturating():
    # big opaque Gaussians: many pixels reach T < 1e-4 (the `done` path, forward.cu:345-349)
    st = _run(P=150, W=32, H=32, focal=20.0, seed=11, bg=(0.0, 0.0, 0.0), deg=3, scale_lo=0.4, scale_hi=1.2)
    assert (st["final_T"] < 1e-3).any()


def test_backward_matches_autograd_precomputed_inputs():
    _run(P=50, W=32, H=32, focal=24.0, seed=5, bg=(0.1, 0.2, 0.3), deg=0, scale_lo=0.1, scale_hi=0.4,
         use_precomp_color=True, use_precomp_cov=True)


def test_backward_matches_autograd_scale_modifier():
    _run(P=50, W=32, H=32, focal=24.0, seed=6, bg=(0.1, 0.2, 0.3), deg=2, scale_lo=0.1, scale_hi=0.4, scale_modifier=0.7)
