"""End-to-end check of the widened path: a few training iterations with every fused piece (raw-parameter render, fused loss,
densification statistics, Adam, pruning) against the same iterations in the reference's data flow -- torch activations,
`GaussianRasterizer`, `(1-l)*l1 + l*(1-ssim)` from separate calls, torch's masked statistics updates, `torch.optim.Adam`
(gaussian_renderer/__init__.py:18-100, train.py:107-157, scene/gaussian_model.py:102-122,164-175,433-435).
"""
import math

import numpy as np
import pytest
import torch

from helpers import scene

pytestmark = pytest.mark.gpu

LRS = dict(xyz=0.00016, f_dc=0.0025, f_rest=0.0025 / 20.0, opacity=0.05, scaling=0.005, rotation=0.001)  # arguments/__init__.py


class Model:
    """The slice of GaussianModel the loop touches (scene/gaussian_model.py:42-66,102-122,158-175)."""

    def __init__(self, g, adam_cls, device="cuda"):
        P = g.xyz.shape[0]
        nn_p = lambda t: torch.nn.Parameter(t.to(device).contiguous().requires_grad_(True))
        self.active_sh_degree = 3
        self._xyz = nn_p(g.xyz)
        self._features_dc = nn_p(g.shs[:, :1])
        self._features_rest = nn_p(g.shs[:, 1:])
        self._opacity = nn_p(torch.logit(g.opacities.clamp(1e-4, 1 - 1e-4)))
        self._scaling = nn_p(torch.log(g.scales))
        self._rotation = nn_p(g.rotations * 1.3)
        self.max_radii2D = torch.zeros(P, device=device)
        self.xyz_gradient_accum = torch.zeros(P, 1, device=device)
        self.denom = torch.zeros(P, 1, device=device)
        groups = [{"params": [getattr(self, a)], "lr": LRS[n], "name": n} for a, n in
                  (("_xyz", "xyz"), ("_features_dc", "f_dc"), ("_features_rest", "f_rest"), ("_opacity", "opacity"),
                   ("_scaling", "scaling"), ("_rotation", "rotation"))]
        self.optimizer = adam_cls(groups, lr=0.0, eps=1e-15)

    get_xyz = property(lambda s: s._xyz)
    get_scaling = property(lambda s: torch.exp(s._scaling))
    get_rotation = property(lambda s: torch.nn.functional.normalize(s._rotation))
    get_opacity = property(lambda s: torch.sigmoid(s._opacity))
    get_features = property(lambda s: torch.cat((s._features_dc, s._features_rest), dim=1))

    def params(self):
        return [self._xyz, self._features_dc, self._features_rest, self._opacity, self._scaling, self._rotation]


class Cam:
    def __init__(self, cam, device="cuda"):
        self.image_height, self.image_width = cam.image_height, cam.image_width
        self.FoVx, self.FoVy = 2 * math.atan(cam.tanfovx), 2 * math.atan(cam.tanfovy)
        self.world_view_transform = cam.world_view_transform.to(device)
        self.full_proj_transform = cam.full_proj_transform.to(device)
        self.camera_center = cam.camera_center.to(device)


class Pipe:
    convert_SHs_python = compute_cov3D_python = debug = False


def _iteration(model, cam, gt, bg, fused):
    from based_on_point_cloud_partitions_amd import densify, loss as L, renderer

    out = renderer.render(cam, model, Pipe(), bg, 1.0, fused=fused)
    image, vsp, radii = out["render"], out["viewspace_points"], out["radii"]
    if fused:
        loss = L.l1_ssim_loss(image, gt, 0.2)
    else:
        loss = (1.0 - 0.2) * L.l1_loss(image, gt) + 0.2 * (1.0 - L.ssim(image, gt))  # train.py:116-117
    loss.backward()
    with torch.no_grad():
        if fused:
            densify.add_densification_stats(model, vsp, radii)
        else:  # train.py:140-142, gaussian_model.py:433-435
            vis = radii > 0
            model.max_radii2D[vis] = torch.max(model.max_radii2D[vis], radii[vis])
            model.xyz_gradient_accum[vis] += torch.norm(vsp.grad[vis, :2], dim=-1, keepdim=True)
            model.denom[vis] += 1
        model.optimizer.step()
        model.optimizer.zero_grad(set_to_none=True)
    return float(loss.detach())


def test_fused_iterations_track_the_reference_dataflow_and_the_loss_goes_down():
    from based_on_point_cloud_partitions_amd import densify, renderer
    from based_on_point_cloud_partitions_amd.optim import Adam

    P, W, H = 4000, 160, 112
    truth, cam0 = scene(P, W, H, 110.0, seed=31, scale_lo=0.02, scale_hi=0.15)
    cam = Cam(cam0)
    bg = torch.tensor([0.1, 0.1, 0.1], device="cuda")
    with torch.no_grad():
        gt = renderer.render(cam, Model(truth, Adam), Pipe(), bg)["render"].clone()
    # start: colours and opacities perturbed
    gen = torch.Generator().manual_seed(5)
    start = truth._replace(shs=truth.shs + 0.3 * torch.randn(truth.shs.shape, generator=gen),
                           opacities=(truth.opacities * 0.6).clamp(0.02, 0.98))
    a, b = Model(start, Adam), Model(start, torch.optim.Adam)
    la = [_iteration(a, cam, gt, bg, True) for _ in range(3)]
    lb = [_iteration(b, cam, gt, bg, False) for _ in range(3)]
    np.testing.assert_allclose(la, lb, rtol=2e-5)
    for pa, pb in zip(a.params(), b.params()):
        d = (pa.detach() - pb.detach()).abs().max().item()
        assert d <= 2e-4 * max(1.0, pb.detach().abs().max().item()), d  # Adam's first steps move every parameter by ~lr
    assert torch.equal(a.denom, b.denom) and torch.equal(a.max_radii2D, b.max_radii2D)
    np.testing.assert_allclose(a.xyz_gradient_accum.cpu().numpy(), b.xyz_gradient_accum.cpu().numpy(), rtol=1e-3, atol=1e-9)
    # keep training the fused model: the loss must go down substantially
    first = la[0]
    last = [_iteration(a, cam, gt, bg, True) for _ in range(40)][-1]
    assert last < 0.7 * first, (first, last)
    # prune the points that never became visible, then keep going
    never = (a.denom.squeeze(1) == 0)
    n_before = a._xyz.shape[0]
    densify.prune_points(a, never)
    assert a._xyz.shape[0] == n_before - int(never.sum()) and a.max_radii2D.shape[0] == a._xyz.shape[0]
    after = _iteration(a, cam, gt, bg, True)
    assert abs(after - last) < 0.05 * first  # invisible points carried no signal: the loss continues where it was
