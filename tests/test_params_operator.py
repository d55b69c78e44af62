"""Raw-parameter operator: exp / sigmoid / normalize / cat of scene/gaussian_model.py:102-122 folded
into the per-Gaussian kernels (C ABI gsr_forward_params / gsr_backward_params, host side renderer.py).

CPU part: the oracle's restatement of the activations (oracle/gs_oracle.c, orc_activate*) is pinned against torch itself
(the reference calls torch.exp / torch.sigmoid / F.normalize / torch.cat -- importable here) including autograd.
GPU part: the fused HIP path vs oracle(activate -> forward -> backward -> activate_backward): forward bit-exact,
gradients within 1e-4 of each tensor's scale; and vs the unfused path (torch activations + GaussianRasterizer).
"""
import numpy as np
import pytest
import torch

from helpers import GRAD_TOL, assert_chain_close, unpack_sums, assert_forward_bit_exact, hip_settings, oracle_forward, scene, two_stage_check
from oracle import oracle


def raw_params(g, seed=0, m=None):
    """Pre-activation parameters whose activations are (close to) the synthetic Gaussians `g` (as GaussianModel stores them)."""
    gen = torch.Generator().manual_seed(seed + 77)
    P = g.xyz.shape[0]
    m = g.shs.shape[1] if m is None else m
    return dict(xyz=g.xyz.clone(), features_dc=g.shs[:, :1].contiguous(), features_rest=g.shs[:, 1:m].contiguous(),
                opacity=torch.logit(g.opacities.clamp(1e-4, 1 - 1e-4)), scaling=torch.log(g.scales),
                rotation=(g.rotations * (0.25 + 3.0 * torch.rand(P, 1, generator=gen))).contiguous())


def activated(raw):
    from based_on_point_cloud_partitions_amd import synthetic as syn

    a = oracle.activate(raw["features_dc"].numpy(), raw["features_rest"].numpy(), raw["opacity"].numpy(), raw["scaling"].numpy(),
                        raw["rotation"].numpy())
    return syn.SyntheticGaussians(raw["xyz"], torch.tensor(a["scales"]), torch.tensor(a["rotations"]), torch.tensor(a["opacities"]),
                                  torch.tensor(a["shs"]))


# ------------------------------------------------------------------------------------------------------------
# CPU: oracle activations pinned by torch
# ------------------------------------------------------------------------------------------------------------
def test_oracle_activations_match_torch_forward_and_autograd():
    gen = torch.Generator().manual_seed(5)
    P, Mrest = 4000, 15
    raw = dict(features_dc=torch.randn(P, 1, 3, generator=gen), features_rest=0.1 * torch.randn(P, Mrest, 3, generator=gen),
               opacity=3.0 * torch.randn(P, 1, generator=gen), scaling=-3.0 + 2.0 * torch.randn(P, 3, generator=gen),
               rotation=torch.randn(P, 4, generator=gen) * torch.exp(2 * torch.randn(P, 1, generator=gen)))
    raw["rotation"][7] = 0.0  # below the 1e-12 clamp of F.normalize
    raw["opacity"][3], raw["opacity"][4] = 30.0, -30.0
    leaves = {k: v.clone().requires_grad_(True) for k, v in raw.items()}
    # scene/gaussian_model.py:102-122
    shs = torch.cat((leaves["features_dc"], leaves["features_rest"]), dim=1)
    opac, scales = torch.sigmoid(leaves["opacity"]), torch.exp(leaves["scaling"])
    rots = torch.nn.functional.normalize(leaves["rotation"])
    a = oracle.activate(*(raw[k].numpy() for k in ("features_dc", "features_rest", "opacity", "scaling", "rotation")))
    assert np.array_equal(a["shs"], shs.detach().numpy())
    for got, want in ((a["opacities"], opac), (a["scales"], scales), (a["rotations"], rots)):
        np.testing.assert_allclose(got, want.detach().numpy(), rtol=5e-7, atol=1e-30)
    up = dict(dL_dsh=torch.randn(P, 1 + Mrest, 3, generator=gen), dL_dopacity=torch.randn(P, 1, generator=gen),
              dL_dscales=torch.randn(P, 3, generator=gen), dL_drotations=torch.randn(P, 4, generator=gen))
    ((shs * up["dL_dsh"]).sum() + (opac * up["dL_dopacity"]).sum() + (scales * up["dL_dscales"]).sum()
     + (rots * up["dL_drotations"]).sum()).backward()
    g = {k: v.numpy() for k, v in up.items()}
    g["dL_dmeans3D"] = g["dL_dmeans2D"] = np.zeros((P, 3), np.float32)
    b = oracle.activate_backward(raw["opacity"].numpy(), raw["scaling"].numpy(), raw["rotation"].numpy(), g)
    assert np.array_equal(b["dL_dfeatures_dc"], leaves["features_dc"].grad.numpy())
    assert np.array_equal(b["dL_dfeatures_rest"], leaves["features_rest"].grad.numpy())
    np.testing.assert_allclose(b["dL_dopacity_raw"], leaves["opacity"].grad.numpy(), rtol=2e-6, atol=1e-12)
    np.testing.assert_allclose(b["dL_dscaling_raw"], leaves["scaling"].grad.numpy(), rtol=2e-6, atol=1e-12)
    want = leaves["rotation"].grad.numpy()
    n = raw["rotation"].norm(dim=1).numpy()
    tol = 2e-5 * np.abs(up["dL_drotations"].numpy()).max() / np.maximum(n, 1e-12)[:, None]  # (g - q q.g)/n cancels for g ~ q
    assert np.all(np.abs(b["dL_drotation_raw"] - want) <= tol + 1e-30)
    assert np.all(b["dL_drotation_raw"][7] == up["dL_drotations"][7].numpy() / np.float32(1e-12))


def test_activate_degree0_model_has_no_rest_coefficients():
    raw = dict(features_dc=torch.randn(10, 1, 3), features_rest=torch.zeros(10, 0, 3), opacity=torch.randn(10, 1),
               scaling=torch.randn(10, 3), rotation=torch.randn(10, 4))
    a = oracle.activate(*(raw[k].numpy() for k in ("features_dc", "features_rest", "opacity", "scaling", "rotation")))
    assert a["shs"].shape == (10, 1, 3) and np.array_equal(a["shs"], raw["features_dc"].numpy())


# ------------------------------------------------------------------------------------------------------------
# GPU: the fused HIP operator
# ------------------------------------------------------------------------------------------------------------
def hip_params_forward(raw, cam, bg, deg, scale_modifier=1.0, device="cuda"):
    from based_on_point_cloud_partitions_amd import rasterizer as rz, renderer as rn

    st = hip_settings(cam, bg, deg, scale_modifier, device)
    d = {k: v.to(device) for k, v in raw.items()}
    R, color, radii, gb, bb, ib = rn.rasterize_gaussian_params_forward(
        st.bg, d["xyz"], d["features_dc"], d["features_rest"], d["opacity"], d["scaling"], d["rotation"], scale_modifier,
        st.viewmatrix, st.projmatrix, st.tanfovx, st.tanfovy, st.image_height, st.image_width, deg, st.campos, False)
    ex = rz.debug_export(d["xyz"].shape[0], R, cam.image_width, cam.image_height, gb, bb, ib)
    torch.cuda.synchronize()
    return dict(st=st, d=d, R=R, color=color, radii=radii, gb=gb, bb=bb, ib=ib, ex=ex, deg=deg, scale_modifier=scale_modifier)


def hip_params_backward(ctx, dL, capture=True):
    """gsr_backward_params; with capture also "sums": the blend sums the kernel's chain consumed (two-stage check)."""
    from based_on_point_cloud_partitions_amd import _lib
    from based_on_point_cloud_partitions_amd import renderer as rn

    st, d = ctx["st"], ctx["d"]
    sums = torch.zeros(d["xyz"].shape[0], 12, device=d["xyz"].device) if capture else None
    _lib.capture_sums(sums)
    try:
        out = rn.rasterize_gaussian_params_backward(
            st.bg, d["xyz"], ctx["radii"], d["features_dc"], d["features_rest"], d["scaling"], d["rotation"], ctx["scale_modifier"],
            st.viewmatrix, st.projmatrix, st.tanfovx, st.tanfovy, torch.as_tensor(dL).to(d["xyz"].device), ctx["deg"], st.campos,
            ctx["gb"], ctx["R"], ctx["bb"], ctx["ib"], False)
        torch.cuda.synchronize()
    finally:
        _lib.capture_sums(None)
    names = ["dL_dmeans2D", "dL_dxyz", "dL_dfeatures_dc", "dL_dfeatures_rest", "dL_dopacity_raw", "dL_dscaling_raw", "dL_drotation_raw"]
    res = {k: v.cpu().numpy() for k, v in zip(names, out)}
    if capture:
        res["sums"] = sums.cpu().numpy()
    return res


def raw_chain(raw, st):
    """sums -> expected raw-parameter gradients: the oracle's per-Gaussian chain on those sums, then the chain rule of the
    activations (scene/gaussian_model.py:102-122)."""
    return lambda hs: oracle.activate_backward(raw["opacity"].numpy(), raw["scaling"].numpy(), raw["rotation"].numpy(),
                                               oracle.chain_backward(st, hs))


def _params_parity(P, W, H, focal, seed=0, deg=3, m=None, bg=(0.0, 0.0, 0.0), scale_modifier=1.0, **kw):
    g, cam = scene(P, W, H, focal, seed=seed, **kw)
    raw = raw_params(g, seed, m)
    ga = activated(raw)
    st = oracle_forward(ga, cam, bg=bg, sh_degree=deg, scale_modifier=scale_modifier)
    ctx = hip_params_forward(raw, cam, bg, deg, scale_modifier)
    ex = assert_forward_bit_exact(ctx, st)
    vis = st["radii"] > 0
    assert np.array_equal(ex["clamped"][vis], st["clamped"][vis])
    assert np.array_equal(ex["cov3D"].view(np.uint32)[vis], st["cov3D"].view(np.uint32)[vis])
    dL = np.random.default_rng(seed + 100).standard_normal((3, H, W)).astype(np.float32)
    got = hip_params_backward(ctx, dL)
    assert got["dL_dfeatures_rest"].shape == (raw["features_rest"].shape[0], raw["features_rest"].shape[1], 3)
    two_stage_check(st, got, dL, chain_of=raw_chain(raw, st))  # blend sums vs the oracle's, raw gradients vs the chain on the HIP sums
    got.pop("sums")
    return st, ctx, got


@pytest.mark.gpu
@pytest.mark.parametrize("deg,m", [(0, 1), (0, 16), (1, 4), (2, 9), (3, 16)])
def test_params_operator_small_scene_all_sh_layouts(deg, m):
    _params_parity(3001, 160, 96, 100.0, seed=deg, deg=deg, m=m, bg=(0.2, 0.1, 0.3), scale_lo=0.01, scale_hi=0.12)


@pytest.mark.gpu
def test_params_operator_config1_and_scale_modifier():
    _params_parity(10_000, 256, 256, 160.0, seed=0)
    _params_parity(4000, 128, 128, 90.0, seed=2, scale_modifier=1.7, scale_lo=0.01, scale_hi=0.08)


@pytest.mark.gpu
def test_params_operator_config2_500k_1080p_full_parity():
    st, ctx, got = _params_parity(500_000, 1920, 1080, 1200.0, seed=0, n_cams=16, cam_index=0)
    assert st["num_rendered"] > 3_000_000
    culled = st["radii"] == 0
    assert culled.any() and all(np.all(v[culled] == 0) for v in got.values())  # every element written, zeros when culled


@pytest.mark.gpu
def test_params_operator_zero_norm_quaternion_and_empty_model():
    g, cam = scene(500, 64, 64, 60.0, seed=9, scale_lo=0.02, scale_hi=0.1)
    raw = raw_params(g, 9)
    raw["rotation"][:50] = 0.0  # normalize clamps the norm at 1e-12: q = 0, R = I
    st = oracle_forward(activated(raw), cam)
    ctx = hip_params_forward(raw, cam, (0.0, 0.0, 0.0), 3)
    assert_forward_bit_exact(ctx, st)
    dL = np.ones((3, 64, 64), np.float32)
    got = hip_params_backward(ctx, dL)
    want = raw_chain(raw, st)(unpack_sums(got["sums"]))
    fin = np.isfinite(want["dL_drotation_raw"]).all(axis=1)
    assert np.array_equal(np.isfinite(got["dL_drotation_raw"]).all(axis=1), fin)
    rot_got, rot_want = got.pop("dL_drotation_raw"), want["dL_drotation_raw"]
    two_stage_check(st, got, dL, chain_of=lambda hs: {k: v for k, v in raw_chain(raw, st)(hs).items() if k != "dL_drotation_raw"})
    # gradients of the clamped quaternions are g / 1e-12: compare in relative terms
    a, b = rot_got[:50].astype(np.float64), rot_want[:50].astype(np.float64)
    ok = np.isfinite(b)
    assert np.all(np.abs(a[ok] - b[ok]) <= 1e-4 * max(1.0, np.abs(b[ok]).max()))
    assert_chain_close({"r": rot_got[50:]}, {"r": rot_want[50:]})
    raw0 = {k: v[:0] for k, v in raw.items()}
    ctx0 = hip_params_forward(raw0, cam, (0.1, 0.2, 0.3), 3)
    assert ctx0["R"] == 0 and torch.all(ctx0["color"] == 0)
    got0 = hip_params_backward(ctx0, dL, capture=False)
    assert got0["dL_dfeatures_rest"].shape == (0, 15, 3) and got0["dL_dopacity_raw"].shape == (0, 1)


class _Model:
    """The attributes of GaussianModel that render() touches (scene/gaussian_model.py:48-56,102-122)."""

    def __init__(self, raw, deg, device):
        self.active_sh_degree = deg
        self._xyz = raw["xyz"].to(device).requires_grad_(True)
        self._features_dc = raw["features_dc"].to(device).requires_grad_(True)
        self._features_rest = raw["features_rest"].to(device).requires_grad_(True)
        self._opacity = raw["opacity"].to(device).requires_grad_(True)
        self._scaling = raw["scaling"].to(device).requires_grad_(True)
        self._rotation = raw["rotation"].to(device).requires_grad_(True)

    get_xyz = property(lambda s: s._xyz)
    get_scaling = property(lambda s: torch.exp(s._scaling))
    get_rotation = property(lambda s: torch.nn.functional.normalize(s._rotation))
    get_opacity = property(lambda s: torch.sigmoid(s._opacity))
    get_features = property(lambda s: torch.cat((s._features_dc, s._features_rest), dim=1))

    def params(self):
        return [self._xyz, self._features_dc, self._features_rest, self._opacity, self._scaling, self._rotation]


class _Cam:
    def __init__(self, cam, device):
        import math
        self.image_height, self.image_width = cam.image_height, cam.image_width
        self.FoVx, self.FoVy = 2 * math.atan(cam.tanfovx), 2 * math.atan(cam.tanfovy)
        self.world_view_transform = cam.world_view_transform.to(device)
        self.full_proj_transform = cam.full_proj_transform.to(device)
        self.camera_center = cam.camera_center.to(device)


class _Pipe:
    convert_SHs_python = False
    compute_cov3D_python = False
    debug = False


@pytest.mark.gpu
def test_render_fused_matches_unfused_autograd_path():
    """renderer.render (fused) vs the same function forced through GaussianRasterizer + torch activations, which is the
    reference's data flow (gaussian_renderer/__init__.py:52-93)."""
    from based_on_point_cloud_partitions_amd import renderer as rn

    P, W, H = 6000, 200, 120
    g, cam = scene(P, W, H, 110.0, seed=8, scale_lo=0.01, scale_hi=0.12)
    raw = raw_params(g, 8)
    dev = "cuda"
    bg = torch.tensor([0.05, 0.1, 0.15], device=dev)
    dL = torch.randn(3, H, W, generator=torch.Generator().manual_seed(3)).to(dev)
    res = {}
    for name in ("fused", "unfused"):
        pc = _Model(raw, 3, dev)
        out = rn.render(_Cam(cam, dev), pc, _Pipe(), bg, 1.0, fused=(name == "fused"))
        assert set(out) == {"render", "viewspace_points", "visibility_filter", "radii"}
        (out["render"] * dL).sum().backward()
        ps = pc.params()
        res[name] = dict(img=out["render"].detach().cpu().numpy(), radii=out["radii"].cpu().numpy(),
                         vsp=out["viewspace_points"].grad.cpu().numpy(), grads=[p.grad.cpu().numpy() for p in ps],
                         vis=out["visibility_filter"].cpu().numpy())
    a, b = res["fused"], res["unfused"]
    # torch's exp / sigmoid on the device and gs_expf differ in the last bit, which may flip an alpha >= 1/255 test: one
    # flip moves a pixel by at most 1/255 of a colour; everything else agrees to rounding
    assert np.array_equal(a["radii"], b["radii"]) and np.array_equal(a["vis"], b["vis"])
    d = np.abs(a["img"] - b["img"])
    assert d.max() <= 5e-3 and d.mean() <= 1e-6
    assert np.abs(a["vsp"] - b["vsp"]).max() <= 10 * GRAD_TOL * max(1.0, np.abs(b["vsp"]).max())
    for x, y in zip(a["grads"], b["grads"]):
        assert x.shape == y.shape and np.abs(x - y).max() <= 10 * GRAD_TOL * max(1.0, np.abs(y).max())


@pytest.mark.gpu
def test_render_override_color_takes_the_reference_path():
    from based_on_point_cloud_partitions_amd import renderer as rn

    g, cam = scene(2000, 96, 96, 80.0, seed=4, scale_lo=0.02, scale_hi=0.1)
    raw = raw_params(g, 4)
    pc = _Model(raw, 3, "cuda")
    col = torch.rand(2000, 3, device="cuda")
    out = rn.render(_Cam(cam, "cuda"), pc, _Pipe(), torch.zeros(3, device="cuda"), 1.0, override_color=col)
    # this path activates with torch on the device (the reference's data flow): feed the oracle those very values
    from based_on_point_cloud_partitions_amd import synthetic as syn
    c = lambda t: t.detach().cpu()
    ga = syn.SyntheticGaussians(c(pc.get_xyz), c(pc.get_scaling), c(pc.get_rotation), c(pc.get_opacity), c(pc.get_features))
    st = oracle_forward(ga, cam, colors_precomp=col.cpu().numpy())
    assert np.array_equal(out["radii"].cpu().numpy(), st["radii"])
    assert np.array_equal(out["render"].detach().cpu().numpy(), st["out_color"])
