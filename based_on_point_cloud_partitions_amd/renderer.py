"""`render()` of the reference (gaussian_renderer/__init__.py:18-100) on the raw-parameter operator.

The reference's `render()` builds the rasterizer's inputs from `GaussianModel`'s parameters with four activation
passes and a concat per call (scene/gaussian_model.py:102-122: `get_scaling = exp(_scaling)`,
`get_rotation = normalize(_rotation)`, `get_opacity = sigmoid(_opacity)`,
`get_features = cat(_features_dc, _features_rest)`), and autograd mirrors them in the backward.  Here the
per-Gaussian HIP kernels read the raw parameters themselves (C ABI `gsr_forward_params` / `gsr_backward_params`,
include/gsrast_mi355x.h) and hand autograd the gradients of the raw parameters directly.

`render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier, override_color)` keeps the reference's signature, its
result dict and its `pipe` switches: with `override_color`, `pipe.convert_SHs_python` or `pipe.compute_cov3D_python`
it goes through `GaussianRasterizer` exactly as the reference does; otherwise it takes the fused operator.
"""
import ctypes
import math

import torch

from . import _lib
from .rasterizer import (NUM_CHANNELS, _guess_rendered, GaussianRasterizationSettings, GaussianRasterizer, _prep, _ptr, _remember_rendered,
                         _require_hip, _stream, check_gradient_rows)


def rasterize_gaussian_params_forward(background, xyz, features_dc, features_rest, opacity, scaling, rotation, scale_modifier,
                                      viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, degree, campos, debug):
    """Forward on GaussianModel's raw parameters.  Same results tuple as `rasterize_gaussians_forward`:
    (num_rendered, out_color, radii, geomBuffer, binningBuffer, imgBuffer)."""
    if xyz.ndimension() != 2 or xyz.size(1) != 3:
        raise _lib.GsrError("means3D must have dimensions (num_points, 3)")  # rasterize_points.cu:57-59
    _require_hip(xyz, "xyz")
    lib = _lib.load()
    dev = xyz.device
    P, H, W = xyz.size(0), int(image_height), int(image_width)
    if features_dc.ndimension() != 3 or features_dc.size(0) != P or features_dc.size(1) != 1 or features_dc.size(2) != 3:
        raise _lib.GsrError("features_dc must have dimensions (num_points, 1, 3)")
    if features_rest.ndimension() != 3 or features_rest.size(0) != P or features_rest.size(2) != 3:
        raise _lib.GsrError("features_rest must have dimensions (num_points, M - 1, 3)")
    M = 1 + features_rest.size(1)
    for t, n, what in ((opacity, P, "opacity"), (scaling, 3 * P, "scaling"), (rotation, 4 * P, "rotation")):
        if t.numel() != n:
            raise _lib.GsrError(f"{what} has {t.numel()} elements, expected {n}")
    with _lib.device_guard(dev):
        alloc = torch.empty if P != 0 else torch.zeros
        out_color = alloc((NUM_CHANNELS, H, W), dtype=torch.float32, device=dev)
        radii = alloc((P,), dtype=torch.int32, device=dev)
        byte = dict(dtype=torch.uint8, device=dev)
        geomBuffer = torch.empty(0, **byte)
        binningBuffer = torch.empty(0, **byte)
        imgBuffer = torch.empty(0, **byte)
        rendered = 0
        if P != 0:
            xyz_c = _prep(xyz, dev, "xyz")
            dc_c, rest_c = _prep(features_dc, dev, "features_dc"), _prep(features_rest, dev, "features_rest")
            op_c, sc_c, rot_c = _prep(opacity, dev, "opacity"), _prep(scaling, dev, "scaling"), _prep(rotation, dev, "rotation")
            vm, pm = _prep(viewmatrix, dev, "viewmatrix"), _prep(projmatrix, dev, "projmatrix")
            cp, bg = _prep(campos, dev, "campos"), _prep(background, dev, "bg")
            geomBuffer = torch.empty(lib.gsr_geometry_bytes(P), **byte)
            imgBuffer = torch.empty(lib.gsr_image_bytes(W, H), **byte)
            key = (dev.index, P, W, H)
            guess = _guess_rendered(key)
            binningBuffer = torch.empty(lib.gsr_binning_bytes(guess + guess // 4, P, W, H) if guess > 0 else 0, **byte)
            R = ctypes.c_int64(0)
            s = _stream(dev)
            rc = _lib.check(lib.gsr_forward_params(
                P, int(degree), int(M), _ptr(bg), _ptr(xyz_c), _ptr(dc_c), _ptr(rest_c), _ptr(op_c), _ptr(sc_c), _ptr(rot_c),
                float(scale_modifier), _ptr(vm), _ptr(pm), _ptr(cp), W, H, float(tan_fovx), float(tan_fovy), _ptr(radii),
                _ptr(geomBuffer), geomBuffer.numel(), _ptr(binningBuffer), binningBuffer.numel(), _ptr(imgBuffer), imgBuffer.numel(),
                _ptr(out_color), ctypes.byref(R), s), "gsr_forward_params")
            rendered = int(R.value)
            if rc == 1:  # estimate too small (or first call): phase 2 with an exactly sized buffer
                binningBuffer = torch.empty(lib.gsr_binning_bytes(rendered, P, W, H), **byte)
                _lib.check(lib.gsr_forward_render(
                    P, rendered, W, H, _ptr(bg), _ptr(geomBuffer), geomBuffer.numel(), _ptr(binningBuffer), binningBuffer.numel(),
                    _ptr(imgBuffer), imgBuffer.numel(), _ptr(out_color), s), "gsr_forward_render")
            _remember_rendered(key, rendered)
            if debug:
                torch.cuda.synchronize(dev)
    return rendered, out_color, radii, geomBuffer, binningBuffer, imgBuffer


def rasterize_gaussian_params_backward(background, xyz, radii, features_dc, features_rest, scaling, rotation, scale_modifier,
                                       viewmatrix, projmatrix, tan_fovx, tan_fovy, dL_dout_color, degree, campos, geomBuffer, R,
                                       binningBuffer, imageBuffer, debug):
    """Backward on the raw parameters.  Returns (dL_dmeans2D, dL_dxyz, dL_dfeatures_dc, dL_dfeatures_rest, dL_dopacity,
    dL_dscaling, dL_drotation), all w.r.t. the PRE-activation parameters."""
    _require_hip(xyz, "xyz")
    lib = _lib.load()
    dev = xyz.device
    P = xyz.size(0)
    H, W = dL_dout_color.size(1), dL_dout_color.size(2)
    Mrest = features_rest.size(1)
    f = dict(dtype=torch.float32, device=dev)
    with _lib.device_guard(dev):
        if P == 0:
            z = lambda *shape: torch.zeros(shape, **f)
            return z(P, 3), z(P, 3), z(P, 1, 3), z(P, Mrest, 3), z(P, 1), z(P, 3), z(P, 4)
        dL_dmeans2D = torch.empty((P, 3), **f)
        dL_dxyz = torch.empty((P, 3), **f)
        dL_ddc = torch.empty((P, 1, 3), **f)
        dL_drest = torch.empty((P, Mrest, 3), **f)
        dL_dopacity = torch.empty((P, 1), **f)
        dL_dscaling = torch.empty((P, 3), **f)
        dL_drotation = torch.empty((P, 4), **f)
        scratch = torch.empty(lib.gsr_backward_scratch_bytes(int(R), W, H) if R > 0 else 0, dtype=torch.uint8, device=dev)
        xyz_c = _prep(xyz, dev, "xyz")
        dc_c, rest_c = _prep(features_dc, dev, "features_dc"), _prep(features_rest, dev, "features_rest")
        sc_c, rot_c = _prep(scaling, dev, "scaling"), _prep(rotation, dev, "rotation")
        vm, pm = _prep(viewmatrix, dev, "viewmatrix"), _prep(projmatrix, dev, "projmatrix")
        cp, bg = _prep(campos, dev, "campos"), _prep(background, dev, "bg")
        grad = _prep(dL_dout_color, dev, "dL_dout_color")
        _lib.check(lib.gsr_backward_params(
            P, int(degree), 1 + Mrest, int(R), W, H, _ptr(bg), _ptr(xyz_c), _ptr(dc_c), _ptr(rest_c), _ptr(sc_c), _ptr(rot_c),
            float(scale_modifier), _ptr(vm), _ptr(pm), _ptr(cp), float(tan_fovx), float(tan_fovy), _ptr(radii.contiguous()),
            _ptr(geomBuffer), geomBuffer.numel(), _ptr(binningBuffer), binningBuffer.numel(), _ptr(imageBuffer), imageBuffer.numel(),
            _ptr(scratch), scratch.numel(), _ptr(grad), _ptr(dL_dmeans2D), _ptr(dL_dxyz), _ptr(dL_ddc), _ptr(dL_drest),
            _ptr(dL_dopacity), _ptr(dL_dscaling), _ptr(dL_drotation), _stream(dev)), "gsr_backward_params")
        if debug:
            torch.cuda.synchronize(dev)
    return dL_dmeans2D, dL_dxyz, dL_ddc, dL_drest, dL_dopacity, dL_dscaling, dL_drotation


class _RasterizeGaussianParams(torch.autograd.Function):
    """autograd node with `_RasterizeGaussians`' protocol (diff_gaussian_rasterization/__init__.py:44-155) on raw parameters."""

    @staticmethod
    def forward(ctx, xyz, means2D, features_dc, features_rest, opacity, scaling, rotation, raster_settings):
        rs = raster_settings
        num_rendered, color, radii, geomBuffer, binningBuffer, imgBuffer = rasterize_gaussian_params_forward(
            rs.bg, xyz, features_dc, features_rest, opacity, scaling, rotation, rs.scale_modifier, rs.viewmatrix, rs.projmatrix,
            rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, rs.sh_degree, rs.campos, rs.debug)
        if any(ctx.needs_input_grad):
            check_gradient_rows(xyz.size(0), num_rendered, geomBuffer, xyz.device)
        ctx.raster_settings = rs
        ctx.num_rendered = num_rendered
        ctx.save_for_backward(xyz, features_dc, features_rest, scaling, rotation, radii, geomBuffer, binningBuffer, imgBuffer)
        ctx.mark_non_differentiable(radii)
        return color, radii

    @staticmethod
    def backward(ctx, grad_out_color, _):
        rs = ctx.raster_settings
        xyz, features_dc, features_rest, scaling, rotation, radii, geomBuffer, binningBuffer, imgBuffer = ctx.saved_tensors
        g2d, gxyz, gdc, grest, gop, gsc, grot = rasterize_gaussian_params_backward(
            rs.bg, xyz, radii, features_dc, features_rest, scaling, rotation, rs.scale_modifier, rs.viewmatrix, rs.projmatrix,
            rs.tanfovx, rs.tanfovy, grad_out_color, rs.sh_degree, rs.campos, geomBuffer, ctx.num_rendered, binningBuffer,
            imgBuffer, rs.debug)
        return gxyz, g2d, gdc, grest, gop, gsc, grot, None


def rasterize_gaussian_params(xyz, means2D, features_dc, features_rest, opacity, scaling, rotation, raster_settings):
    """(color [3,H,W], radii [P]) from GaussianModel's raw parameters `_xyz`, `_features_dc` [P,1,3], `_features_rest`
    [P,M-1,3], `_opacity` [P,1], `_scaling` [P,3], `_rotation` [P,4]; `means2D` only receives the screen-space gradient."""
    return _RasterizeGaussianParams.apply(xyz, means2D, features_dc, features_rest, opacity, scaling, rotation, raster_settings)


def render(viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None, *, fused=True):
    """Drop-in for gaussian_renderer/__init__.py:18-100.  `pc` needs the attributes of `GaussianModel` that function
    touches (`get_xyz`, `active_sh_degree`, and either the raw `_features_dc/_features_rest/_opacity/_scaling/_rotation`
    or the getters); background tensor must be on the GPU.  `fused=False` forces the reference's data flow (getters +
    GaussianRasterizer)."""
    screenspace_points = torch.zeros_like(pc.get_xyz, dtype=pc.get_xyz.dtype, requires_grad=True, device=pc.get_xyz.device) + 0
    try:
        screenspace_points.retain_grad()
    except Exception:
        pass
    raster_settings = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width),
        tanfovx=math.tan(viewpoint_camera.FoVx * 0.5), tanfovy=math.tan(viewpoint_camera.FoVy * 0.5), bg=bg_color,
        scale_modifier=scaling_modifier, viewmatrix=viewpoint_camera.world_view_transform,
        projmatrix=viewpoint_camera.full_proj_transform, sh_degree=pc.active_sh_degree, campos=viewpoint_camera.camera_center,
        prefiltered=False, debug=getattr(pipe, "debug", False))
    fused = (fused and override_color is None and not getattr(pipe, "convert_SHs_python", False)
             and not getattr(pipe, "compute_cov3D_python", False) and hasattr(pc, "_features_dc"))
    if fused:
        rendered_image, radii = rasterize_gaussian_params(pc.get_xyz, screenspace_points, pc._features_dc, pc._features_rest,
                                                          pc._opacity, pc._scaling, pc._rotation, raster_settings)
    else:
        rasterizer = GaussianRasterizer(raster_settings=raster_settings)
        scales = rotations = cov3D_precomp = None
        if getattr(pipe, "compute_cov3D_python", False):
            cov3D_precomp = pc.get_covariance(scaling_modifier)
        else:
            scales, rotations = pc.get_scaling, pc.get_rotation
        shs = colors_precomp = None
        if override_color is None:
            if getattr(pipe, "convert_SHs_python", False):
                raise _lib.GsrError("pipe.convert_SHs_python needs utils/sh_utils.eval_sh of the reference; pass the result as "
                                    "override_color")
            shs = pc.get_features
        else:
            colors_precomp = override_color
        rendered_image, radii = rasterizer(means3D=pc.get_xyz, means2D=screenspace_points, shs=shs, colors_precomp=colors_precomp,
                                           opacities=pc.get_opacity, scales=scales, rotations=rotations, cov3D_precomp=cov3D_precomp)
    return {"render": rendered_image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0, "radii": radii}
