"""Host-side mirror of the reference's operator surface for the rasterizer, on top of the C ABI.

Names, argument order/meaning and error behaviour follow
submodules/diff-gaussian-rasterization/diff_gaussian_rasterization/__init__.py
(`rasterize_gaussians` :21-42, `_RasterizeGaussians` :44-155, `GaussianRasterizationSettings` :157-169,
`GaussianRasterizer` :171-220) and the pybind functions it calls (rasterize_points.cu:35-217), so that
gaussian_renderer/__init__.py:14,36-93 works unchanged when `dropin/` is on PYTHONPATH.
"""
import ctypes
import threading
from typing import NamedTuple

import torch
import torch.nn as nn

from . import _lib

NUM_CHANNELS = 3  # cuda_rasterizer/config.h:14

# Largest num_rendered seen per (device, P, W, H): sizes the binning buffer BEFORE the forward so that both phases run
# in one native call (gsr_forward); a too-small guess only costs a second call.  Densification changes P every few
# hundred iterations, so the table is kept small (oldest keys dropped).
_R_SEEN = {}
_R_SEEN_MAX = 64
_R_SEEN_LOCK = threading.Lock()  # autograd runs backward (and users may run forwards) from several host threads


def _remember_rendered(key, rendered):
    with _R_SEEN_LOCK:
        if rendered > _R_SEEN.get(key, 0):
            _R_SEEN.pop(key, None)
            _R_SEEN[key] = rendered  # most recently raised key last
            while len(_R_SEEN) > _R_SEEN_MAX:
                _R_SEEN.pop(next(iter(_R_SEEN)))


def _guess_rendered(key):
    """Largest num_rendered seen for (device, P, W, H); for a P not seen yet (the model after a densification step: a new P every
    hundred iterations, train.py:145-150) the most recent entry of the same device and image size whose P is within a factor of two,
    scaled by the ratio of the Gaussian counts -- so the first frames behind a densification keep the one-call forward instead of re-allocating their way up."""
    with _R_SEEN_LOCK:
        r = _R_SEEN.get(key)
        if r is not None:
            return r
        dev, P, W, H = key
        for (d2, P2, W2, H2), r2 in reversed(_R_SEEN.items()):
            if d2 == dev and W2 == W and H2 == H and P2 > 0 and 2 * P2 >= P >= P2 // 2:  # (a densification step, not another scene)
                return int(r2 * (P / P2)) + 1
        return 0


def _ptr(t):
    """Device pointer or NULL for an absent (empty) optional tensor (rasterize_points.cu:94-111 convention)."""
    if t is None or t.numel() == 0:
        return None
    return ctypes.c_void_p(t.data_ptr())


def _prep(t, device, what):
    """float32 + contiguous, as the reference's .contiguous().data<float>() does; empty tensors stay empty."""
    if t is None or t.numel() == 0:
        return None
    if t.device != device:
        raise _lib.GsrError(f"{what} is on {t.device}, expected {device} (the HIP rasterizer has no CPU fallback)")
    if t.dtype != torch.float32:
        raise _lib.GsrError(f"{what} must be float32, got {t.dtype}")
    return t.contiguous()


def _stream(device):
    return _lib.stream_ptr(device)


def _require_hip(t, what):
    if not t.is_cuda:
        raise _lib.GsrError(f"{what} must live on a HIP device (got {t.device}); there is no CPU fallback")


def rasterize_gaussians_forward(background, means3D, colors, opacity, scales, rotations, scale_modifier, cov3D_precomp,
                                viewmatrix, projmatrix, tan_fovx, tan_fovy, image_height, image_width, sh, degree, campos,
                                prefiltered, debug):
    """Counterpart of `_C.rasterize_gaussians` (RasterizeGaussiansCUDA, rasterize_points.cu:35-115).
    Returns (num_rendered, out_color, radii, geomBuffer, binningBuffer, imgBuffer)."""
    if means3D.ndimension() != 2 or means3D.size(1) != 3:
        raise _lib.GsrError("means3D must have dimensions (num_points, 3)")  # rasterize_points.cu:57-59
    _require_hip(means3D, "means3D")
    lib = _lib.load()
    dev = means3D.device
    P, H, W = means3D.size(0), int(image_height), int(image_width)
    with _lib.device_guard(dev):
        # P > 0: every pixel and every radius is written by the kernels; P == 0: nothing runs and both stay zero
        # (torch::full(0.0) in the reference, rasterize_points.cu:68-69)
        alloc = torch.empty if P != 0 else torch.zeros
        out_color = alloc((NUM_CHANNELS, H, W), dtype=torch.float32, device=dev)
        radii = alloc((P,), dtype=torch.int32, device=dev)
        byte = dict(dtype=torch.uint8, device=dev)
        geomBuffer = torch.empty(0, **byte)
        binningBuffer = torch.empty(0, **byte)
        imgBuffer = torch.empty(0, **byte)
        rendered = 0
        if P != 0:
            M = sh.size(1) if (sh is not None and sh.numel() != 0) else 0
            means3D_c = _prep(means3D, dev, "means3D")
            sh_c, colors_c = _prep(sh, dev, "sh"), _prep(colors, dev, "colors_precomp")
            opacity_c = _prep(opacity, dev, "opacities")
            scales_c, rot_c, cov_c = _prep(scales, dev, "scales"), _prep(rotations, dev, "rotations"), _prep(cov3D_precomp, dev, "cov3D_precomp")
            vm, pm = _prep(viewmatrix, dev, "viewmatrix"), _prep(projmatrix, dev, "projmatrix")
            cp, bg = _prep(campos, dev, "campos"), _prep(background, dev, "bg")
            if opacity_c is None or opacity_c.numel() != P:
                raise _lib.GsrError("opacities must have one value per Gaussian")
            geomBuffer = torch.empty(lib.gsr_geometry_bytes(P), **byte)
            imgBuffer = torch.empty(lib.gsr_image_bytes(W, H), **byte)
            key = (dev.index, P, W, H)
            guess = _guess_rendered(key)
            binningBuffer = torch.empty(lib.gsr_binning_bytes(guess + guess // 4, P, W, H) if guess > 0 else 0, **byte)
            R = ctypes.c_int64(0)
            s = _stream(dev)
            rc = _lib.check(lib.gsr_forward(
                P, int(degree), int(M), _ptr(bg), _ptr(means3D_c), _ptr(scales_c), float(scale_modifier), _ptr(rot_c),
                _ptr(opacity_c), _ptr(sh_c), _ptr(cov_c), _ptr(colors_c), _ptr(vm), _ptr(pm), _ptr(cp), W, H, float(tan_fovx),
                float(tan_fovy), int(bool(prefiltered)), _ptr(radii), _ptr(geomBuffer), geomBuffer.numel(), _ptr(binningBuffer),
                binningBuffer.numel(), _ptr(imgBuffer), imgBuffer.numel(), _ptr(out_color), ctypes.byref(R), s), "gsr_forward")
            rendered = int(R.value)
            if rc == 1:  # estimate too small (or first call): phase 2 with an exactly sized buffer
                binningBuffer = torch.empty(lib.gsr_binning_bytes(rendered, P, W, H), **byte)
                _lib.check(lib.gsr_forward_render(
                    P, rendered, W, H, _ptr(bg), _ptr(geomBuffer), geomBuffer.numel(), _ptr(binningBuffer), binningBuffer.numel(),
                    _ptr(imgBuffer), imgBuffer.numel(), _ptr(out_color), s), "gsr_forward_render")
            _remember_rendered(key, rendered)
            if debug:
                torch.cuda.synchronize(dev)  # CHECK_CUDA(..., debug), auxiliary.h:166-173
    return rendered, out_color, radii, geomBuffer, binningBuffer, imgBuffer


def check_gradient_rows(P, rendered, geomBuffer, dev):
    """The blend backward numbers its gradient rows in 28 bits (include/gsrast_mi355x.h, gsr_gradient_row_limit).  Called by the
    autograd Functions at FORWARD time when a gradient will be asked for, so that an over-sized frame fails where it was rendered and not
    inside `loss.backward()`.  Rows <= instances: nothing is read from the device unless num_rendered itself reaches the limit."""
    lib = _lib.load()
    limit = int(lib.gsr_gradient_row_limit())
    if rendered < limit:
        return
    rows = ctypes.c_int64(0)
    with _lib.device_guard(dev):
        _lib.check(lib.gsr_gradient_rows(int(P), _ptr(geomBuffer), geomBuffer.numel(), ctypes.byref(rows), _stream(dev)), "gsr_gradient_rows")
    if rows.value >= limit:
        raise _lib.GsrError(f"{rows.value} gradient rows ({rendered} Gaussian-tile instances): the backward of this rasterizer numbers its "
                            f"gradient rows in 28 bits (limit {limit}); render without gradients, or split the scene / lower the resolution")


def rasterize_gaussians_backward(background, means3D, radii, colors, scales, rotations, scale_modifier, cov3D_precomp, viewmatrix,
                                 projmatrix, tan_fovx, tan_fovy, dL_dout_color, sh, degree, campos, geomBuffer, R, binningBuffer,
                                 imageBuffer, debug):
    """Counterpart of `_C.rasterize_gaussians_backward` (RasterizeGaussiansBackwardCUDA, rasterize_points.cu:117-196).
    Returns (dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations)."""
    _require_hip(means3D, "means3D")
    lib = _lib.load()
    dev = means3D.device
    P = means3D.size(0)
    H, W = dL_dout_color.size(1), dL_dout_color.size(2)
    M = sh.size(1) if (sh is not None and sh.numel() != 0) else 0
    f = dict(dtype=torch.float32, device=dev)
    with _lib.device_guard(dev):
        if P == 0:
            z = lambda *shape: torch.zeros(shape, **f)
            return z(P, 3), z(P, NUM_CHANNELS), z(P, 1), z(P, 3), z(P, 6), z(P, M, 3), z(P, 3), z(P, 4)
        # every element is written by the kernels -> torch.empty instead of the reference's torch::zeros (:151-159)
        dL_dmeans3D = torch.empty((P, 3), **f)
        dL_dmeans2D = torch.empty((P, 3), **f)
        dL_dcolors = torch.empty((P, NUM_CHANNELS), **f)
        dL_dopacity = torch.empty((P, 1), **f)
        dL_dcov3D = torch.empty((P, 6), **f)
        dL_dsh = torch.empty((P, M, 3), **f)
        dL_dscales = torch.empty((P, 3), **f)
        dL_drotations = torch.empty((P, 4), **f)
        scratch = torch.empty(lib.gsr_backward_scratch_bytes(int(R), W, H) if R > 0 else 0, dtype=torch.uint8, device=dev)
        means3D_c = _prep(means3D, dev, "means3D")
        sh_c, colors_c = _prep(sh, dev, "sh"), _prep(colors, dev, "colors_precomp")
        scales_c, rot_c, cov_c = _prep(scales, dev, "scales"), _prep(rotations, dev, "rotations"), _prep(cov3D_precomp, dev, "cov3D_precomp")
        vm, pm = _prep(viewmatrix, dev, "viewmatrix"), _prep(projmatrix, dev, "projmatrix")
        cp, bg = _prep(campos, dev, "campos"), _prep(background, dev, "bg")
        grad = _prep(dL_dout_color, dev, "dL_dout_color")
        radii_c = radii.contiguous()
        _lib.check(lib.gsr_backward(
            P, int(degree), int(M), int(R), W, H, _ptr(bg), _ptr(means3D_c), _ptr(sh_c), _ptr(colors_c), _ptr(scales_c),
            float(scale_modifier), _ptr(rot_c), _ptr(cov_c), _ptr(vm), _ptr(pm), _ptr(cp), float(tan_fovx), float(tan_fovy),
            _ptr(radii_c), _ptr(geomBuffer), geomBuffer.numel(), _ptr(binningBuffer), binningBuffer.numel(), _ptr(imageBuffer),
            imageBuffer.numel(), _ptr(scratch), scratch.numel(), _ptr(grad), _ptr(dL_dmeans2D), _ptr(dL_dcolors), _ptr(dL_dopacity),
            _ptr(dL_dmeans3D), _ptr(dL_dcov3D), _ptr(dL_dsh), _ptr(dL_dscales), _ptr(dL_drotations), _stream(dev)), "gsr_backward")
        if debug:
            torch.cuda.synchronize(dev)
    return dL_dmeans2D, dL_dcolors, dL_dopacity, dL_dmeans3D, dL_dcov3D, dL_dsh, dL_dscales, dL_drotations


def mark_visible(means3D, viewmatrix, projmatrix):
    """Counterpart of `_C.mark_visible` (markVisible, rasterize_points.cu:198-217)."""
    _require_hip(means3D, "means3D")
    lib = _lib.load()
    dev = means3D.device
    P = means3D.size(0)
    present = torch.zeros((P,), dtype=torch.bool, device=dev)
    if P != 0:
        with _lib.device_guard(dev):
            m, vm, pm = _prep(means3D, dev, "means3D"), _prep(viewmatrix, dev, "viewmatrix"), _prep(projmatrix, dev, "projmatrix")
            _lib.check(lib.gsr_mark_visible(P, _ptr(m), _ptr(vm), _ptr(pm), ctypes.c_void_p(present.data_ptr()), _stream(dev)),
                       "gsr_mark_visible")
    return present


class _C:
    """Namespace with the three names the reference binds in ext.cpp:15-19."""
    rasterize_gaussians = staticmethod(rasterize_gaussians_forward)
    rasterize_gaussians_backward = staticmethod(rasterize_gaussians_backward)
    mark_visible = staticmethod(mark_visible)


def cpu_deep_copy_tuple(input_tuple):
    copied_tensors = [item.cpu().clone() if isinstance(item, torch.Tensor) else item for item in input_tuple]
    return tuple(copied_tensors)


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings):
    return _RasterizeGaussians.apply(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                                     raster_settings)


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings):
        args = (raster_settings.bg, means3D, colors_precomp, opacities, scales, rotations, raster_settings.scale_modifier,
                cov3Ds_precomp, raster_settings.viewmatrix, raster_settings.projmatrix, raster_settings.tanfovx,
                raster_settings.tanfovy, raster_settings.image_height, raster_settings.image_width, sh,
                raster_settings.sh_degree, raster_settings.campos, raster_settings.prefiltered, raster_settings.debug)
        if raster_settings.debug:
            cpu_args = cpu_deep_copy_tuple(args)  # snapshot before anything can be corrupted (__init__.py:83-90)
            try:
                num_rendered, color, radii, geomBuffer, binningBuffer, imgBuffer = _C.rasterize_gaussians(*args)
            except Exception as ex:
                torch.save(cpu_args, "snapshot_fw.dump")
                print("\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
                raise ex
        else:
            num_rendered, color, radii, geomBuffer, binningBuffer, imgBuffer = _C.rasterize_gaussians(*args)
        if any(ctx.needs_input_grad):
            check_gradient_rows(means3D.size(0), num_rendered, geomBuffer, means3D.device)
        ctx.raster_settings = raster_settings
        ctx.num_rendered = num_rendered
        ctx.save_for_backward(colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer, binningBuffer,
                              imgBuffer)
        ctx.mark_non_differentiable(radii)
        # autograd would otherwise materialise a zero "gradient" for the int32 radii on every backward (a 500k-element fill kernel
        # between the two blend kernels, 5 us per step on the benchmark); the reference's Function lets it (__init__.py:99-101)
        ctx.set_materialize_grads(False)
        return color, radii

    @staticmethod
    def backward(ctx, grad_out_color, _):
        num_rendered = ctx.num_rendered
        raster_settings = ctx.raster_settings
        if grad_out_color is None:  # only possible with set_materialize_grads(False): nothing flowed into the image
            grad_out_color = torch.zeros((NUM_CHANNELS, raster_settings.image_height, raster_settings.image_width),
                                         dtype=torch.float32, device=ctx.saved_tensors[1].device)
        colors_precomp, means3D, scales, rotations, cov3Ds_precomp, radii, sh, geomBuffer, binningBuffer, imgBuffer = ctx.saved_tensors
        args = (raster_settings.bg, means3D, radii, colors_precomp, scales, rotations, raster_settings.scale_modifier,
                cov3Ds_precomp, raster_settings.viewmatrix, raster_settings.projmatrix, raster_settings.tanfovx,
                raster_settings.tanfovy, grad_out_color, sh, raster_settings.sh_degree, raster_settings.campos, geomBuffer,
                num_rendered, binningBuffer, imgBuffer, raster_settings.debug)
        if raster_settings.debug:
            cpu_args = cpu_deep_copy_tuple(args)
            try:
                grads8 = _C.rasterize_gaussians_backward(*args)
            except Exception as ex:
                torch.save(cpu_args, "snapshot_bw.dump")
                print("\nAn error occured in backward. Writing snapshot_bw.dump for debugging.\n")
                raise ex
        else:
            grads8 = _C.rasterize_gaussians_backward(*args)
        grad_means2D, grad_colors_precomp, grad_opacities, grad_means3D, grad_cov3Ds_precomp, grad_sh, grad_scales, grad_rotations = grads8
        # order of the 9 forward inputs (__init__.py:143-153)
        return (grad_means3D, grad_means2D, grad_sh, grad_colors_precomp, grad_opacities, grad_scales, grad_rotations,
                grad_cov3Ds_precomp, None)


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions):
        with torch.no_grad():
            raster_settings = self.raster_settings
            visible = _C.mark_visible(positions, raster_settings.viewmatrix, raster_settings.projmatrix)
        return visible

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None):
        raster_settings = self.raster_settings
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception('Please provide excatly one of either SHs or precomputed colors!')
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
        # absent inputs become empty CPU tensors, exactly like the reference (__init__.py:196-206)
        if shs is None:
            shs = torch.Tensor([])
        if colors_precomp is None:
            colors_precomp = torch.Tensor([])
        if scales is None:
            scales = torch.Tensor([])
        if rotations is None:
            rotations = torch.Tensor([])
        if cov3D_precomp is None:
            cov3D_precomp = torch.Tensor([])
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                                   raster_settings)


def debug_export(P, R, W, H, geomBuffer, binningBuffer, imgBuffer):
    """Copies the opaque state into named tensors (tests / inspection only)."""
    lib = _lib.load()
    dev = geomBuffer.device
    T = ((W + 15) // 16) * ((H + 15) // 16)
    f = dict(dtype=torch.float32, device=dev)
    out = dict(means2D=torch.zeros((P, 2), **f), depths=torch.zeros((P,), **f), conic_opacity=torch.zeros((P, 4), **f),
               rgb=torch.zeros((P, 3), **f), cov3D=torch.zeros((P, 6), **f),
               clamped=torch.zeros((P, 3), dtype=torch.uint8, device=dev),
               tiles_touched=torch.zeros((P,), dtype=torch.int32, device=dev),
               point_offsets=torch.zeros((P,), dtype=torch.int32, device=dev),
               keys_sorted=torch.zeros((max(R, 1),), dtype=torch.int64, device=dev)[:R],
               point_list=torch.zeros((max(R, 1),), dtype=torch.int32, device=dev)[:R],
               ranges=torch.zeros((T, 2), dtype=torch.int32, device=dev), final_T=torch.zeros((H, W), **f),
               n_contrib=torch.zeros((H, W), dtype=torch.int32, device=dev))
    if P:
        with _lib.device_guard(dev):
            order = ["means2D", "depths", "conic_opacity", "rgb", "cov3D", "clamped", "tiles_touched", "point_offsets",
                     "keys_sorted", "point_list", "ranges", "final_T", "n_contrib"]
            _lib.check(lib.gsr_debug_export(P, R, W, H, _ptr(geomBuffer), _ptr(binningBuffer), _ptr(imgBuffer),
                                            *[_ptr(out[k]) for k in order], _stream(dev)), "gsr_debug_export")
    return out
