"""Loss block of the training step on the C ABI.

Mirrors the names of the reference's utils/loss_utils.py (`l1_loss` :17-18, `ssim` :33-63) and adds the combination
train.py:112-117 actually uses, `l1_ssim_loss(image, gt, lambda_dssim, l1_image) = (1-l)*L1(l1_image, gt) + l*(1-SSIM(image, gt))`,
as ONE forward and ONE backward kernel; its gradient is the dL/dcolor the rasterizer's backward consumes.  HIP only (no CPU fallback).
"""
import ctypes

import torch

from . import _lib


def _check(img1, img2):
    if not (img1.is_cuda and img2.is_cuda):
        raise _lib.GsrError("loss inputs must live on a HIP device; there is no CPU fallback")
    if img1.shape != img2.shape or img1.dim() not in (3, 4) or (img1.dim() == 4 and img1.shape[0] != 1):
        raise _lib.GsrError(f"expected two [C,H,W] images of equal shape, got {tuple(img1.shape)} and {tuple(img2.shape)}")
    if img1.dtype != torch.float32 or img2.dtype != torch.float32:
        raise _lib.GsrError("loss inputs must be float32")


class _L1SSIM(torch.autograd.Function):
    """which: 0 -> combined loss, 1 -> L1 alone, 2 -> SSIM alone (all from the same two kernels).  `img_l1` (optional) is the
    image the L1 term is taken on instead of img1."""

    @staticmethod
    def forward(ctx, img1, img2, lambda_dssim, which, img_l1):
        _check(img1, img2)
        if img_l1 is not None:
            _check(img_l1, img2)
        lib = _lib.load()
        dev = img1.device
        a, b = img1.contiguous(), img2.contiguous()
        l = img_l1.contiguous() if img_l1 is not None else None
        C, H, W = a.shape[-3:]
        lam = {0: float(lambda_dssim), 1: 0.0, 2: 1.0}[which]
        with torch.cuda.device(dev):
            scratch = torch.empty(lib.gsr_loss_scratch_bytes(C, H, W), dtype=torch.uint8, device=dev)
            out = torch.empty(3, dtype=torch.float32, device=dev)
            s = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            _lib.check(lib.gsr_loss_forward(C, H, W, ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(b.data_ptr()),
                                            ctypes.c_void_p(l.data_ptr()) if l is not None else None, lam,
                                            ctypes.c_void_p(scratch.data_ptr()), scratch.numel(), ctypes.c_void_p(out.data_ptr()), s),
                       "gsr_loss_forward")
        ctx.save_for_backward(a, b, scratch, *([l] if l is not None else []))
        ctx.lam, ctx.which, ctx.shape, ctx.split = lam, which, img1.shape, l is not None
        return out[which]

    @staticmethod
    def backward(ctx, grad):
        a, b, scratch = ctx.saved_tensors[:3]
        l = ctx.saved_tensors[3] if ctx.split else None
        lib = _lib.load()
        dev = a.device
        C, H, W = a.shape[-3:]
        g = grad.to(torch.float32).contiguous()
        if ctx.which == 2:
            g = -g  # SSIM = 1 - loss(lambda = 1)
        with torch.cuda.device(dev):
            d = torch.empty_like(a)
            dl = torch.empty_like(l) if l is not None else None
            s = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            _lib.check(lib.gsr_loss_backward(C, H, W, ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(b.data_ptr()),
                                             ctypes.c_void_p(l.data_ptr()) if l is not None else None, ctx.lam,
                                             ctypes.c_void_p(scratch.data_ptr()), scratch.numel(), ctypes.c_void_p(g.data_ptr()),
                                             ctypes.c_void_p(d.data_ptr()), ctypes.c_void_p(dl.data_ptr()) if dl is not None else None, s),
                       "gsr_loss_backward")
        return d.view(ctx.shape), None, None, None, (dl.view(ctx.shape) if dl is not None else None)


def l1_ssim_loss(image, gt, lambda_dssim=0.2, l1_image=None):
    """(1 - lambda_dssim) * l1_loss(l1_image, gt) + lambda_dssim * (1 - ssim(image, gt)) in one kernel pair.
    `l1_image=None` takes the L1 term on `image` itself (plain 3DGS); the reference's train.py:112-117 passes the
    appearance-decoupled image there: `l1_ssim_loss(image, gt_image, opt.lambda_dssim, l1_image=decouple_image)`."""
    return _L1SSIM.apply(image, gt, lambda_dssim, 0, l1_image)


def l1_loss(network_output, gt):
    """utils/loss_utils.py:17-18."""
    return _L1SSIM.apply(network_output, gt, 0.0, 1, None)


def ssim(img1, img2, window_size=11, size_average=True):
    """utils/loss_utils.py:33-63 for the configuration the training loop uses (11-tap window, mean over all elements)."""
    if window_size != 11 or not size_average:
        raise NotImplementedError("only window_size=11, size_average=True (the training configuration) is implemented")
    return _L1SSIM.apply(img1, img2, 1.0, 2, None)
