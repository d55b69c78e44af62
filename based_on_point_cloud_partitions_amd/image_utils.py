"""Evaluation metrics of the reference's utils/image_utils.py:17-22 (`mse`, `psnr`) on the C ABI -- what train.py:227 reports per
test view and metrics.py:93 per rendered image (SSIM is `loss.ssim`, LPIPS needs downloaded weights and
is not available offline).  Same names, argument meaning and result shapes; HIP only (a CPU tensor raises).
"""
import ctypes

import torch

from . import _lib


def mse(img1, img2):
    """(((img1 - img2)) ** 2).view(img1.shape[0], -1).mean(1, keepdim=True)  -- utils/image_utils.py:17-18."""
    if not (img1.is_cuda and img2.is_cuda):
        raise _lib.GsrError("mse inputs must live on a HIP device; there is no CPU fallback")
    if img1.shape != img2.shape or img1.dim() < 1 or img1.dtype != torch.float32 or img2.dtype != torch.float32:
        raise _lib.GsrError(f"expected two float32 tensors of equal shape, got {tuple(img1.shape)} and {tuple(img2.shape)}")
    lib = _lib.load()
    a, b = img1.detach().contiguous(), img2.detach().contiguous()
    rows = a.shape[0]
    n = a.numel() // max(rows, 1)
    dev = a.device
    with torch.cuda.device(dev):
        out = torch.empty(rows, 1, dtype=torch.float32, device=dev)
        scratch = torch.empty(lib.gsr_mse_scratch_bytes(rows), dtype=torch.uint8, device=dev)
        s = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(lib.gsr_mse(rows, n, ctypes.c_void_p(a.data_ptr()), ctypes.c_void_p(b.data_ptr()), ctypes.c_void_p(out.data_ptr()),
                               ctypes.c_void_p(scratch.data_ptr()), scratch.numel(), s), "gsr_mse")
    return out


def psnr(img1, img2):
    """20 * log10(1 / sqrt(mse))  -- utils/image_utils.py:20-22 (one value per leading index, shape [rows, 1])."""
    return 20 * torch.log10(1.0 / torch.sqrt(mse(img1, img2)))
