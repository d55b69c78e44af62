"""`distCUDA2` on the C ABI: mirror of simple_knn._C.distCUDA2 (submodules/simple-knn/spatial.cu:15-26, ext.cpp:15-17),
called once per sub-scene by GaussianModel.create_from_pcd (scene/gaussian_model.py:144)."""
import ctypes

import torch

from . import _lib


def distCUDA2(points):
    """points: float32 [P,3] on a HIP device -> float32 [P], mean squared distance to the 3 nearest other points."""
    if not points.is_cuda:
        raise _lib.GsrError(f"points must live on a HIP device (got {points.device}); there is no CPU fallback")
    if points.dtype != torch.float32:
        raise _lib.GsrError(f"points must be float32, got {points.dtype}")
    lib = _lib.load()
    P = points.size(0)
    dev = points.device
    means = torch.zeros((P,), dtype=torch.float32, device=dev)  # torch::full({P}, 0.0), spatial.cu:20
    if P:
        with torch.cuda.device(dev):
            pts = points.contiguous()
            scratch = torch.empty(lib.gsr_knn_scratch_bytes(P), dtype=torch.uint8, device=dev)
            s = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            _lib.check(lib.gsr_knn_mean_dist2(P, ctypes.c_void_p(pts.data_ptr()), ctypes.c_void_p(means.data_ptr()),
                                              ctypes.c_void_p(scratch.data_ptr()), scratch.numel(), s), "gsr_knn_mean_dist2")
    return means
