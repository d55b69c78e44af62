"""Per-iteration densification bookkeeping and point pruning of the reference's training loop, on the HIP library.

`add_densification_stats` = train.py:140-142 (`max_radii2D[vis] = max(max_radii2D[vis], radii[vis])`) +
`GaussianModel.add_densification_stats` (scene/gaussian_model.py:433-435) in one streaming pass instead of three masked
read-modify-writes (each a `nonzero` with a device sync in torch).
`prune_points` = `GaussianModel.prune_points` + `_prune_optimizer` (gaussian_model.py:297-331): the same optimizer-state
surgery, with the 21 boolean indexings `t[mask]` (6 parameters, 12 Adam moments, 3 statistics) done by ONE mask scan and
ONE gather launch (C ABI gsr_compact_plan / gsr_compact_rows).  Results are identical to torch's boolean indexing.
"""
import ctypes

import torch
import torch.nn as nn

from . import _lib

_SKIP_GROUPS = ("appearance_embeddings", "appearance_network")  # gaussian_model.py:300,335


def _stream(dev):
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _check(t, what, dtype=torch.float32):
    if not t.is_cuda or t.dtype != dtype or not t.is_contiguous():
        raise _lib.GsrError(f"{what} must be a contiguous {dtype} tensor on a HIP device (no CPU fallback); got {t.dtype} on {t.device}")
    return t


def add_densification_stats(gaussians, viewspace_point_tensor, radii):
    """In-place update of gaussians.max_radii2D [P], .xyz_gradient_accum [P,1], .denom [P,1] for the points with radii > 0."""
    lib = _lib.load()
    grad = viewspace_point_tensor.grad if isinstance(viewspace_point_tensor, torch.Tensor) and viewspace_point_tensor.grad is not None \
        else viewspace_point_tensor
    P = radii.numel()
    if grad.shape != (P, 3):
        raise _lib.GsrError(f"viewspace gradient must be [P, 3], got {tuple(grad.shape)}")
    dev = radii.device
    _check(radii, "radii", torch.int32)
    g = _check(grad if grad.is_contiguous() else grad.contiguous(), "viewspace gradient")
    for t, n in ((gaussians.max_radii2D, "max_radii2D"), (gaussians.xyz_gradient_accum, "xyz_gradient_accum"), (gaussians.denom, "denom")):
        _check(t, n)
        if t.numel() != P:
            raise _lib.GsrError(f"{n} has {t.numel()} elements, expected {P}")
    with torch.cuda.device(dev):
        _lib.check(lib.gsr_densification_stats(P, radii.data_ptr(), g.data_ptr(), gaussians.max_radii2D.data_ptr(),
                                               gaussians.xyz_gradient_accum.data_ptr(), gaussians.denom.data_ptr(), _stream(dev)),
                   "gsr_densification_stats")


def compact_rows(tensors, keep_mask):
    """[t[keep_mask] for t in tensors] for float32 tensors sharing their first dimension, with one scan and one gather."""
    lib = _lib.load()
    P = keep_mask.numel()
    dev = keep_mask.device
    if not keep_mask.is_cuda or keep_mask.dtype != torch.bool:
        raise _lib.GsrError("keep_mask must be a bool tensor on a HIP device")
    mask8 = keep_mask.contiguous().view(torch.uint8)
    srcs = []
    for t in tensors:
        _check(t.detach(), "tensor")
        if t.shape[0] != P:
            raise _lib.GsrError(f"tensor with {t.shape[0]} rows, mask with {P}")
        srcs.append(t.detach())
    with torch.cuda.device(dev):
        scratch = torch.empty(lib.gsr_compact_scratch_bytes(P), dtype=torch.uint8, device=dev)
        count = ctypes.c_int64(0)
        _lib.check(lib.gsr_compact_plan(P, mask8.data_ptr(), scratch.data_ptr(), scratch.numel(), ctypes.byref(count), _stream(dev)),
                   "gsr_compact_plan")
        n = int(count.value)
        outs = [torch.empty((n,) + tuple(t.shape[1:]), dtype=torch.float32, device=dev) for t in srcs]
        desc = (_lib.CompactTensor * len(srcs))(*[
            _lib.CompactTensor(s.data_ptr(), o.data_ptr(), (s.numel() // P) if P else 0) for s, o in zip(srcs, outs)])
        if P and n:
            _lib.check(lib.gsr_compact_rows(P, mask8.data_ptr(), scratch.data_ptr(), scratch.numel(), len(srcs),
                                            ctypes.cast(desc, ctypes.c_void_p), _stream(dev)), "gsr_compact_rows")
    return outs


def prune_points(gaussians, mask):
    """GaussianModel.prune_points(mask) (gaussian_model.py:317-331): drops the points where `mask` is True from the six
    parameters, their optimizer state and the densification statistics."""
    valid = ~mask
    opt = gaussians.optimizer
    jobs = []  # (kind, group, state or None)
    tensors = []
    for group in opt.param_groups:
        if group["name"] in _SKIP_GROUPS:
            continue
        p = group["params"][0]
        stored = opt.state.get(p, None)
        tensors.append(p)
        if stored is not None:
            tensors.extend([stored["exp_avg"], stored["exp_avg_sq"]])
        jobs.append((group, stored))
    tensors.extend([gaussians.xyz_gradient_accum, gaussians.denom, gaussians.max_radii2D])
    outs = iter(compact_rows(tensors, valid))
    optimizable = {}
    for group, stored in jobs:
        new_p = nn.Parameter(next(outs).requires_grad_(True))
        if stored is not None:
            stored["exp_avg"] = next(outs)
            stored["exp_avg_sq"] = next(outs)
            del opt.state[group["params"][0]]
            group["params"][0] = new_p
            opt.state[new_p] = stored
        else:
            group["params"][0] = new_p
        optimizable[group["name"]] = new_p
    gaussians._xyz = optimizable["xyz"]
    gaussians._features_dc = optimizable["f_dc"]
    gaussians._features_rest = optimizable["f_rest"]
    gaussians._opacity = optimizable["opacity"]
    gaussians._scaling = optimizable["scaling"]
    gaussians._rotation = optimizable["rotation"]
    gaussians.xyz_gradient_accum = next(outs)
    gaussians.denom = next(outs)
    gaussians.max_radii2D = next(outs)
    return optimizable
