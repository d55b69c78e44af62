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


# ------------------------------------------------------------------------------------------------------------------
# The rest of GaussianModel's densification (scene/gaussian_model.py:231-234,279-295,333-431), same names and state
# surgery; the row gathers `t[selected_pts_mask]` of the six parameters go through one compaction launch.
# ------------------------------------------------------------------------------------------------------------------
def _param_attrs():
    return (("xyz", "_xyz"), ("f_dc", "_features_dc"), ("f_rest", "_features_rest"), ("opacity", "_opacity"),
            ("scaling", "_scaling"), ("rotation", "_rotation"))


def _set_params(gaussians, optimizable):
    for name, attr in _param_attrs():
        setattr(gaussians, attr, optimizable[name])


def replace_tensor_to_optimizer(gaussians, tensor, name):
    """gaussian_model.py:279-295: swaps in a new tensor for one group and zeroes its Adam moments."""
    opt = gaussians.optimizer
    optimizable = {}
    for group in opt.param_groups:
        if group["name"] in _SKIP_GROUPS:
            continue
        if group["name"] == name:
            stored = opt.state.get(group["params"][0], None)
            stored["exp_avg"] = torch.zeros_like(tensor)
            stored["exp_avg_sq"] = torch.zeros_like(tensor)
            del opt.state[group["params"][0]]
            group["params"][0] = nn.Parameter(tensor.requires_grad_(True))
            opt.state[group["params"][0]] = stored
            optimizable[group["name"]] = group["params"][0]
    return optimizable


def reset_opacity(gaussians):
    """gaussian_model.py:231-234."""
    op = torch.sigmoid(gaussians._opacity)
    new = torch.min(op, torch.ones_like(op) * 0.01)
    opacities_new = torch.log(new / (1 - new))  # inverse_sigmoid, utils/general_utils.py:18-19
    gaussians._opacity = replace_tensor_to_optimizer(gaussians, opacities_new, "opacity")["opacity"]


def cat_tensors_to_optimizer(gaussians, tensors_dict):
    """gaussian_model.py:333-356: appends rows to the six parameters and zero rows to their Adam moments."""
    opt = gaussians.optimizer
    optimizable = {}
    for group in opt.param_groups:
        if group["name"] in _SKIP_GROUPS:
            continue
        assert len(group["params"]) == 1
        ext = tensors_dict[group["name"]]
        stored = opt.state.get(group["params"][0], None)
        new_p = nn.Parameter(torch.cat((group["params"][0], ext), dim=0).requires_grad_(True))
        if stored is not None:
            stored["exp_avg"] = torch.cat((stored["exp_avg"], torch.zeros_like(ext)), dim=0)
            stored["exp_avg_sq"] = torch.cat((stored["exp_avg_sq"], torch.zeros_like(ext)), dim=0)
            del opt.state[group["params"][0]]
            group["params"][0] = new_p
            opt.state[new_p] = stored
        else:
            group["params"][0] = new_p
        optimizable[group["name"]] = new_p
    return optimizable


def densification_postfix(gaussians, new_xyz, new_features_dc, new_features_rest, new_opacities, new_scaling, new_rotation):
    """gaussian_model.py:358-376."""
    d = {"xyz": new_xyz, "f_dc": new_features_dc, "f_rest": new_features_rest, "opacity": new_opacities, "scaling": new_scaling,
         "rotation": new_rotation}
    _set_params(gaussians, cat_tensors_to_optimizer(gaussians, d))
    P, dev = gaussians._xyz.shape[0], gaussians._xyz.device
    gaussians.xyz_gradient_accum = torch.zeros((P, 1), device=dev)
    gaussians.denom = torch.zeros((P, 1), device=dev)
    gaussians.max_radii2D = torch.zeros((P,), device=dev)


def _selected_rows(gaussians, mask):
    """The six parameters' rows under `mask` (one compaction launch instead of six boolean indexings)."""
    return compact_rows([getattr(gaussians, attr) for _, attr in _param_attrs()], mask)


def _build_rotation(r):
    """utils/general_utils.py:78-99 (on the input's device)."""
    norm = torch.sqrt(r[:, 0] * r[:, 0] + r[:, 1] * r[:, 1] + r[:, 2] * r[:, 2] + r[:, 3] * r[:, 3])
    q = r / norm[:, None]
    R = torch.zeros((q.size(0), 3, 3), device=r.device)
    r_, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R[:, 0, 0] = 1 - 2 * (y * y + z * z)
    R[:, 0, 1] = 2 * (x * y - r_ * z)
    R[:, 0, 2] = 2 * (x * z + r_ * y)
    R[:, 1, 0] = 2 * (x * y + r_ * z)
    R[:, 1, 1] = 1 - 2 * (x * x + z * z)
    R[:, 1, 2] = 2 * (y * z - r_ * x)
    R[:, 2, 0] = 2 * (x * z - r_ * y)
    R[:, 2, 1] = 2 * (y * z + r_ * x)
    R[:, 2, 2] = 1 - 2 * (x * x + y * y)
    return R


def densify_and_clone(gaussians, grads, grad_threshold, scene_extent):
    """gaussian_model.py:404-418."""
    mask = torch.where(torch.norm(grads, dim=-1) >= grad_threshold, True, False)
    mask = torch.logical_and(mask, torch.max(torch.exp(gaussians._scaling), dim=1).values <= gaussians.percent_dense * scene_extent)
    densification_postfix(gaussians, *_selected_rows(gaussians, mask))


def densify_and_split(gaussians, grads, grad_threshold, scene_extent, N=2, samples=None):
    """gaussian_model.py:378-402.  `samples` (optional, [N * selected, 3]) replaces the torch.normal draw (tests)."""
    n_init = gaussians._xyz.shape[0]
    dev = gaussians._xyz.device
    padded_grad = torch.zeros((n_init,), device=dev)
    padded_grad[:grads.shape[0]] = grads.squeeze()
    mask = torch.where(padded_grad >= grad_threshold, True, False)
    mask = torch.logical_and(mask, torch.max(torch.exp(gaussians._scaling), dim=1).values > gaussians.percent_dense * scene_extent)
    xyz, f_dc, f_rest, opacity, scaling, rotation = _selected_rows(gaussians, mask)
    stds = torch.exp(scaling).repeat(N, 1)
    if samples is None:
        samples = torch.normal(mean=torch.zeros((stds.size(0), 3), device=dev), std=stds)
    rots = _build_rotation(rotation).repeat(N, 1, 1)
    new_xyz = torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + xyz.repeat(N, 1)
    new_scaling = torch.log(torch.exp(scaling).repeat(N, 1) / (0.8 * N))
    densification_postfix(gaussians, new_xyz, f_dc.repeat(N, 1, 1), f_rest.repeat(N, 1, 1), opacity.repeat(N, 1), new_scaling,
                          rotation.repeat(N, 1))
    prune_filter = torch.cat((mask, torch.zeros(N * int(mask.sum()), device=dev, dtype=bool)))
    prune_points(gaussians, prune_filter)


def densify_and_prune(gaussians, max_grad, min_opacity, extent, max_screen_size, split_samples=None):
    """gaussian_model.py:420-431."""
    grads = gaussians.xyz_gradient_accum / gaussians.denom
    grads[grads.isnan()] = 0.0
    densify_and_clone(gaussians, grads, max_grad, extent)
    densify_and_split(gaussians, grads, max_grad, extent, samples=split_samples)
    prune_mask = (torch.sigmoid(gaussians._opacity) < min_opacity).squeeze()
    if max_screen_size:
        big_points_vs = gaussians.max_radii2D > max_screen_size
        big_points_ws = torch.exp(gaussians._scaling).max(dim=1).values > 0.1 * extent
        prune_mask = torch.logical_or(torch.logical_or(prune_mask, big_points_vs), big_points_ws)
    prune_points(gaussians, prune_mask)
