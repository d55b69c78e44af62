"""`torch.optim.Adam` of the reference's training loop on one HIP launch.

scene/gaussian_model.py:164-175 builds `torch.optim.Adam(l, lr=0.0, eps=1e-15)` over eight parameter groups (xyz, f_dc,
f_rest, opacity, scaling, rotation, appearance embeddings / network) and train.py:155-157 calls `.step()` /
`.zero_grad(set_to_none=True)` every iteration.  `Adam` below is a drop-in: same constructor, same `param_groups`
(`lr`, `name`, ...), same per-parameter state keys (`step`, `exp_avg`, `exp_avg_sq`), so the state surgery of
`cat_tensors_to_optimizer` / `_prune_optimizer` / `replace_tensor_to_optimizer` (gaussian_model.py:297-370) and
`state_dict()` / `load_state_dict()` (`capture` / `restore`, :69-100) keep working; `step()` updates every parameter
that has a gradient with ONE kernel (C ABI `gsr_adam_step`, csrc/adam.hip) instead of torch's per-op chains.
Dense Adam exactly as torch computes it (weight_decay = 0, amsgrad = False); no CPU fallback.
"""
import ctypes

import torch

from . import _lib


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        if lr < 0.0 or eps < 0.0 or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0:
            raise ValueError("invalid Adam hyper-parameter")  # torch/optim/adam.py raises the same way
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load()
        # one launch per (device, betas, eps) combination: the reference has exactly one
        batches = {}
        keep = []
        for group in self.param_groups:
            beta1, beta2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.grad.is_sparse:
                    raise RuntimeError("Adam does not support sparse gradients")
                if not p.is_cuda or p.dtype != torch.float32 or not p.is_contiguous():
                    raise _lib.GsrError("the HIP Adam step takes contiguous float32 parameters on a HIP device (no CPU fallback); "
                                        f"got {p.dtype} on {p.device}")
                state = self.state[p]
                if len(state) == 0:  # lazy state initialisation, as torch/optim/adam.py
                    state["step"] = torch.tensor(0.0, dtype=torch.float32)
                    state["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                state["step"] += 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                m, v = state["exp_avg"], state["exp_avg_sq"]
                if not (m.is_contiguous() and v.is_contiguous()):
                    raise _lib.GsrError("exp_avg / exp_avg_sq must be contiguous")
                keep.append(g)
                batches.setdefault((p.device, float(beta1), float(beta2), float(group["eps"])), []).append(
                    _lib.AdamTensor(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), float(group["lr"]),
                                    int(state["step"])))
        for (dev, beta1, beta2, eps), items in batches.items():
            arr = (_lib.AdamTensor * len(items))(*items)
            with torch.cuda.device(dev):
                _lib.check(lib.gsr_adam_step(len(items), ctypes.cast(arr, ctypes.c_void_p), beta1, beta2, eps,
                                             ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), "gsr_adam_step")
        return loss
