"""Appearance decoupling of the reference's training loop on the C ABI.

`decouple_appearance(image, gaussians, view_idx)` has the name, arguments and results of scene/ptgs/appearance_network.py:7-17
(train.py:110-111).  The network is the caller's own `gaussians.appearance_network` (the reference's AppearanceNetwork, used as
is: its parameters are read, never copied).  Its low-resolution head -- conv1 and the four pixel-shuffle blocks, at most a
quarter of the image's pixels -- runs as the module's own layers, except the last block `up4` (PixelShuffle + conv 8->16 + ReLU
at half the image's resolution, a third of the head's time), which is HIP too when it has the reference's shape; everything from
the bilinear upsampling to full resolution on (:54-60) and the final `mapping_image * image` (:15) is one HIP forward and one HIP
backward (csrc/appearance.hip): as stock torch ops that tail is 2.6 of the network's 3.7 ms per iteration at 1080p.  HIP only: a
CPU tensor or a missing library raises.
"""
import ctypes

import torch
import torch.nn.functional as F

from . import _lib


def _p(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else None


def _check_tail_shapes(feat, w2, b2, w3, b3, image):
    if feat.dim() != 3 or feat.shape[0] != 16 or tuple(w2.shape) != (16, 16, 3, 3) or tuple(b2.shape) != (16,) or \
            tuple(w3.shape) != (3, 16, 3, 3) or tuple(b3.shape) != (3,) or image.dim() != 3 or image.shape[0] != 3:
        raise _lib.GsrError("appearance tail: expected feat [16,h,w], conv2 16->16 3x3, conv3 16->3 3x3 and an image [3,H,W] "
                            f"(appearance_network.py:47-48), got {tuple(feat.shape)}, {tuple(w2.shape)}, {tuple(w3.shape)}, "
                            f"{tuple(image.shape)}")


class _AppearanceTail(torch.autograd.Function):
    @staticmethod
    def forward(ctx, feat, w2, b2, w3, b3, image):
        for t in (feat, w2, b2, w3, b3, image):
            if not t.is_cuda:
                raise _lib.GsrError("appearance tail inputs must live on a HIP device; there is no CPU fallback")
            if t.dtype != torch.float32:
                raise _lib.GsrError("appearance tail inputs must be float32")
        _check_tail_shapes(feat, w2, b2, w3, b3, image)
        lib = _lib.load()
        feat, w2, b2, w3, b3, image = (t.contiguous() for t in (feat, w2, b2, w3, b3, image))
        ctx.set_materialize_grads(False)
        with torch.cuda.device(image.device):
            transformed, mapping, saved = _tail_forward(lib, feat, w2, b2, w3, b3, image)
        ctx.save_for_backward(feat, w2, w3, image, mapping, saved)
        return transformed, mapping

    @staticmethod
    def backward(ctx, g_transformed, g_mapping):
        feat, w2, w3, image, mapping, saved = ctx.saved_tensors
        with torch.cuda.device(image.device):
            return _tail_backward(_lib.load(), feat, w2, w3, image, mapping, saved, g_transformed, g_mapping, False)


def _tail_forward(lib, feat, w2, b2, w3, b3, image):
    dev = image.device
    H, W = image.shape[1:]
    saved = torch.empty(lib.gsr_appearance_saved_bytes(H, W), dtype=torch.uint8, device=dev)
    mapping, transformed = torch.empty_like(image), torch.empty_like(image)
    s = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(lib.gsr_appearance_tail_forward(H, W, feat.shape[1], feat.shape[2], _p(feat), _p(w2), _p(b2), _p(w3), _p(b3), _p(image),
                                               _p(mapping), _p(transformed), _p(saved), saved.numel(), s), "gsr_appearance_tail_forward")
    return transformed, mapping, saved


def _tail_backward(lib, feat, w2, w3, image, mapping, saved, g_transformed, g_mapping, mask_relu):
    dev = image.device
    H, W = image.shape[1:]
    g_t = g_transformed.to(torch.float32).contiguous() if g_transformed is not None else torch.zeros_like(image)
    g_m = g_mapping.to(torch.float32).contiguous() if g_mapping is not None else None
    d_image, d_feat = torch.empty_like(image), torch.empty_like(feat)
    d_w2, d_w3 = torch.empty_like(w2), torch.empty_like(w3)
    d_b2 = torch.empty(16, dtype=torch.float32, device=dev)
    d_b3 = torch.empty(3, dtype=torch.float32, device=dev)
    scratch = torch.empty(lib.gsr_appearance_scratch_bytes(H, W), dtype=torch.uint8, device=dev)
    s = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    _lib.check(lib.gsr_appearance_tail_backward(H, W, feat.shape[1], feat.shape[2], _p(feat), _p(w2), _p(w3), _p(image), _p(mapping),
                                                _p(saved), saved.numel(), _p(g_t), _p(g_m), _p(d_image), _p(d_feat), _p(d_w2),
                                                _p(d_b2), _p(d_w3), _p(d_b3), _p(scratch), scratch.numel(), int(mask_relu), s),
               "gsr_appearance_tail_backward")
    return d_feat, d_w2, d_b2, d_w3, d_b3, d_image


class _AppearanceUpTail(torch.autograd.Function):
    """`up4` (PixelShuffle(2) -> Conv2d(8, 16, 3, padding=1) -> ReLU) and the tail in one node: the ReLU's mask is applied where the
    tail's backward writes dL/dfeat, and feat never passes through torch."""

    @staticmethod
    def forward(ctx, x, w4, b4, w2, b2, w3, b3, image):
        for t in (x, w4, b4, w2, b2, w3, b3, image):
            if not t.is_cuda or t.dtype != torch.float32:
                raise _lib.GsrError("appearance inputs must be float32 tensors on a HIP device; there is no CPU fallback")
        lib = _lib.load()
        dev = image.device
        x, w4, b4, w2, b2, w3, b3, image = (t.contiguous() for t in (x, w4, b4, w2, b2, w3, b3, image))
        fh, fw = 2 * x.shape[1], 2 * x.shape[2]
        ctx.set_materialize_grads(False)
        with torch.cuda.device(dev):
            feat = torch.empty(16, fh, fw, dtype=torch.float32, device=dev)
            s = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            _lib.check(lib.gsr_appearance_up_forward(fh, fw, _p(x), _p(w4), _p(b4), _p(feat), s), "gsr_appearance_up_forward")
            transformed, mapping, saved = _tail_forward(lib, feat, w2, b2, w3, b3, image)
        ctx.save_for_backward(x, w4, feat, w2, w3, image, mapping, saved)
        return transformed, mapping

    @staticmethod
    def backward(ctx, g_transformed, g_mapping):
        x, w4, feat, w2, w3, image, mapping, saved = ctx.saved_tensors
        lib = _lib.load()
        dev = image.device
        with torch.cuda.device(dev):
            dz, d_w2, d_b2, d_w3, d_b3, d_image = _tail_backward(lib, feat, w2, w3, image, mapping, saved, g_transformed, g_mapping, True)
            d_x, d_w4 = torch.empty_like(x), torch.empty_like(w4)
            d_b4 = torch.empty(16, dtype=torch.float32, device=dev)
            fh, fw = feat.shape[1:]
            scratch = torch.empty(lib.gsr_appearance_up_scratch_bytes(fh, fw), dtype=torch.uint8, device=dev)
            s = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
            _lib.check(lib.gsr_appearance_up_backward(fh, fw, _p(x), _p(w4), _p(dz), _p(d_x), _p(d_w4), _p(d_b4), _p(scratch),
                                                      scratch.numel(), s), "gsr_appearance_up_backward")
        return d_x, d_w4, d_b4, d_w2, d_b2, d_w3, d_b3, d_image


class _BiasReLU(torch.autograd.Function):
    """relu(x + bias[c]) for x [1,C,H,W] fresh from a bias-free convolution (written in place), with dz and the bias gradient from one
    kernel in the backward -- instead of torch's bias add, ReLU, threshold_backward and bias-gradient reduction per layer."""

    @staticmethod
    def forward(ctx, x, bias):
        lib = _lib.load()
        C, n = x.shape[1], x.shape[2] * x.shape[3]
        with torch.cuda.device(x.device):
            s = ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
            _lib.check(lib.gsr_bias_relu_forward(C, n, _p(x), _p(bias), s), "gsr_bias_relu_forward")
        ctx.mark_dirty(x)
        ctx.save_for_backward(x)
        return x

    @staticmethod
    def backward(ctx, g):
        (y,) = ctx.saved_tensors
        lib = _lib.load()
        C, n = y.shape[1], y.shape[2] * y.shape[3]
        g = g.contiguous()
        with torch.cuda.device(y.device):
            dz = torch.empty_like(y)
            db = torch.empty(C, dtype=torch.float32, device=y.device)
            scratch = torch.empty(lib.gsr_bias_relu_scratch_bytes(C), dtype=torch.uint8, device=y.device)
            s = ctypes.c_void_p(torch.cuda.current_stream(y.device).cuda_stream)
            _lib.check(lib.gsr_bias_relu_backward(C, n, _p(g), _p(y), _p(dz), _p(db), _p(scratch), scratch.numel(), s), "gsr_bias_relu_backward")
        return dz, db


def _conv_bias_relu(conv, x):
    """relu(conv(x)) for a plain 3x3 Conv2d followed by nn.ReLU: the module's own weights, torch's convolution, fused bias + ReLU."""
    y = F.conv2d(x, conv.weight, None, conv.stride, conv.padding, conv.dilation, conv.groups)
    return _BiasReLU.apply(y, conv.bias)


def _plain_conv(conv):
    return isinstance(conv, torch.nn.Conv2d) and conv.bias is not None and conv.padding_mode == "zeros" and conv.weight.dtype == torch.float32


def _has_hooks(p):
    """Tensor hooks or post-accumulate-grad hooks on a parameter: their owners expect the gradient to arrive through autograd."""
    return bool(getattr(p, "_backward_hooks", None)) or bool(getattr(p, "_post_accumulate_grad_hooks", None))


_SIDE_STREAMS = {}


def _side_stream(device):
    """One extra HIP stream per device for work that may run beside the main stream (the head's weight gradients)."""
    key = (device.type, device.index)
    if key not in _SIDE_STREAMS:
        _SIDE_STREAMS[key] = torch.cuda.Stream(device=device)
    return _SIDE_STREAMS[key]


class _Head(torch.autograd.Function):
    """conv1 + up1..up3 of the reference's AppearanceNetwork (appearance_network.py:48-51) as ONE autograd node around torch's own
    convolutions, so that the backward can be ordered for the training step: the DATA-gradient chain
    (up3 -> up2 -> up1 -> conv1 -> the network's input, which carries dL/dimage on to the rasterizer backward) runs on the caller's
    stream, the four WEIGHT gradients -- nothing downstream needs them before the optimizer step -- are enqueued on a side stream
    and run beside the rasterizer backward (its blend kernel is bound by VALU issue in 8160 workgroups; MIOpen's weight-gradient
    kernels fit in beside it).  The side stream is joined, and the weight gradients are accumulated into `.grad`, in a callback the
    autograd engine runs when the whole backward pass has been enqueued (`queue_callback`): after `loss.backward()` returns
    everything is ordered on the caller's stream as usual.  Measured at 1080p (tools/r4_head_probe.py): head forward 0.136 ms,
    backward 0.40 ms of which data chain 0.160 and weight gradients 0.188.
    Limit: the weight gradients reach the parameters' `.grad` only (what `loss.backward()` -- train.py:118 -- uses);
    `torch.autograd.grad(..., inputs=[a head weight])`, tensor hooks and AccumulateGrad consumers (a DDP reducer) do not see them:
    `_head` takes this path only for hook-free leaf weights, and GSR_APPEARANCE_OVERLAP=0 switches back to plain autograd (required
    when the network is wrapped in DistributedDataParallel; the reference trains every partition's network on its own, train.py:65-73)."""

    @staticmethod
    def forward(ctx, x, *params):
        lib = _lib.load()
        ws, bs = params[0::2], params[1::2]
        ins, outs = [], []
        h = x
        with torch.cuda.device(x.device):
            s = ctypes.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
            for i, (w, b) in enumerate(zip(ws, bs)):
                if i:
                    h = F.pixel_shuffle(h, 2)
                ins.append(h)
                h = F.conv2d(h, w, None, 1, 1)
                _lib.check(lib.gsr_bias_relu_forward(h.shape[1], h.shape[2] * h.shape[3], _p(h), _p(b.contiguous()), s), "gsr_bias_relu_forward")
                outs.append(h)
        ctx.save_for_backward(*ins, *outs, *ws)
        ctx.n = len(ws)
        ctx.param_refs = params  # the Parameters themselves: their .grad is written by the end-of-backward callback
        return h

    @staticmethod
    def backward(ctx, g):
        n = ctx.n
        saved = ctx.saved_tensors
        ins, outs, ws = saved[:n], saved[n:2 * n], saved[2 * n:]
        lib = _lib.load()
        dev = g.device
        main, side = torch.cuda.current_stream(dev), _side_stream(dev)
        cb = torch.ops.aten.convolution_backward
        dbs, dws = [None] * n, [None] * n
        g = g.contiguous()
        with torch.cuda.device(dev):
            s = ctypes.c_void_p(main.cuda_stream)
            for i in reversed(range(n)):
                y = outs[i]
                C, npx = y.shape[1], y.shape[2] * y.shape[3]
                dz = torch.empty_like(y)
                db = torch.empty(C, dtype=torch.float32, device=dev)
                scratch = torch.empty(lib.gsr_bias_relu_scratch_bytes(C), dtype=torch.uint8, device=dev)
                _lib.check(lib.gsr_bias_relu_backward(C, npx, _p(g), _p(y), _p(dz), _p(db), _p(scratch), scratch.numel(), s),
                           "gsr_bias_relu_backward")
                dbs[i] = db
                ready = main.record_event()
                with torch.cuda.stream(side):
                    side.wait_event(ready)
                    dws[i] = cb(dz, ins[i], ws[i], None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [False, True, False])[1]
                # everything the side-stream kernel reads was allocated on the main stream: without these the caching allocator may hand
                # the saved activations to the next main-stream allocation (the rasterizer backward's scratch) as soon as autograd drops
                # them, while the weight-gradient kernel is still reading
                for t in (dz, ins[i], ws[i]):
                    t.record_stream(side)
                if i > 0 or ctx.needs_input_grad[0]:
                    gin = cb(dz, ins[i], ws[i], None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, [True, False, False])[0]
                    g = F.pixel_unshuffle(gin, 2).contiguous() if i > 0 else gin
                else:
                    g = None
        weights = ctx.param_refs[0::2]

        def join():
            main.wait_stream(side)
            cur = torch.cuda.current_stream(dev)  # backward() called under another stream than the forward's: order the adds there too
            if cur != main:
                cur.wait_stream(side)
            with torch.no_grad():
                for w, dw in zip(weights, dws):
                    dw.record_stream(cur)
                    if w.grad is None:
                        w.grad = dw
                    else:
                        w.grad.add_(dw)

        torch.autograd.Variable._execution_engine.queue_callback(join)
        grads = [g]
        for i in range(n):
            grads += [None, dbs[i]]
        return tuple(grads)


def _head(net, x):
    """conv1 + up1..up3 (appearance_network.py:48-51).  When the layers are the reference's (Conv2d -> ReLU, PixelShuffle -> Conv2d ->
    ReLU) the convolutions stay torch's, bias + ReLU are fused around them and the backward's weight gradients run on a side stream
    (_Head); any other module is simply called."""
    ups = (net.up1, net.up2, net.up3)
    standard = _plain_conv(net.conv1) and isinstance(net.relu, torch.nn.ReLU) and x.is_cuda and all(
        isinstance(getattr(u, "pixel_shuffle", None), torch.nn.PixelShuffle) and _plain_conv(getattr(u, "conv", None)) and
        isinstance(getattr(u, "relu", None), torch.nn.ReLU) and len(list(u.children())) == 3 for u in ups)
    if not standard:
        return net.up3(net.up2(net.up1(net.relu(net.conv1(x)))))
    convs = (net.conv1, net.up1.conv, net.up2.conv, net.up3.conv)
    overlap = _lib.get_variant("APPEARANCE_OVERLAP", 1) and all(
        tuple(c.kernel_size) == (3, 3) and tuple(c.stride) == (1, 1) and tuple(c.padding) == (1, 1) and tuple(c.dilation) == (1, 1) and
        c.groups == 1 for c in convs) and all(getattr(u.pixel_shuffle, "upscale_factor", None) == 2 for u in ups) and \
        torch.is_grad_enabled() and all(c.weight.requires_grad and c.weight.is_leaf and not _has_hooks(c.weight) for c in convs)
    if overlap:
        params = []
        for c in convs:
            params += [c.weight, c.bias]
        return _Head.apply(x.contiguous(), *params)
    h = _conv_bias_relu(net.conv1, x)
    for u in ups:
        h = _conv_bias_relu(u.conv, u.pixel_shuffle(h))
    return h


def appearance_up_tail(x, up_weight, up_bias, conv2_weight, conv2_bias, conv3_weight, conv3_bias, image):
    """(transformed_image, mapping_image) from `up3`'s output [32,h,w] (appearance_network.py:51) and the rendered image: `up4` + tail."""
    if x.dim() != 3 or x.shape[0] != 32 or tuple(up_weight.shape) != (16, 8, 3, 3) or tuple(up_bias.shape) != (16,):
        raise _lib.GsrError(f"appearance up block: expected x [32,h,w] and Conv2d(8, 16, 3), got {tuple(x.shape)}, {tuple(up_weight.shape)}")
    _check_tail_shapes(x.new_empty(16, 1, 1), conv2_weight, conv2_bias, conv3_weight, conv3_bias, image)
    return _AppearanceUpTail.apply(x, up_weight, up_bias, conv2_weight, conv2_bias, conv3_weight, conv3_bias, image)


def appearance_tail(feat, conv2_weight, conv2_bias, conv3_weight, conv3_bias, image):
    """(transformed_image, mapping_image) from `up4`'s output [16,h,w] and the rendered image [3,H,W] (appearance_network.py:54-60,15)."""
    return _AppearanceTail.apply(feat, conv2_weight, conv2_bias, conv3_weight, conv3_bias, image)


def decouple_appearance(image, gaussians, view_idx):
    """scene/ptgs/appearance_network.py:7-17, same arguments and results: (transformed_image, mapping_image)."""
    appearance_embedding = gaussians.get_apperance_embedding(view_idx)
    H, W = image.size(1), image.size(2)
    h, w = H // 32, W // 32
    crop_image_down = F.interpolate(image[None], size=(h, w), mode="bilinear", align_corners=True)[0]
    x = torch.cat([crop_image_down, appearance_embedding[None].repeat(h, w, 1).permute(2, 0, 1)], dim=0)[None]
    net = gaussians.appearance_network
    x = _head(net, x)  # appearance_network.py:48-51: the module's own parameters and torch's convolutions
    up4_conv = getattr(net.up4, "conv", None)
    if up4_conv is not None and tuple(up4_conv.weight.shape) == (16, 8, 3, 3) and up4_conv.bias is not None and x.shape[1] == 32 and \
            getattr(getattr(net.up4, "pixel_shuffle", None), "upscale_factor", None) == 2:
        return appearance_up_tail(x[0], up4_conv.weight, up4_conv.bias, net.conv2.weight, net.conv2.bias, net.conv3.weight, net.conv3.bias,
                                  image)
    return appearance_tail(net.up4(x)[0], net.conv2.weight, net.conv2.bias, net.conv3.weight, net.conv3.bias, image)
