#!/usr/bin/env python3
"""How much of the appearance head's backward (conv1, up1-up3: the caller's torch layers) could overlap with the rasterizer backward?
Only its WEIGHT gradients: the data-gradient chain feeds dL/dimage, which the rasterizer backward needs.  Times (HIP events, 1080p):
the head's forward, its whole backward, the data-gradient chain alone, the weight gradients alone."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from based_on_point_cloud_partitions_amd.synthetic import make_appearance_model  # noqa: E402


def timeit(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n


def main():
    dev = "cuda"
    net = make_appearance_model(16, 3, dev).appearance_network
    H, W = 1080, 1920
    h, w = H // 32, W // 32
    x0 = torch.randn(1, 67, h, w, device=dev)
    convs = [net.conv1, net.up1.conv, net.up2.conv, net.up3.conv]

    def forward(keep=None):
        x = x0
        for i, c in enumerate(convs):
            if i:
                x = F.pixel_shuffle(x, 2)
            if keep is not None:
                keep.append(x)
            x = F.relu(F.conv2d(x, c.weight, c.bias, padding=1))
        return x

    ins = []
    y = forward(ins)
    g = torch.randn_like(y)
    xr = x0.clone().requires_grad_(True)

    def full():
        x = xr
        for i, c in enumerate(convs):
            if i:
                x = F.pixel_shuffle(x, 2)
            x = F.relu(F.conv2d(x, c.weight, c.bias, padding=1))
        x.backward(g)
        for c in convs:
            c.weight.grad = None
            c.bias.grad = None
        xr.grad = None

    outs = []
    x = x0
    for i, c in enumerate(convs):
        if i:
            x = F.pixel_shuffle(x, 2)
        x = F.conv2d(x, c.weight, c.bias, padding=1)
        outs.append(x)
        x = F.relu(x)
    cb = torch.ops.aten.convolution_backward

    def chain(mask):
        gg = g
        for i in reversed(range(4)):
            dz = gg * (outs[i] > 0)
            r = cb(dz, ins[i], convs[i].weight, None, [1, 1], [1, 1], [1, 1], False, [0, 0], 1, mask)
            if mask[0]:
                gg = F.pixel_unshuffle(r[0], 2) if i else r[0]
            elif i:
                gg = torch.empty_like(ins[i - 1]).expand(outs[i - 1].shape) if False else torch.zeros_like(outs[i - 1])

    with torch.no_grad():
        t_f = timeit(lambda: forward())
        t_d = timeit(lambda: chain([True, False, False]))
        t_w = timeit(lambda: chain([False, True, False]))
    t_full = timeit(full)
    print(f"head forward {t_f:.3f} ms; forward + backward (autograd) {t_full:.3f} ms; data-gradient chain {t_d:.3f} ms; "
          f"weight gradients (incl. the relu masks again) {t_w:.3f} ms")


if __name__ == "__main__":
    main()
