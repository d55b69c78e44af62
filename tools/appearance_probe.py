"""Times the appearance-decoupling CNN of the reference's training loop (scene/ptgs/appearance_network.py:7-61, train.py:110-117)
as stock torch/MIOpen ops on the GPU: a context figure for SURVEY.md §8f-3 (is the tail worth a HIP kernel?).  Not product code."""
import json
import sys
import time

import torch
import torch.nn as nn
import torch.nn.functional as F


def make_net(cin=67, cout=3):
    def up(ci, co):
        return nn.Sequential(nn.PixelShuffle(2), nn.Conv2d(ci // 4, co, 3, padding=1), nn.ReLU())
    head = nn.Sequential(nn.Conv2d(cin, 256, 3, padding=1), nn.ReLU(), up(256, 128), up(128, 64), up(64, 32), up(32, 16))
    tail = nn.Sequential(nn.Conv2d(16, 16, 3, padding=1), nn.ReLU(), nn.Conv2d(16, cout, 3, padding=1), nn.Sigmoid())
    return head, tail


def timed(fn, n=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) / n * 1e3


def main():
    H, W = 1080, 1920
    dev = "cuda"
    torch.manual_seed(0)
    head, tail = make_net()
    head.to(dev), tail.to(dev)
    emb = torch.randn(64, device=dev, requires_grad=True)
    image = torch.rand(3, H, W, device=dev, requires_grad=True)
    gt = torch.rand(3, H, W, device=dev)
    res = {}

    def fwd(detach_head=False):
        down = F.interpolate(image[None], size=(H // 32, W // 32), mode="bilinear", align_corners=True)[0]
        x = torch.cat([down, emb[None].repeat(H // 32, W // 32, 1).permute(2, 0, 1)], dim=0)[None]
        h = head(x)
        u = F.interpolate(h, size=(H, W), mode="bilinear", align_corners=True)
        m = tail(u).squeeze()
        return m * image

    def step():
        out = fwd()
        (out - gt).abs().mean().backward()

    with torch.no_grad():
        res["forward_ms"] = timed(fwd)
    res["forward_backward_ms"] = timed(step)
    # the two halves
    x = torch.randn(1, 67, H // 32, W // 32, device=dev)
    with torch.no_grad():
        res["head_forward_ms"] = timed(lambda: head(x))
        h = head(x)
        res["upsample_forward_ms"] = timed(lambda: F.interpolate(h, size=(H, W), mode="bilinear", align_corners=True))
        u = F.interpolate(h, size=(H, W), mode="bilinear", align_corners=True)
        res["tail_forward_ms"] = timed(lambda: tail(u))
    u = u.clone().requires_grad_(True)
    res["tail_forward_backward_ms"] = timed(lambda: tail(u).sum().backward())
    try:  # the same tail through the HIP kernels (based_on_point_cloud_partitions_amd/appearance.py)
        import os
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        from based_on_point_cloud_partitions_amd.appearance import appearance_tail
        c2, c3 = tail[0], tail[2]
        hh = h[0].detach().clone().requires_grad_(True)
        img = image.detach()
        with torch.no_grad():
            res["hip_tail_forward_ms"] = timed(lambda: appearance_tail(hh, c2.weight, c2.bias, c3.weight, c3.bias, img))
        res["hip_tail_forward_backward_ms"] = timed(lambda: appearance_tail(hh, c2.weight, c2.bias, c3.weight, c3.bias, img)[0].sum().backward())

        def hip_step():
            down = F.interpolate(image[None], size=(H // 32, W // 32), mode="bilinear", align_corners=True)[0]
            x = torch.cat([down, emb[None].repeat(H // 32, W // 32, 1).permute(2, 0, 1)], dim=0)[None]
            out = appearance_tail(head(x)[0], c2.weight, c2.bias, c3.weight, c3.bias, image)[0]
            (out - gt).abs().mean().backward()

        res["hip_forward_backward_ms"] = timed(hip_step)
    except Exception as e:  # noqa: BLE001 -- the probe still reports the torch figures
        res["hip_error"] = repr(e)
    for tf32 in (False, True):
        torch.backends.cudnn.allow_tf32 = tf32
        res[f"forward_backward_ms_tf32_{tf32}"] = timed(step)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
