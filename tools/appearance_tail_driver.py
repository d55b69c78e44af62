"""Runs the HIP part of the appearance network (`up4` block + full-resolution tail, forward + backward) N times at 1080p: the driver behind profiles/r01_appearance_*.  Usage, on
the GPU box: rocprofv3 --kernel-trace --stats --output-format csv -d out -o app -- python3 tools/appearance_tail_driver.py 10"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from based_on_point_cloud_partitions_amd.appearance import appearance_up_tail
H, W = 1080, 1920
torch.manual_seed(0)
x = torch.rand(32, 264, 480, device="cuda", requires_grad=True)  # up3's output
w4 = (0.2 * torch.randn(16, 8, 3, 3, device="cuda")).requires_grad_(True); b4 = torch.zeros(16, device="cuda", requires_grad=True)
w2 = (0.2 * torch.randn(16, 16, 3, 3, device="cuda")).requires_grad_(True); b2 = torch.zeros(16, device="cuda", requires_grad=True)
w3 = (0.2 * torch.randn(3, 16, 3, 3, device="cuda")).requires_grad_(True); b3 = torch.zeros(3, device="cuda", requires_grad=True)
img = torch.rand(3, H, W, device="cuda")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for _ in range(n):
    t, m = appearance_up_tail(x, w4, b4, w2, b2, w3, b3, img)
    t.sum().backward()
torch.cuda.synchronize()
