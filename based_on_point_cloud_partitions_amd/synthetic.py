"""Synthetic sub-scenes and cameras for parity tests, smoke() and bench.py.

Workload definition: BASELINE.md §3 / SURVEY.md §8(d).  The camera matrices
follow the reference's conventions exactly (scene/cameras.py:103-112,
utils/graphics_utils.py:37-71): `viewmatrix` is the transposed world-to-camera
matrix, `projmatrix` the transposed full projection, both flattened row-major,
so element [4*c + r] is row r / column c of the mathematical matrix.
Nothing here reads /root/reference; tests/golden pins it against the
reference's own graphics_utils.
"""
import math
from typing import NamedTuple

import numpy as np
import torch


class SyntheticCamera(NamedTuple):
    image_width: int
    image_height: int
    FoVx: float
    FoVy: float
    tanfovx: float
    tanfovy: float
    world_view_transform: torch.Tensor  # [4,4] = W2C^T
    full_proj_transform: torch.Tensor  # [4,4] = (P @ W2C)^T
    camera_center: torch.Tensor  # [3]

    def to(self, device):
        return self._replace(world_view_transform=self.world_view_transform.to(device),
                             full_proj_transform=self.full_proj_transform.to(device),
                             camera_center=self.camera_center.to(device))


def focal2fov(focal, pixels):  # utils/graphics_utils.py:76-77
    return 2 * math.atan(pixels / (2 * focal))


def world2view(R, t):
    """utils/graphics_utils.py:45-56 with translate=0, scale=1 (R is the camera-to-world rotation)."""
    Rt = np.zeros((4, 4))
    Rt[:3, :3] = R.transpose()
    Rt[:3, 3] = t.flatten()
    Rt[3, 3] = 1.0
    C2W = np.linalg.inv(Rt)
    Rt = np.linalg.inv(C2W)
    return np.float32(Rt)


def projection_matrix(znear, zfar, fovX, fovY):
    """utils/graphics_utils.py:59-77."""
    tanHalfFovY = math.tan((fovY / 2))
    tanHalfFovX = math.tan((fovX / 2))
    top = tanHalfFovY * znear
    bottom = -top
    right = tanHalfFovX * znear
    left = -right
    P = torch.zeros(4, 4)
    z_sign = 1.0
    P[0, 0] = 2.0 * znear / (right - left)
    P[1, 1] = 2.0 * znear / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = z_sign
    P[2, 2] = z_sign * zfar / (zfar - znear)
    P[2, 3] = -(zfar * znear) / (zfar - znear)
    return P


def look_at_RT(cam_pos, target=(0.0, 0.0, 0.0), up=(0.0, 0.0, 1.0)):
    """COLMAP-style pose (x right, y down, z forward). Returns (R = C2W rotation, T = W2C translation)."""
    c = np.asarray(cam_pos, dtype=np.float64)
    f = np.asarray(target, dtype=np.float64) - c
    f /= np.linalg.norm(f)
    right = np.cross(f, np.asarray(up, dtype=np.float64))
    right /= np.linalg.norm(right)
    down = np.cross(f, right)
    Rw2c = np.stack([right, down, f], axis=0)
    T = -Rw2c @ c
    return Rw2c.T.copy(), T


def make_camera(W, H, focal, cam_pos, target=(0.0, 0.0, 0.0), znear=0.01, zfar=100.0):
    """Builds the matrices the way scene/cameras.py:103-112 does."""
    R, T = look_at_RT(cam_pos, target)
    FoVx, FoVy = focal2fov(focal, W), focal2fov(focal, H)
    wvt = torch.tensor(world2view(R, T)).transpose(0, 1)
    proj = projection_matrix(znear=znear, zfar=zfar, fovX=FoVx, fovY=FoVy).transpose(0, 1)
    full = (wvt.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0)
    center = wvt.inverse()[3, :3]
    return SyntheticCamera(W, H, FoVx, FoVy, math.tan(FoVx * 0.5), math.tan(FoVy * 0.5),
                           wvt.contiguous(), full.contiguous(), center.contiguous())


def ring_cameras(n, W, H, focal, radius=6.0, height=2.0, center=(0.0, 0.0, 0.0)):
    """n cameras on a circle of `radius` at `height`, looking at `center`."""
    cams = []
    for k in range(n):
        th = 2.0 * math.pi * k / n
        pos = (center[0] + radius * math.cos(th), center[1] + radius * math.sin(th), center[2] + height)
        cams.append(make_camera(W, H, focal, pos, target=center))
    return cams


class SyntheticGaussians(NamedTuple):
    xyz: torch.Tensor  # [P,3]
    scales: torch.Tensor  # [P,3] (activated: exp)
    rotations: torch.Tensor  # [P,4] unit quaternions (r,x,y,z)
    opacities: torch.Tensor  # [P,1] (activated: sigmoid)
    shs: torch.Tensor  # [P,16,3]

    def to(self, device):
        return SyntheticGaussians(*[t.to(device) for t in self])


def make_gaussians(P, seed=0, box_min=(-4.0, -4.0, -1.0), box_max=(4.0, 4.0, 1.0), sh_coeffs=16,
                   scale_lo=0.004, scale_hi=0.04):
    """Random-init Gaussians with the distributions of BASELINE.md §3 (CPU generator, fixed seed)."""
    g = torch.Generator().manual_seed(seed)
    lo = torch.tensor(box_min, dtype=torch.float32)
    hi = torch.tensor(box_max, dtype=torch.float32)
    xyz = lo + (hi - lo) * torch.rand(P, 3, generator=g)
    log_s = math.log(scale_lo) + (math.log(scale_hi) - math.log(scale_lo)) * torch.rand(P, 3, generator=g)
    scales = torch.exp(log_s)
    q = torch.randn(P, 4, generator=g)
    rotations = q / q.norm(dim=1, keepdim=True).clamp_min(1e-12)
    opacities = torch.sigmoid(1.5 * torch.randn(P, 1, generator=g))
    f_dc = 0.5 * torch.randn(P, 1, 3, generator=g)
    f_rest = 0.1 * torch.randn(P, sh_coeffs - 1, 3, generator=g)
    shs = torch.cat((f_dc, f_rest), dim=1).contiguous()
    return SyntheticGaussians(xyz.contiguous(), scales.contiguous(), rotations.contiguous(), opacities.contiguous(), shs)


def partition_origin_box(rank, world_size, cell=(8.0, 8.0), grid_cols=4):
    """origin_box of partition `rank` in a grid of cells (the reference's balanced split yields
    axis-aligned XY boxes: scene/ptgs/density_partition.py:9-99).  Returns (xmin, xmax, ymin, ymax)."""
    cols = min(grid_cols, world_size)
    cx, cy = rank % cols, rank // cols
    x0, y0 = cx * cell[0] - 4.0, cy * cell[1] - 4.0
    return (x0, x0 + cell[0], y0, y0 + cell[1])


def make_partition(rank, world_size, P, seed=0, overlap=0.1, cell=(8.0, 8.0)):
    """Gaussians + camera-ring centre of one partition: the box is expanded by `overlap` on every side
    (scene/ptgs/shen_partation_densify.py:182-252 expands partitions by 10 %).  `cell` = the XY size of an origin box."""
    x0, x1, y0, y1 = partition_origin_box(rank, world_size, cell=cell)
    ex, ey = overlap * (x1 - x0), overlap * (y1 - y0)
    g = make_gaussians(P, seed=seed + 1000 * rank, box_min=(x0 - ex, y0 - ey, -1.0), box_max=(x1 + ex, y1 + ey, 1.0))
    centre = (0.5 * (x0 + x1), 0.5 * (y0 + y1), 0.0)
    return g, centre, (x0, x1, y0, y1)


# named workloads of BASELINE.json `configs`
CONFIGS = {
    "cfg1": dict(P=10_000, W=256, H=256, focal=160.0, cameras=4),
    "cfg2": dict(P=500_000, W=1920, H=1080, focal=1200.0, cameras=16),
    "cfg4": dict(P=2_000_000, W=1920, H=1080, focal=1200.0, cameras=16),
    # not a BASELINE.json configuration: cfg2's box at twice / three times the density (mean reference tile list ~930 / ~1400), used to
    # place the list-length threshold between the two work splits of the blend backward
    "dense1m": dict(P=1_000_000, W=1920, H=1080, focal=1200.0, cameras=16),
    "dense1m5": dict(P=1_500_000, W=1920, H=1080, focal=1200.0, cameras=16),
    # ... and two smaller images of the same scene (2,500 / 920 tiles), used to place the tile-count threshold of that choice
    "mid800": dict(P=300_000, W=800, H=800, focal=600.0, cameras=8),
    "mid640": dict(P=150_000, W=640, H=360, focal=400.0, cameras=8),
    # ... and cfg2's scene at four times the pixels (32,400 tiles, the same list lengths): many more tiles than wave slots
    "uhd500k": dict(P=500_000, W=3840, H=2160, focal=2400.0, cameras=8),
}


def make_appearance_model(n_cameras=16, seed=0, device="cpu"):
    """Random-init stand-in for the appearance state train.py:110-111 consumes: `.appearance_network` with the layer shapes and
    attribute names of the reference's AppearanceNetwork(3 + 64, 3) (scene/ptgs/appearance_network.py:31-61: conv1 67->256, four
    pixel-shuffle blocks 256->128->64->32->16, bilinear resize, conv2 16->16, conv3 16->3, sigmoid) and
    `.get_apperance_embedding(idx)` over an [n_cameras, 64] table.  For benchmarks and tests (no checkpoints offline)."""
    nn = torch.nn

    class Up(nn.Module):
        def __init__(self, ci, co):
            super().__init__()
            self.pixel_shuffle, self.conv, self.relu = nn.PixelShuffle(2), nn.Conv2d(ci // 4, co, 3, padding=1), nn.ReLU()

        def forward(self, x):
            return self.relu(self.conv(self.pixel_shuffle(x)))

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.conv1 = nn.Conv2d(67, 256, 3, padding=1)
            self.up1, self.up2, self.up3, self.up4 = Up(256, 128), Up(128, 64), Up(64, 32), Up(32, 16)
            self.conv2, self.conv3 = nn.Conv2d(16, 16, 3, padding=1), nn.Conv2d(16, 3, 3, padding=1)
            self.relu, self.sigmoid = nn.ReLU(), nn.Sigmoid()

        def forward(self, x, H, W):
            x = self.up4(self.up3(self.up2(self.up1(self.relu(self.conv1(x))))))
            x = nn.functional.interpolate(x, size=(H, W), mode="bilinear", align_corners=True)
            return self.sigmoid(self.conv3(self.relu(self.conv2(x))))

    class Model:
        def __init__(self):
            torch.manual_seed(seed)
            self.appearance_network = Net().to(device)
            self._appearance_embeddings = nn.Parameter((0.5 * torch.randn(n_cameras, 64)).to(device))

        def get_apperance_embedding(self, idx):
            return self._appearance_embeddings[idx]

    return Model()


def decouple_appearance_torch(image, model, view_idx):
    """The data flow of scene/ptgs/appearance_network.py:7-17 with every layer a stock torch op: the baseline the HIP tail
    (appearance.py) is timed and tested against."""
    H, W = image.size(1), image.size(2)
    emb = model.get_apperance_embedding(view_idx)
    down = torch.nn.functional.interpolate(image[None], size=(H // 32, W // 32), mode="bilinear", align_corners=True)[0]
    x = torch.cat([down, emb[None].repeat(H // 32, W // 32, 1).permute(2, 0, 1)], dim=0)[None]
    mapping = model.appearance_network(x, H, W).squeeze()
    return mapping * image, mapping
