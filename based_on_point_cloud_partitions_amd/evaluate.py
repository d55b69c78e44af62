"""Config 5 of BASELINE.json on this path: seamless merge of the trained partitions + evaluation of the merged scene.

The reference does this in three scripts over files on disk -- every partition writes its point cloud
(auto_train.py:22-51), `seamless_merge` crops, concatenates and de-duplicates them (scene/ptgs/merge_partitions.py:111-170),
`render.py` + `metrics.py:41-113` render the test views of the merged model and score them (PSNR, SSIM; LPIPS needs
downloaded weights and is not available offline).  Here every rank holds its partition's parameters in HBM, the merge is the
all-gather of `merge.py`, and each rank renders and scores its share of the test views of the merged scene with the HIP
rasterizer, `image_utils.psnr` and `loss.ssim`; the per-view numbers are summed with one tiny all-reduce.
"""
import time

import torch

from . import merge
from .image_utils import psnr
from .loss import ssim
from .rasterizer import GaussianRasterizationSettings
from .renderer import rasterize_gaussian_params_forward


def render_rows(rows, cam, bg, sh_degree=3, scale_modifier=1.0):
    """Image [3,H,W] of a merged [N,59] scene (raw parameters, PLY column order) from one camera: the forward of
    gaussian_renderer/__init__.py:18-100 on the raw-parameter operator (no autograd)."""
    p = merge.unpack_block(rows)
    with torch.no_grad():
        _, color, _, _, _, _ = rasterize_gaussian_params_forward(
            bg, p["xyz"], p["features_dc"], p["features_rest"], p["opacity"], p["scaling"], p["rotation"], scale_modifier,
            cam.world_view_transform, cam.full_proj_transform, cam.tanfovx, cam.tanfovy, cam.image_height, cam.image_width,
            sh_degree, cam.camera_center, False)
    return color


def seamless_merge_and_evaluate(block, index, origin_boxes, views, gts, bg, group=None, sh_degree=3):
    """One rank's part of config 5.  block: this partition's [N_i,59] rows; origin_boxes: [(minx, miny, maxx, maxy)] of all
    partitions (index = this one); views / gts: ALL test cameras (on this device) and their ground-truth images -- every rank scores
    views[rank::world].  Returns dict(merged, counts, world_size, merge_allgather_ms, dedup_ms, psnr, ssim, views_scored) with
    psnr / ssim the means over all test views (metrics.py:91-97)."""
    import torch.distributed as dist

    dev = block.device
    # a world-size-1 group still goes through the collectives (the same RCCL entry points an 8-GPU run uses)
    distributed = dist.is_available() and dist.is_initialized()
    world = dist.get_world_size(group) if distributed else 1
    rank = dist.get_rank(group) if distributed else 0
    sync = (lambda: torch.cuda.synchronize(dev)) if dev.type == "cuda" else (lambda: None)
    box = merge.dynamic_bounding_box(index, origin_boxes)
    kept = block[merge.crop_mask(block[:, :3], box)].contiguous()
    sync()
    t0 = time.perf_counter()
    if distributed:
        rows, counts = merge.all_gather_blocks(kept, group)
    else:
        rows, counts = kept, [kept.shape[0]]
    sync()
    t1 = time.perf_counter()
    merged = merge.dedup_first_lexicographic(rows)
    sync()
    t2 = time.perf_counter()
    acc = torch.zeros(3, dtype=torch.float64, device=dev)  # sum psnr, sum ssim, views
    for v in range(rank, len(views), world):
        img = render_rows(merged, views[v], bg, sh_degree).clamp(0.0, 1.0)  # render.py clamps before saving
        acc[0] += psnr(img[None], gts[v][None]).double().mean()
        acc[1] += ssim(img, gts[v]).double()
        acc[2] += 1
    if distributed:
        if acc.is_cuda and dist.get_backend(group) == "gloo":  # single-GPU rehearsal
            host = acc.cpu()
            dist.all_reduce(host, group=group)
            acc = host.to(dev)
        else:
            dist.all_reduce(acc, group=group)
    n = max(float(acc[2]), 1.0)
    return dict(merged=merged, counts=counts, world_size=world, merge_allgather_ms=1e3 * (t1 - t0), dedup_ms=1e3 * (t2 - t1),
                psnr=float(acc[0]) / n, ssim=float(acc[1]) / n, views_scored=int(acc[2]))
