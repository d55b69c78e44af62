"""BASELINE.json configs[2] and configs[4] on ONE GPU: the eight partition workloads that `bench.py --gpus 8` gives its ranks -- built
with the bench's own origin-box size, `bench.PARTITION_CELL`, so these ARE the scenes the scaling run times (cfg3: independent
sub-scenes, auto_train.py:22-51) -- through the HIP path against the oracle, and the seamless merge + evaluation of
cfg5 (scene/ptgs/merge_partitions.py:95-170, metrics.py:41-113) on device tensors against the reference's numpy sequence.
The collective itself is covered by the world_size-2 gloo tests (tests/test_merge_gloo.py); here world_size is 1 and the eight
partitions are processed one after the other.
"""
import numpy as np
import pytest
import torch

import bench
from helpers import assert_forward_bit_exact, check_backward, hip_forward_raw, oracle_forward

pytestmark = pytest.mark.gpu

WORLD, P, W, H, FOCAL = 8, 500_000, 1920, 1080, 1200.0


def _boxes():
    from based_on_point_cloud_partitions_amd import synthetic as syn

    out = []
    for r in range(WORLD):
        x0, x1, y0, y1 = syn.partition_origin_box(r, WORLD, cell=bench.PARTITION_CELL)
        out.append((x0, y0, x1, y1))
    return out


@pytest.mark.parametrize("rank", range(WORLD))
def test_cfg3_partition_workload_full_parity(rank):
    """Partition `rank` of 8 as `bench.py --gpus 8` builds it (origin boxes of 6.67 x 6.67 in a 4 x 2 grid from (-4, -4): other fp
    magnitudes than cfg2's origin-centred box; expanded by 10 % per side each holds the N = 1 scene's 8 x 8 volume),
    500k Gaussians, 1080p, one camera of its ring: forward bit-exact in both tile-list modes, gradients at the element-wise bar."""
    from based_on_point_cloud_partitions_amd import _lib
    from based_on_point_cloud_partitions_amd import synthetic as syn

    g, centre, _box = syn.make_partition(rank, WORLD, P, seed=0, cell=bench.PARTITION_CELL)
    cam = syn.ring_cameras(16, W, H, FOCAL, center=centre)[(5 * rank + 1) % 16]
    st = oracle_forward(g, cam)
    assert st["num_rendered"] > 2_000_000
    try:
        for cull in (0, 1):
            _lib.set_variant("CULL", cull)
            ctx = hip_forward_raw(g, cam)
            assert_forward_bit_exact(ctx, st)
        dL = np.random.default_rng(100 + rank).standard_normal((3, H, W)).astype(np.float32)
        check_backward(ctx, st, dL)  # default (culled) lists
    finally:
        _lib.set_variant("CULL", 1)


def _partition_blocks(n_per_rank, device):
    """Raw-parameter blocks [N_i, 59] of the eight partitions, with exact-xyz duplicates across neighbours (a Gaussian of the
    overlap margin carried by both partitions, with different other parameters) and inside one partition."""
    from based_on_point_cloud_partitions_amd import merge
    from based_on_point_cloud_partitions_amd import synthetic as syn

    blocks = []
    for r in range(WORLD):
        g, _c, _b = syn.make_partition(r, WORLD, n_per_rank, seed=0, cell=bench.PARTITION_CELL)
        raw_op = torch.logit(g.opacities.clamp(1e-4, 1 - 1e-4))
        blk = merge.pack_block(g.xyz, g.shs[:, :1], g.shs[:, 1:], raw_op, torch.log(g.scales), g.rotations)
        if r > 0:  # rows whose xyz an earlier partition already holds: np.unique keeps the FIRST occurrence
            src = blocks[r - 1]
            k = min(200, src.shape[0], blk.shape[0])
            blk[:k, :3] = src[-k:, :3]
        blk[7, :3] = blk[6, :3]
        blocks.append(blk)
    return [b.to(device) for b in blocks]


def test_cfg5_crop_merge_dedup_on_device_matches_numpy_unique():
    """crop (dynamic boxes, closed intervals, open outer sides) -> concatenate in partition order -> np.unique(axis=0,
    return_index=True) semantics, on device tensors at the full 8 x 500k size, against the reference's numpy lines."""
    from based_on_point_cloud_partitions_amd import merge

    dev = "cuda"
    blocks = _partition_blocks(P, dev)
    boxes = _boxes()
    kept_dev, kept_np = [], []
    for r, blk in enumerate(blocks):
        box = merge.dynamic_bounding_box(r, boxes)
        # the 4 x 2 grid: outer sides open, inner sides closed at the shared edge
        assert (box[0] is None) == (r % 4 == 0) and (box[2] is None) == (r % 4 == 3)
        assert (box[1] is None) == (r < 4) and (box[3] is None) == (r >= 4)
        kept_dev.append(blk[merge.crop_mask(blk[:, :3], box)])
        a = blk.cpu().numpy()
        m = np.ones(a.shape[0], bool)  # extract_point_cloud_mask, merge_partitions.py:95-109
        minx, miny, maxx, maxy = box
        if minx is not None:
            m &= a[:, 0] >= minx
        if maxx is not None:
            m &= a[:, 0] <= maxx
        if miny is not None:
            m &= a[:, 1] >= miny
        if maxy is not None:
            m &= a[:, 1] <= maxy
        kept_np.append(a[m])
    rows = torch.cat(kept_dev, 0)
    allr = np.concatenate(kept_np, 0)
    assert rows.shape[0] == allr.shape[0] > 2_000_000
    got = merge.dedup_first_lexicographic(rows)
    _, idx = np.unique(allr[:, :3], axis=0, return_index=True)  # merge_partitions.py:152-158
    want = allr[idx]
    assert got.shape[0] == want.shape[0] < rows.shape[0]  # duplicates were dropped
    assert np.array_equal(got.cpu().numpy(), want)
    # the single-process path of the collective wrapper is the same function
    one = merge.merge_partitions(blocks[2], merge.dynamic_bounding_box(2, boxes))
    _, i2 = np.unique(kept_np[2][:, :3], axis=0, return_index=True)
    assert np.array_equal(one.cpu().numpy(), kept_np[2][i2])


def test_cfg5_merged_scene_renders_and_scores_like_the_reference_expressions():
    """render (merged scene) -> image_utils.psnr / loss.ssim: the image is bit-exact vs the oracle on the merged parameters, PSNR
    equals 20 log10(1 / sqrt(mse)) (utils/image_utils.py:17-22) and SSIM the oracle's restatement of utils/loss_utils.py:33-63
    (pinned by the reference's own values in tests/golden/ref_loss.npz)."""
    from based_on_point_cloud_partitions_amd import evaluate, merge
    from based_on_point_cloud_partitions_amd import synthetic as syn
    from oracle import oracle

    dev = torch.device("cuda")
    n = 40_000
    blocks = _partition_blocks(n, dev)
    boxes = _boxes()
    rows = torch.cat([b[merge.crop_mask(b[:, :3], merge.dynamic_bounding_box(r, boxes))] for r, b in enumerate(blocks)], 0)
    merged = merge.dedup_first_lexicographic(rows)
    # two test views over the whole 32 x 16 scene
    views = [syn.make_camera(640, 360, 400.0, (12.0, -14.0, 9.0), target=(12.0, 4.0, 0.0)),
             syn.make_camera(640, 360, 400.0, (30.0, 20.0, 7.0), target=(14.0, 4.0, 0.0))]
    bg = torch.tensor([0.1, 0.2, 0.3], device=dev)
    gts = []
    for k, cam in enumerate(views):
        p = merge.unpack_block(merged.cpu())
        a = oracle.activate(p["features_dc"].numpy(), p["features_rest"].numpy(), p["opacity"].numpy(), p["scaling"].numpy(),
                            p["rotation"].numpy())
        g = syn.SyntheticGaussians(p["xyz"], torch.from_numpy(a["scales"]), torch.from_numpy(a["rotations"]),
                                   torch.from_numpy(a["opacities"]), torch.from_numpy(a["shs"]))
        st = oracle_forward(g, cam, bg=(0.1, 0.2, 0.3))
        img = evaluate.render_rows(merged, cam.to(dev), bg)
        assert np.array_equal(img.cpu().numpy().view(np.uint32), st["out_color"].view(np.uint32))
        gen = torch.Generator().manual_seed(k)
        gts.append((img.cpu().clamp(0, 1) + 0.03 * torch.randn(3, 360, 640, generator=gen)).clamp(0, 1).to(dev))
    res = evaluate.seamless_merge_and_evaluate(blocks[0], 0, boxes, [v.to(dev) for v in views], gts, bg)
    assert res["world_size"] == 1 and res["views_scored"] == 2 and len(res["counts"]) == 1
    assert res["counts"][0] - res["merged"].shape[0] in (0, 1)  # the in-partition duplicate, if the crop kept it
    # ... the scores of the full merged scene, computed here view by view
    ps, ss = [], []
    for cam, gt in zip(views, gts):
        img = evaluate.render_rows(merged, cam.to(dev), bg).clamp(0, 1)
        a, b = img.cpu().double().numpy(), gt.cpu().double().numpy()
        want_psnr = 20 * np.log10(1.0 / np.sqrt(((a - b) ** 2).reshape(1, -1).mean(1)))  # utils/image_utils.py:17-22 on [1,3,H,W]
        from based_on_point_cloud_partitions_amd.image_utils import psnr
        from based_on_point_cloud_partitions_amd.loss import ssim
        got_psnr = psnr(img[None], gt[None]).cpu().numpy()
        assert got_psnr.shape == (1, 1) and abs(float(got_psnr[0, 0]) - float(want_psnr[0])) < 2e-5
        _, _, want_ssim, _ = oracle.l1_ssim_loss(img.cpu().numpy(), gt.cpu().numpy(), 0.2, grad=False)
        got_ssim = float(ssim(img, gt))
        assert abs(got_ssim - float(want_ssim)) < 2e-6
        ps.append(float(got_psnr[0, 0]))
        ss.append(got_ssim)
    full = evaluate.seamless_merge_and_evaluate(merged, 0, [(None, None, None, None)], [v.to(dev) for v in views], gts, bg)
    assert abs(full["psnr"] - np.mean(ps)) < 1e-5 and abs(full["ssim"] - np.mean(ss)) < 1e-6
    assert full["merged"].shape == merged.shape and torch.equal(full["merged"], merged)  # merging a merged scene is idempotent
