"""Per-8x8-block work lists (include/gsrast_mi355x.h, gsr_debug_export_blocks): what the forward blend leaves behind for the
Gaussian-per-lane backward.  No reference counterpart (its backward re-walks whole tile lists, backward.cu:470-482); the
oracle restates the definition with the forward's own predicates (oracle/gs_oracle.c, orc_block_flags).

CPU part: the restatement against a brute-force numpy evaluation of forward.cu:332-357 on a small scene.
GPU part: the lists are exactly the oracle's sets, in tile-list order, and the row-ownership bytes match them.
"""
import numpy as np
import pytest

from helpers import hip_forward_raw, oracle_forward, scene
from oracle import oracle


@pytest.fixture
def block_lists_on():
    """The per-block lists + Gaussian-per-lane backward are an alternative (GSR_BLK, off by default): switch them on for a test."""
    from based_on_point_cloud_partitions_amd import _lib

    _lib.set_variant("BLK", 1)
    yield
    _lib.set_variant("BLK", 0)


def _expected_lists(st):
    """-> {(tile, q): [(gaussian id, position in the tile list), ...]} from the oracle's flags."""
    flags = oracle.block_flags(st)
    rg, pl = st["ranges"], st["point_list"]
    out = {}
    for tile in range(rg.shape[0]):
        r0, r1 = int(rg[tile, 0]), int(rg[tile, 1])
        f = flags[r0:r1]
        for q in range(4):
            ks = np.nonzero((f >> q) & 1)[0]
            out[(tile, q)] = np.stack([pl[r0:r1][ks], ks.astype(np.uint32)], 1) if ks.size else np.zeros((0, 2), np.uint32)
    return out


def test_oracle_block_flags_match_bruteforce_numpy():
    g, cam = scene(400, 96, 80, 60.0, seed=3, scale_lo=0.02, scale_hi=0.2)
    st = oracle_forward(g, cam)
    flags = oracle.block_flags(st)
    W, H = 96, 80
    gx = (W + 15) // 16
    m2, co, pl, rg, nc = st["means2D"], st["conic_opacity"], st["point_list"], st["ranges"], st["n_contrib"]
    want = np.zeros_like(flags)
    for tile in range(rg.shape[0]):
        r0, r1 = int(rg[tile, 0]), int(rg[tile, 1])
        ty, tx = divmod(tile, gx)
        for k in range(r0, r1):
            i = pl[k]
            for ly in range(16):
                for lx in range(16):
                    px, py = tx * 16 + lx, ty * 16 + ly
                    if px >= W or py >= H or (k - r0) >= nc[py, px]:
                        continue  # entries at or behind the pixel's last contributor are never blended
                    dx, dy = np.float32(m2[i, 0] - np.float32(px)), np.float32(m2[i, 1] - np.float32(py))
                    power = np.float32(-0.5) * (co[i, 0] * dx * dx + co[i, 2] * dy * dy) - co[i, 1] * dx * dy
                    if power > 0:
                        continue
                    alpha = min(np.float32(0.99), co[i, 3] * oracle.gs_expf(np.float32(power))[()])
                    if alpha >= np.float32(1.0 / 255.0):
                        want[k] |= 1 << (2 * (ly // 8) + lx // 8)
    # the brute force evaluates `power` without the kernels' fmaf pairs: allow the (measure-zero) borderline entries to differ
    assert np.count_nonzero(want != flags) <= 2
    assert np.count_nonzero(flags) > 100


@pytest.mark.gpu
@pytest.mark.parametrize("P,W,H,focal,kw", [
    (3000, 256, 192, 150.0, dict(scale_lo=0.01, scale_hi=0.08)),
    (20000, 400, 304, 250.0, {}),                      # ragged image edge (400 = 25 tiles, 304 = 19 tiles)
    (1500, 200, 120, 120.0, dict(scale_lo=0.05, scale_hi=0.4)),   # large splats: long lists, saturating pixels
])
def test_hip_block_lists_equal_the_oracle_sets(block_lists_on, P, W, H, focal, kw):
    import torch

    from based_on_point_cloud_partitions_amd import _lib
    from based_on_point_cloud_partitions_amd import rasterizer as rz

    _lib.set_variant("CULL", 0)  # the expected lists below are written in the reference's list positions
    try:
        _check_block_lists(P, W, H, focal, kw)
    finally:
        _lib.set_variant("CULL", 1)


def _check_block_lists(P, W, H, focal, kw):
    import torch

    from based_on_point_cloud_partitions_amd import rasterizer as rz

    g, cam = scene(P, W, H, focal, seed=5, **kw)
    st = oracle_forward(g, cam, bg=(0.2, 0.1, 0.0))
    ctx = hip_forward_raw(g, cam, bg=(0.2, 0.1, 0.0))
    R = ctx["R"]
    assert R == st["num_rendered"] and R > 0
    ex = rz.debug_export_blocks(P, R, W, H, ctx["bb"])
    torch.cuda.synchronize()
    blist = ex["blist"].cpu().numpy().view(np.uint32)
    bcount = ex["bcount"].cpu().numpy().view(np.uint32)
    bmask = ex["bmask"].cpu().numpy().reshape(-1, 4)
    want = _expected_lists(st)
    rg = st["ranges"]
    gx = (W + 15) // 16
    # instance slot of (gaussian, tile): exclusive point offset + position of the tile in the Gaussian's rectangle
    offs = np.concatenate([[0], st["point_offsets"][:-1]]).astype(np.int64)
    want_mask = np.zeros_like(bmask)
    m2, rad = st["means2D"], st["radii"]
    gy = (H + 15) // 16
    total = 0
    for (tile, q), w in want.items():
        r0, n = int(rg[tile, 0]), int(rg[tile, 1]) - int(rg[tile, 0])
        c = int(bcount[4 * tile + q])
        got = blist[4 * r0 + q * n: 4 * r0 + q * n + c]
        assert c == w.shape[0], (tile, q, c, w.shape[0])
        assert np.array_equal(got, w), (tile, q)
        total += c
        ty, tx = divmod(tile, gx)
        for i in w[:, 0]:
            r = float(rad[i])
            mnx = min(gx, max(0, int(np.float32(np.float32(m2[i, 0]) - np.float32(r)) / np.float32(16))))
            mxx = min(gx, max(0, int(np.float32(np.float32(m2[i, 0]) + np.float32(r) + np.float32(15)) / np.float32(16))))
            mny = min(gy, max(0, int(np.float32(np.float32(m2[i, 1]) - np.float32(r)) / np.float32(16))))
            want_mask[offs[i] + (ty - mny) * (mxx - mnx) + (tx - mnx), q] = 1
    assert total > 0
    assert np.array_equal(bmask, want_mask)


@pytest.mark.gpu
@pytest.mark.parametrize("P,W,H,focal,kw", [
    (6000, 256, 192, 150.0, dict(scale_lo=0.01, scale_hi=0.08)),
    (20000, 400, 304, 250.0, {}),
    (1500, 200, 120, 120.0, dict(scale_lo=0.05, scale_hi=0.4)),
    (64, 13, 143, 40.0, dict(scale_lo=0.02, scale_hi=1.0)),          # one tile column, ragged
    (60000, 2100, 2100, 1500.0, dict(scale_lo=0.004, scale_hi=0.03)),  # 17,424 tiles, mostly short lists
])
def test_gaussian_per_lane_backward_matches_the_oracle(block_lists_on, P, W, H, focal, kw):
    """render_backward_skew.hip (GSR_BLK=1) against the oracle's gradients, same bar as the default kernel."""
    from helpers import assert_forward_bit_exact, assert_grads_close, hip_backward_raw

    g, cam = scene(P, W, H, focal, seed=11, **kw)
    bg = (0.3, 0.0, 0.6)
    st = oracle_forward(g, cam, bg=bg)
    ctx = hip_forward_raw(g, cam, bg=bg)
    assert_forward_bit_exact(ctx, st)  # the list-building forward renders the same image, bit for bit
    dL = np.random.default_rng(2).standard_normal((3, H, W)).astype(np.float32)
    assert_grads_close(hip_backward_raw(ctx, dL), oracle.backward(st, dL))


@pytest.mark.gpu
def test_gaussian_per_lane_backward_config2_size(block_lists_on):
    """Full 500k-Gaussian 1080p frame: every gradient row is written exactly once (the result matches the default kernel's)."""
    import torch

    from based_on_point_cloud_partitions_amd import _lib
    from helpers import assert_grads_close, hip_backward_raw

    g, cam = scene(500_000, 1920, 1080, 1200.0, seed=0, n_cams=16)
    dL = np.random.default_rng(3).standard_normal((3, 1080, 1920)).astype(np.float32)
    got = hip_backward_raw(hip_forward_raw(g, cam), dL)
    _lib.set_variant("BLK", 0)
    want = hip_backward_raw(hip_forward_raw(g, cam), dL)
    _lib.set_variant("BLK", 1)
    assert_grads_close(got, want, tol=2e-5)
    torch.cuda.empty_cache()
