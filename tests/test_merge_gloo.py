"""world_size-2 gloo test of the only collective on the path: the partition merge."""
import os

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from based_on_point_cloud_partitions_amd import merge


def _blocks():
    rng = np.random.default_rng(0)
    a = rng.uniform(-1, 1, (700, merge.ROW)).astype(np.float32)
    b = rng.uniform(-1, 1, (500, merge.ROW)).astype(np.float32)
    a[:, 0] = rng.uniform(-4, 0.5, 700)   # partition 0 covers x <= 0 plus a margin
    b[:, 0] = rng.uniform(-0.5, 4, 500)   # partition 1 covers x >= 0 plus a margin
    b[:40, :3] = a[100:140, :3]           # Gaussians in the overlap that both partitions carry (same xyz, other params)
    a[5, 0] = 0.0                         # exactly on the shared boundary: kept by both (closed intervals), deduplicated
    b[41, :3] = a[5, :3]
    a[7, :3] = a[6, :3]                   # duplicate inside one partition
    return a, b


BOXES = [(None, None, 0.0, None), (0.0, None, None, None)]  # shared edge x = 0; outer sides open (no neighbour)


def _reference(a, b):
    """The reference's numpy sequence (scene/ptgs/merge_partitions.py:95-109,143-158)."""
    kept = []
    for blk, (minx, miny, maxx, maxy) in zip((a, b), BOXES):
        m = np.ones(len(blk), bool)
        if minx is not None:
            m &= blk[:, 0] >= minx
        if maxx is not None:
            m &= blk[:, 0] <= maxx
        kept.append(blk[m])
    allr = np.concatenate(kept, 0)
    pts, idx = np.unique(allr[:, :3], axis=0, return_index=True)
    return allr[idx]


def _worker(rank, world, port, out_dir):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    a, b = _blocks()
    mine = torch.from_numpy((a, b)[rank])
    merged = merge.merge_partitions(mine, BOXES[rank])
    np.save(os.path.join(out_dir, f"merged_{rank}.npy"), merged.numpy())
    dist.destroy_process_group()


def test_merge_two_ranks_matches_numpy_unique(tmp_path):
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    want = _reference(*_blocks())
    for r in range(2):
        got = np.load(tmp_path / f"merged_{r}.npy")
        assert got.shape == want.shape
        assert np.array_equal(got, want)


def _worker_eval(rank, world, port, out_dir):
    import torch.distributed as dist

    from based_on_point_cloud_partitions_amd import evaluate

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    a, b = _blocks()
    origin = [(-4.0, -1.0, 0.0, 1.0), (0.0, -1.0, 4.0, 1.0)]  # two boxes sharing the edge x = 0
    res = evaluate.seamless_merge_and_evaluate(torch.from_numpy((a, b)[rank]), rank, origin, [], [], torch.zeros(3))
    assert res["world_size"] == world and len(res["counts"]) == world and res["views_scored"] == 0
    np.save(os.path.join(out_dir, f"eval_{rank}.npy"), res["merged"].numpy())
    np.save(os.path.join(out_dir, f"counts_{rank}.npy"), np.asarray(res["counts"]))
    dist.destroy_process_group()


def test_config5_driver_two_ranks(tmp_path):
    """evaluate.seamless_merge_and_evaluate (crop by the dynamic boxes -> all_gather_into_tensor -> dedup) over gloo, world 2:
    both ranks end with the reference's merged scene and report the world size and per-rank counts they saw."""
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_worker_eval, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    want = _reference(*_blocks())
    a, b = _blocks()
    for r in range(2):
        assert np.array_equal(np.load(tmp_path / f"eval_{r}.npy"), want)
        assert np.load(tmp_path / f"counts_{r}.npy").tolist() == [int((a[:, 0] <= 0).sum()), int((b[:, 0] >= 0).sum())]


def test_dynamic_bounding_box_follows_the_neighbours():
    """scene/ptgs/merge_partitions.py:10-57: a side is closed only where a partition sharing a boundary SEGMENT has its opposite
    side on it; corners do not count (shen_partition.py:165-186: the intersection must be a LineString of non-zero length)."""
    from based_on_point_cloud_partitions_amd import synthetic as syn

    boxes = []
    for r in range(8):
        x0, x1, y0, y1 = syn.partition_origin_box(r, 8)
        boxes.append((x0, y0, x1, y1))
    assert merge.dynamic_bounding_box(0, boxes) == (None, None, 4.0, 4.0)
    assert merge.dynamic_bounding_box(1, boxes) == (4.0, None, 12.0, 4.0)
    assert merge.dynamic_bounding_box(7, boxes) == (20.0, 4.0, None, None)
    assert merge.dynamic_bounding_box(5, boxes) == (4.0, 4.0, 12.0, None)
    # a lone partition has no closed side; two boxes touching in a corner only are not neighbours
    assert merge.dynamic_bounding_box(0, [(0.0, 0.0, 1.0, 1.0)]) == (None, None, None, None)
    assert merge.dynamic_bounding_box(0, [(0.0, 0.0, 1.0, 1.0), (1.0, 1.0, 2.0, 2.0)]) == (None, None, None, None)
    # a balanced binary split (density_partition.py): unequal boxes, the big one has two neighbours on its right side
    split = [(0.0, 0.0, 2.0, 4.0), (2.0, 0.0, 5.0, 1.5), (2.0, 1.5, 5.0, 4.0)]
    assert merge.dynamic_bounding_box(0, split) == (None, None, 2.0, None)
    assert merge.dynamic_bounding_box(1, split) == (2.0, None, None, 1.5)
    assert merge.dynamic_bounding_box(2, split) == (2.0, 1.5, None, None)


def test_unpack_block_inverts_pack_block():
    N = 5
    xyz, f_dc, f_rest = torch.rand(N, 3), torch.rand(N, 1, 3), torch.rand(N, 15, 3)
    op, sc, rot = torch.rand(N, 1), torch.rand(N, 3), torch.rand(N, 4)
    p = merge.unpack_block(merge.pack_block(xyz, f_dc, f_rest, op, sc, rot))
    for k, v in dict(xyz=xyz, features_dc=f_dc, features_rest=f_rest, opacity=op, scaling=sc, rotation=rot).items():
        assert torch.equal(p[k], v), k


def test_dedup_and_crop_single_process():
    a, b = _blocks()
    rows = torch.from_numpy(np.concatenate([a, b], 0))
    got = merge.dedup_first_lexicographic(rows).numpy()
    allr = np.concatenate([a, b], 0)
    _, idx = np.unique(allr[:, :3], axis=0, return_index=True)
    assert np.array_equal(got, allr[idx])
    assert merge.dedup_first_lexicographic(rows[:0]).shape == (0, merge.ROW)
    m = merge.crop_mask(torch.tensor([[0.0, 0.0, 0.0], [1.0, 5.0, 0.0], [-1.0, 0.0, 0.0]]), (0.0, None, 1.0, 2.0))
    assert m.tolist() == [True, False, False]


def test_pack_block_matches_ply_column_order():
    N = 4
    xyz, f_dc, f_rest = torch.rand(N, 3), torch.rand(N, 1, 3), torch.rand(N, 15, 3)
    op, sc, rot = torch.rand(N, 1), torch.rand(N, 3), torch.rand(N, 4)
    blk = merge.pack_block(xyz, f_dc, f_rest, op, sc, rot)
    assert blk.shape == (N, merge.ROW)
    # scene/gaussian_model.py:208-209: transpose(1, 2).flatten(start_dim=1)
    assert torch.equal(blk[:, 6:51], f_rest.transpose(1, 2).flatten(start_dim=1))
    assert torch.equal(blk[:, 51:52], op) and torch.equal(blk[:, 55:], rot)


def test_partition_layout_of_the_multi_gpu_benchmark():
    """bench.py --gpus N: rank r owns origin_box r of a 4-column grid, Gaussians cover the box + 10 % margin
    (scene/ptgs/shen_partation_densify.py:182-252), and cropping every partition to its box keeps disjoint sets."""
    from based_on_point_cloud_partitions_amd import synthetic as syn

    world = 8
    boxes = [syn.partition_origin_box(r, world) for r in range(world)]
    assert boxes[0] == (-4.0, 4.0, -4.0, 4.0) and boxes[3][0] == 20.0 and boxes[4] == (-4.0, 4.0, 4.0, 12.0)
    kept = []
    for r in range(world):
        g, centre, box = syn.make_partition(r, world, 2000, seed=0)
        x0, x1, y0, y1 = box
        assert abs(centre[0] - 0.5 * (x0 + x1)) < 1e-6 and g.xyz[:, 0].min() < x0 and g.xyz[:, 0].max() > x1  # overlap margin
        m = merge.crop_mask(g.xyz, (x0, y0, x1, y1))
        assert 0.6 < m.float().mean() < 0.8  # (1/1.2)^2 of the expanded box
        kept.append(g.xyz[m])
    allxyz = torch.cat(kept)
    assert torch.unique(allxyz, dim=0).shape[0] == allxyz.shape[0]


def test_bench_launcher_starts_the_ranks_as_a_child_process(monkeypatch):
    """bench.py --gpus N without RANK in the environment: torch.distributed.run as a child, same arguments, 127.0.0.1 rendezvous,
    return code relayed; with RANK set (a launcher started us) or N = 1 nothing is spawned."""
    import subprocess
    import sys

    import bench

    seen = {}

    class Done:
        returncode = 7

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return Done()

    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.delenv("RANK", raising=False)
    with pytest.raises(SystemExit) as e:
        bench.launch_ranks_if_needed(["--gpus", "4", "--steps", "3"])
    assert e.value.code == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "3"]
    assert cmd[-5].endswith("bench.py") and seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    seen.clear()
    bench.launch_ranks_if_needed(["--gpus", "1"])
    bench.launch_ranks_if_needed(["--steps", "3"])
    monkeypatch.setenv("RANK", "0")
    bench.launch_ranks_if_needed(["--gpus=4"])
    assert not seen


def test_dedup_matches_numpy_unique_on_adversarial_coordinates():
    """np.unique(points, axis=0, return_index=True) (scene/ptgs/merge_partitions.py:152-158) against merge.dedup_first_lexicographic on
    coordinates chosen to break an integer-key sort: +-0.0 (equal under float comparison), denormals, +-inf, huge / tiny magnitudes,
    many exact duplicates in every prefix of (x, y, z), and rows that differ in z only (the least significant key)."""
    from hypothesis import given, settings
    from hypothesis import strategies as st

    special = [0.0, -0.0, 1.0, -1.0, 1e-45, -1e-45, 1.17549435e-38, 3.4e38, -3.4e38, float("inf"), float("-inf"), 0.5, 2.0 ** -126, 1.0000001]
    coord = st.one_of(st.sampled_from(special), st.floats(width=32, allow_nan=False, allow_infinity=True),
                      st.integers(-3, 3).map(float))

    @settings(max_examples=150, deadline=None)
    @given(st.lists(st.tuples(coord, coord, coord), min_size=0, max_size=60), st.integers(0, 2 ** 31 - 1))
    def check(points, seed):
        n = len(points)
        rows = np.random.default_rng(seed).standard_normal((n, merge.ROW)).astype(np.float32)
        if n:
            rows[:, :3] = np.asarray(points, np.float32).reshape(n, 3)
        got = merge.dedup_first_lexicographic(torch.from_numpy(rows.copy())).numpy()
        if n == 0:
            assert got.shape == (0, merge.ROW)
            return
        _, idx = np.unique(rows[:, :3], axis=0, return_index=True)
        want = rows[idx]
        assert got.shape == want.shape
        assert np.array_equal(got[:, 3:], want[:, 3:])           # the same rows survive, in the same order
        assert np.array_equal(got[:, :3], want[:, :3])           # (+0.0 == -0.0 here, as for np.unique)

    check()


def test_dedup_keeps_every_row_with_a_nan_coordinate():
    """np.unique(axis=0) compares fields as floats, so a row with a NaN coordinate equals no other and every one survives
    (scene/ptgs/merge_partitions.py:152-158); the integer-key sort must not merge NaNs of identical bits.  The rows without NaN
    are deduplicated exactly as before."""
    nan = float("nan")
    pts = np.array([[nan, 1, 2], [0, 1, 2], [nan, 1, 2], [0, 1, 2], [1, nan, 0], [1, nan, 0], [3, 3, nan], [-1, 0, 0]], np.float32)
    rows = np.random.default_rng(0).standard_normal((len(pts), merge.ROW)).astype(np.float32)
    rows[:, :3] = pts
    got = merge.dedup_first_lexicographic(torch.from_numpy(rows.copy())).numpy()
    _, idx = np.unique(pts, axis=0, return_index=True)
    assert got.shape[0] == len(idx) == 7
    has_nan = np.isnan(got[:, :3]).any(axis=1)
    assert has_nan.sum() == 5
    assert sorted(got[has_nan][:, 3].tolist()) == sorted(rows[np.isnan(pts).any(axis=1)][:, 3].tolist())
    clean = rows[~np.isnan(pts).any(axis=1)]
    _, ci = np.unique(clean[:, :3], axis=0, return_index=True)
    assert np.array_equal(got[~has_nan], clean[ci])


def test_bench_per_rank_report_names_the_slowest_rank():
    """The N > 1 bench line carries every rank's own figures (the reference's launcher reports per partition, auto_train.py:22-51):
    from the all-gathered rows rank 0 derives who set the MAX, the start skew and whether that rank had more work."""
    import bench

    from based_on_point_cloud_partitions_amd import _lib

    rb, rf = float(_lib.STAGES.index("render_backward")), float(_lib.STAGES.index("render_forward"))
    rows = [[0.0200, 0.10, 100.0000, 5e5, 3.70e6, 2.30e6, 450, 2000, 0.32, 0.32, 0.70, 0.05, 0.69, 0, rb],
            [0.0230, 0.30, 100.0004, 5e5, 3.90e6, 2.50e6, 470, 2100, 0.36, 0.36, 0.75, 0.01, 0.70, 1, rb],
            [0.0190, 0.00, 100.0001, 5e5, 3.60e6, 2.20e6, 440, 1900, 0.18, 0.31, 0.69, 0.09, 0.68, 2, rf]]
    r = bench.per_rank_report(rows, 20, "render_backward")
    assert r["slowest_rank"] == 1 and r["fastest_rank"] == 2 and abs(r["skew_ms"] - 0.4) < 1e-6
    assert [p["rank"] for p in r["per_rank"]] == [0, 1, 2] and r["per_rank"][1]["ms_per_step"] == 1.15
    assert r["per_rank"][2]["start_offset_ms"] == 0.1 and r["per_rank"][0]["start_offset_ms"] == 0.0
    assert abs(r["slowest_vs_fastest"]["instances_listed"] - 2.5 / 2.2) < 1e-3
    assert abs(r["sum_of_rank_rates_it_per_s"] - (1000.0 + 20 / 0.023 + 20 / 0.019)) < 0.05
    assert [p["dominant_stage"] for p in r["per_rank"]] == ["render_backward", "render_backward", "render_forward"]


def test_bench_partitions_fill_the_single_gpu_scene_volume():
    """bench.py --gpus N: a rank's partition (origin box expanded by the reference's 10 % per side) is as large as the N = 1 scene, so the
    per-GPU work is the same distribution at every N (with 8-wide origin boxes a rank had 15 % fewer instances)."""
    import bench
    from based_on_point_cloud_partitions_amd import synthetic as syn

    assert abs(bench.PARTITION_CELL[0] * 1.2 - 8.0) < 1e-9 and abs(bench.PARTITION_CELL[1] * 1.2 - 8.0) < 1e-9
    for world in (2, 8):
        for r in range(world):
            g, centre, box = syn.make_partition(r, world, 4000, seed=0, cell=bench.PARTITION_CELL)
            lo, hi = g.xyz.min(0).values, g.xyz.max(0).values
            assert torch.all(hi[:2] - lo[:2] < 8.0 + 1e-4) and torch.all(hi[:2] - lo[:2] > 7.9)
            assert abs(float(0.5 * (lo[0] + hi[0])) - centre[0]) < 0.05 and abs(float(0.5 * (lo[1] + hi[1])) - centre[1]) < 0.05
            assert box == syn.partition_origin_box(r, world, cell=bench.PARTITION_CELL)


def test_bench_price_list_prices_three_register_fmas_and_dpp_adds():
    """bench.price_list_cycles: the class counters call a DPP add an add and a three-register fma an fma; their shares come from the static
    listing of the template instance that ran (profiles/blend_kernels_issue_shares.json), and the committed PMC profile names that instance."""
    import json

    import bench

    v = {"sq_insts_valu": 100.0, "sq_insts_valu_add_f32": 30.0, "sq_insts_valu_mul_f32": 30.0, "sq_insts_valu_fma_f32": 20.0,
         "sq_insts_valu_int32": 0.0, "sq_insts_valu_trans_f32": 4.0, "kernel_name": "k<2>"}
    plain = bench.price_list_cycles(v)  # no shares: everything counted as add / mul / fma is plain
    assert abs(plain - (2.4 * 80 + 8.5 * 4 + 4.3 * 16) / 100) < 1e-9
    shares = {"three_register_share_of_fp32_fma": {"k<2>": 0.5}, "dpp_share_of_fp32_add": {"k<2>": 1.0}}
    priced = bench.price_list_cycles(v, shares)
    assert abs(priced - (2.4 * (30 + 10) + 4.2 * 30 + 4.0 * 10 + 8.5 * 4 + 4.3 * 16) / 100) < 1e-9
    assert bench.price_list_cycles({"sq_insts_valu": 1.0}) is None
    # the committed files fit together: the profile names the instance, the shares file knows it
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pmc = json.load(open(os.path.join(root, "profiles", "pmc_traffic.json")))
    sh = bench.issue_shares()
    for stage in ("render_backward", "render_forward"):
        name = pmc[stage]["kernel_name"]
        assert name in sh["three_register_share_of_fp32_fma"] and name in sh["dpp_share_of_fp32_add"]
        assert bench.price_list_cycles(pmc[stage], sh) >= bench.price_list_cycles(pmc[stage])
