"""3DGS PLY round trip for the merge block (format of scene/gaussian_model.py:189-278)."""
import numpy as np
import torch

from based_on_point_cloud_partitions_amd import merge, ply_io


def test_round_trip_and_header(tmp_path):
    blk = np.random.default_rng(0).normal(size=(257, merge.ROW)).astype(np.float32)
    p = tmp_path / "point_cloud.ply"
    ply_io.save_ply(str(p), torch.from_numpy(blk))
    raw = open(p, "rb").read()
    head, body = raw.split(b"end_header\n", 1)
    lines = head.decode().splitlines()
    assert lines[:3] == ["ply", "format binary_little_endian 1.0", "element vertex 257"]
    names = [l.split()[2] for l in lines[3:]]
    # construct_list_of_attributes order, gaussian_model.py:189-201
    assert names[:9] == ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"]
    assert names[9:54] == [f"f_rest_{i}" for i in range(45)] and names[54:] == ["opacity", "scale_0", "scale_1", "scale_2", "rot_0", "rot_1", "rot_2", "rot_3"]
    assert len(body) == 257 * 62 * 4
    assert np.array_equal(ply_io.load_ply(str(p)), blk)
    rows = np.frombuffer(body, "<f4").reshape(257, 62)
    assert np.all(rows[:, 3:6] == 0)  # normals


def test_loader_sorts_properties_by_suffix(tmp_path):
    # the reference looks properties up by name and sorts f_rest_/scale_/rot by suffix (gaussian_model.py:248-263)
    blk = np.random.default_rng(1).normal(size=(5, merge.ROW)).astype(np.float32)
    names = ply_io.attribute_names()
    perm = np.random.default_rng(2).permutation(len(names))
    full = np.zeros((5, 62), "<f4")
    full[:, :3] = blk[:, :3]
    full[:, 6:] = blk[:, 3:]
    header = "ply\nformat binary_little_endian 1.0\nelement vertex 5\n" + "".join(f"property float {names[i]}\n" for i in perm) + "end_header\n"
    p = tmp_path / "shuffled.ply"
    with open(p, "wb") as f:
        f.write(header.encode())
        f.write(np.ascontiguousarray(full[:, perm]).tobytes())
    assert np.array_equal(ply_io.load_ply(str(p)), blk)


def test_pack_save_load_merge_pipeline(tmp_path):
    N = 50
    g = torch.Generator().manual_seed(3)
    parts = dict(xyz=torch.rand(N, 3, generator=g), f_dc=torch.rand(N, 1, 3, generator=g), f_rest=torch.rand(N, 15, 3, generator=g),
                 op=torch.rand(N, 1, generator=g), sc=torch.rand(N, 3, generator=g), rot=torch.rand(N, 4, generator=g))
    blk = merge.pack_block(parts["xyz"], parts["f_dc"], parts["f_rest"], parts["op"], parts["sc"], parts["rot"])
    ply_io.save_ply(str(tmp_path / "a.ply"), blk)
    back = torch.from_numpy(ply_io.load_ply(str(tmp_path / "a.ply")))
    merged = merge.merge_partitions(back, (0.2, None, None, 0.9))
    keep = (parts["xyz"][:, 0] >= 0.2) & (parts["xyz"][:, 1] <= 0.9)
    assert merged.shape[0] == int(keep.sum())
    # load_ply layout: f_rest columns are channel-major -> [N,3,15] -> transpose(1,2) gives back [N,15,3] (gaussian_model.py:256,271)
    fr = back[:, 6:51].reshape(N, 3, 15).transpose(1, 2)
    assert torch.equal(fr, parts["f_rest"])
