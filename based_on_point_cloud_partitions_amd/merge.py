"""Seamless merge of independently trained partitions: the one exchange step of the path.

Semantics follow scene/ptgs/merge_partitions.py:95-170 (and scene/sever_code/merge_partitions.py:54-106):
  1. every partition keeps the Gaussians inside its `origin_box` -- closed intervals, a side without a neighbouring
     partition is left open (`dynamic_bounding_box`, :10-57; `extract_point_cloud_mask`, :95-109);
  2. the survivors of all partitions are concatenated in partition order;
  3. `np.unique(points, axis=0, return_index=True)` (:152-158): exact-xyz duplicates keep their FIRST occurrence and the
     result is emitted in lexicographic (x, y, z) order.
The reference does this through PLY files on disk; here every rank holds its partition's `[N_i, 59]` block of raw
(pre-activation) parameters -- xyz 3, f_dc 3, f_rest 45, opacity 1, scale 3, rot 4, the PLY column order of
scene/gaussian_model.py:189-220 without the normals -- and the exchange is two all-gathers (counts, then padded blocks)
over torch.distributed: RCCL over xGMI on the 8-GPU node (backend "nccl"), gloo in the CPU tests.  ~118 MB per rank
at 500k Gaussians, so a single un-bucketed all-gather is the right size for the point-to-point xGMI links.
"""
import torch

ROW = 59  # floats per Gaussian in the exchange block


def pack_block(xyz, f_dc, f_rest, opacity, scaling, rotation):
    """[N,3],[N,1,3] or [N,3],[N,15,3] or [N,45],[N,1],[N,3],[N,4] -> [N,59] in PLY column order
    (save_ply flattens features as transpose(1,2).flatten(1): channel-major)."""
    N = xyz.shape[0]
    if f_dc.dim() == 3:
        f_dc = f_dc.transpose(1, 2).reshape(N, -1)
    if f_rest.dim() == 3:
        f_rest = f_rest.transpose(1, 2).reshape(N, -1)
    return torch.cat((xyz, f_dc, f_rest, opacity.reshape(N, 1), scaling, rotation), dim=1).contiguous()


def crop_mask(xyz, box):
    """box = (minx, miny, maxx, maxy); None = open side (no neighbouring partition on that side)."""
    minx, miny, maxx, maxy = box
    mask = torch.ones(xyz.shape[0], dtype=torch.bool, device=xyz.device)
    if minx is not None:
        mask &= xyz[:, 0] >= minx
    if maxx is not None:
        mask &= xyz[:, 0] <= maxx
    if miny is not None:
        mask &= xyz[:, 1] >= miny
    if maxy is not None:
        mask &= xyz[:, 1] <= maxy
    return mask


def _ordered_int(col):
    """float32 column -> int32 with the same order under signed comparison (-0.0 counted as +0.0, as float comparison does)."""
    i = (col + 0.0).contiguous().view(torch.int32)  # x + 0.0 turns -0.0 into +0.0
    return i ^ ((i >> 31) & 0x7FFFFFFF)


def dedup_first_lexicographic(rows):
    """np.unique(rows[:, :3], axis=0, return_index=True) semantics applied to whole rows: output sorted by (x,y,z),
    for equal xyz the row with the smallest original index survives.
    Two stable sorts of integer keys -- z, then (x, y) packed into one int64 -- over the three contiguous coordinate columns, and ONE
    gather of the surviving 236-byte rows (round 3: three stable float sorts, each behind a strided gather of a column of the
    [N, 59] block: 206 ms for 500k rows on the MI355X)."""
    n = rows.shape[0]
    if n == 0:
        return rows
    xyz = rows[:, :3].t().contiguous()  # [3, n]: coalesced columns
    ix, iy, iz = _ordered_int(xyz[0]), _ordered_int(xyz[1]), _ordered_int(xyz[2])
    _, order = torch.sort(iz, stable=True)
    # (x, y) as one signed 64-bit key: x in the high half as it is, y in the low half shifted into 0 .. 2^32 - 1
    kxy = (ix.to(torch.int64) << 32) | (iy.to(torch.int64) + 2147483648)
    _, idx = torch.sort(kxy[order], stable=True)
    order = order[idx]
    sx, sy, sz = ix[order], iy[order], iz[order]
    first = torch.ones(n, dtype=torch.bool, device=rows.device)
    first[1:] = (sx[1:] != sx[:-1]) | (sy[1:] != sy[:-1]) | (sz[1:] != sz[:-1])
    # a row with a NaN coordinate equals no other row under the float comparison np.unique's field-wise compare makes: never dropped
    # (the integer keys would merge NaNs of identical bits)
    first |= torch.isnan(xyz).any(dim=0)[order]
    return rows[order[first]]


def unpack_block(rows):
    """[N,59] -> dict of GaussianModel's raw parameter tensors (`_xyz`, `_features_dc` [N,1,3], `_features_rest` [N,15,3], ...):
    the inverse of pack_block, i.e. what load_ply builds (scene/gaussian_model.py:236-278)."""
    N = rows.shape[0]
    return dict(xyz=rows[:, 0:3].contiguous(), features_dc=rows[:, 3:6].reshape(N, 3, 1).transpose(1, 2).contiguous(),
                features_rest=rows[:, 6:51].reshape(N, 3, 15).transpose(1, 2).contiguous(), opacity=rows[:, 51:52].contiguous(),
                scaling=rows[:, 52:55].contiguous(), rotation=rows[:, 55:59].contiguous())


def dynamic_bounding_box(index, origin_boxes, eps=1e-9):
    """Crop box of partition `index` among axis-aligned `origin_boxes` [(minx, miny, maxx, maxy), ...]: a side stays closed only
    where a neighbour -- a box sharing a boundary segment of non-zero length (find_adjacent_partitions,
    scene/ptgs/shen_partition.py:165-186) -- has its opposite side on it; a side without neighbour is left open (None)
    (dynamic_bounding_box, scene/ptgs/merge_partitions.py:10-57)."""
    minx, miny, maxx, maxy = origin_boxes[index]
    left = right = down = up = False
    for j, (nx0, ny0, nx1, ny1) in enumerate(origin_boxes):
        if j == index or (nx0, ny0, nx1, ny1) == (minx, miny, maxx, maxy):
            continue
        # intersection of the two rectangles must be a segment of non-zero length (not a corner, not an area)
        ix0, ix1, iy0, iy1 = max(minx, nx0), min(maxx, nx1), max(miny, ny0), min(maxy, ny1)
        if ix1 < ix0 or iy1 < iy0:
            continue
        w, h = ix1 - ix0, iy1 - iy0
        if not ((w == 0) != (h == 0)):
            continue
        left |= abs(nx1 - minx) < eps
        right |= abs(nx0 - maxx) < eps
        down |= abs(ny1 - miny) < eps
        up |= abs(ny0 - maxy) < eps
    return (minx if left else None, miny if down else None, maxx if right else None, maxy if up else None)


def all_gather_blocks(kept, group=None):
    """The exchange step: every rank contributes its cropped [n_i, 59] block, every rank receives the concatenation in rank
    (= partition) order.  Two collectives -- the counts, then ONE all-gather of the blocks padded to the largest count
    (`all_gather_into_tensor`: a single RCCL call over xGMI, ~118 MB per rank at 500k Gaussians) -- and one host sync for the
    counts.  Returns (rows [sum n_i, 59], counts list)."""
    import torch.distributed as dist

    world = dist.get_world_size(group)
    if kept.is_cuda and dist.get_backend(group) == "gloo":  # single-GPU rehearsals of the N > 1 path: gloo gathers host tensors
        rows, counts = all_gather_blocks(kept.cpu(), group)
        return rows.to(kept.device), counts
    count = torch.tensor([kept.shape[0]], dtype=torch.int64, device=kept.device)
    counts_t = torch.empty(world, dtype=torch.int64, device=kept.device)
    dist.all_gather_into_tensor(counts_t, count, group=group)
    counts = counts_t.tolist()
    maxn = max(max(counts), 1)
    padded = torch.zeros((maxn, ROW), dtype=kept.dtype, device=kept.device)
    padded[:kept.shape[0]] = kept
    gathered = torch.empty((world * maxn, ROW), dtype=kept.dtype, device=kept.device)
    dist.all_gather_into_tensor(gathered, padded, group=group)
    gathered = gathered.view(world, maxn, ROW)
    return torch.cat([gathered[r, :c] for r, c in enumerate(counts)], dim=0), counts


def merge_partitions(block, box, group=None):
    """Collective: every rank passes its own [N_i, 59] block and crop box and receives the merged [N, 59] scene."""
    import torch.distributed as dist

    kept = block[crop_mask(block[:, :3], box)].contiguous()
    if not (dist.is_available() and dist.is_initialized()):
        return dedup_first_lexicographic(kept)
    allrows, _ = all_gather_blocks(kept, group)  # partition (= rank) order
    return dedup_first_lexicographic(allrows)
