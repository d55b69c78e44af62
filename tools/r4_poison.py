#!/usr/bin/env python3
"""Inputs a diverging optimisation hands the rasterizer: NaN, +-Inf, 1e30, denormals, negative scales, zero quaternions, opacities
outside [0, 1] in ~2 % of the entries of every input tensor.  The reference's behaviour on those Gaussians is undefined ((int)NaN in
getRect, auxiliary.h:46-58); what must hold here is memory safety and termination: every index the kernels derive stays inside the
caller's buffers (rectangles are clamped to the tile grid before any use), forward + backward + kNN return, the binning is
self-consistent (offsets end at num_rendered, ranges and ids in bounds), and the per-Gaussian outputs of the UNTOUCHED Gaussians are
still bit-exact against the oracle.  python tools/r4_poison.py [n_scenes]"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from helpers import hip_backward_raw, hip_forward_raw, oracle_forward  # noqa: E402

from based_on_point_cloud_partitions_amd import _lib, distCUDA2  # noqa: E402
from based_on_point_cloud_partitions_amd import synthetic as syn  # noqa: E402

BAD = np.array([np.nan, np.inf, -np.inf, 1e30, -1e30, 1e-42, 0.0, -1.0, 3e38], dtype=np.float32)


def poison(t, rng, frac=0.02):
    a = t.numpy().copy()
    flat = a.reshape(-1)
    k = max(1, int(frac * flat.size))
    where = rng.choice(flat.size, size=k, replace=False)
    flat[where] = rng.choice(BAD, size=k)
    rows = np.unique(where // int(np.prod(a.shape[1:])))
    return torch.from_numpy(a), rows


def run(n, quiet=False):
    done = 0
    for seed in range(n):
        rng = np.random.default_rng(9100 + seed)
        P = int(rng.choice([64, 1000, 5000, 30000]))
        W, H = int(rng.integers(16, 400)), int(rng.integers(16, 300))
        g = syn.make_gaussians(P, seed=seed, scale_lo=0.004, scale_hi=float(rng.choice([0.04, 0.4])))
        cam = syn.ring_cameras(4, W, H, float(rng.uniform(0.3, 1.2)) * max(W, H))[seed % 4]
        fields, touched = [], set()
        for t in (g.xyz, g.scales, g.rotations, g.opacities, g.shs):
            pt, rows = poison(t, rng)
            fields.append(pt)
            touched.update(rows.tolist())
        gp = syn.SyntheticGaussians(*fields)
        clean = np.array(sorted(set(range(P)) - touched), dtype=np.int64)
        st = oracle_forward(gp, cam)
        for cull in (1, 0):
            _lib.set_variant("CULL", cull)
            ctx = hip_forward_raw(gp, cam)
            ex, R = {k: v.cpu().numpy() for k, v in ctx["ex"].items()}, int(ctx["R"])
            radii = ctx["radii"].cpu().numpy()
            assert np.array_equal(radii[clean], st["radii"][clean]), "radii of untouched Gaussians"
            tt = ex["tiles_touched"].view(np.uint32)
            assert np.array_equal(tt[clean], st["tiles_touched"][clean]), "tiles_touched of untouched Gaussians"
            assert int(ex["point_offsets"].view(np.uint32)[-1]) == R == int(tt.astype(np.int64).sum()), "offsets end at num_rendered"
            rg = ex["ranges"].view(np.uint32).reshape(-1, 2).astype(np.int64)
            assert (rg[:, 0] <= rg[:, 1]).all() and rg.max(initial=0) <= R, "ranges in bounds"
            listed = int((rg[:, 1] - rg[:, 0]).sum())
            assert (ex["point_list"].view(np.uint32)[:listed].astype(np.int64) < P).all(), "ids in bounds"
            assert cull == 1 or listed == R
            dL = rng.standard_normal((3, H, W)).astype(np.float32)
            dL[:, ::7, ::5] = np.nan  # and a loss that went NaN in places
            got = hip_backward_raw(ctx, dL)
            assert all(v.shape[0] == P for v in got.values())
        _lib.set_variant("CULL", 1)
        d = distCUDA2(gp.xyz.to("cuda"))
        torch.cuda.synchronize()
        assert d.shape == (P,)
        done += 1
        if seed % 10 == 9 and not quiet:
            print(done, "scenes survived", flush=True)
    return done


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    done = run(n)
    print("poisoned scenes:", done, "of", n, "-- forward, backward and kNN returned, binning self-consistent, untouched Gaussians bit-exact")


if __name__ == "__main__":
    main()
