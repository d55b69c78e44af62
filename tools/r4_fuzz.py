#!/usr/bin/env python3
"""Randomised parity run outside the test suite: many small scenes with cameras anywhere (inside the cloud, looking away from it,
splats crossing the near plane and the 1.3 tan(fov) clamp of forward.cu:82-87), odd image shapes, every SH degree, precomputed
colours / covariances, extreme opacities -- each one checked like a parity test (forward bit-exact, both tile-list modes; backward
through the two-stage check of tests/helpers.py).  Usage: python tools/r4_fuzz.py [n_scenes] [first_seed] -> gpurun_out/r4_fuzz.json"""
import json
import math
import os
import sys
import time
import traceback

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from helpers import assert_forward_bit_exact, check_backward, hip_forward_raw, oracle_forward  # noqa: E402

from based_on_point_cloud_partitions_amd import _lib  # noqa: E402
from based_on_point_cloud_partitions_amd import synthetic as syn  # noqa: E402


def one_scene(seed):
    rng = np.random.default_rng(77000 + seed)
    P = int(rng.choice([1, 5, 64, 300, 1000, 3000, 8000, 20000]))
    shape = rng.integers(0, 4)
    if shape == 0:
        W, H = int(rng.integers(1, 40)), int(rng.integers(200, 700))       # a sliver, tall
    elif shape == 1:
        W, H = int(rng.integers(200, 900)), int(rng.integers(1, 40))       # a sliver, wide
    else:
        W, H = int(rng.integers(16, 500)), int(rng.integers(16, 400))
    focal = float(rng.uniform(0.15, 1.5)) * max(W, H)
    deg = int(rng.integers(0, 4))
    lo = float(10 ** rng.uniform(-3.5, -1.0))
    hi = lo * float(10 ** rng.uniform(0.0, 2.0))
    g = syn.make_gaussians(P, seed=seed, scale_lo=lo, scale_hi=hi)
    where = int(rng.choice([0, 0, 0, 0, 1, 1, 1, 1, 2]))
    if where == 0:      # inside the cloud: Gaussians behind the camera, across the near plane, far outside the frustum
        pos = tuple(float(v) for v in rng.uniform((-3.5, -3.5, -0.9), (3.5, 3.5, 0.9)))
        target = tuple(float(v) for v in rng.uniform((-4, -4, -1), (4, 4, 1)))
    elif where == 1:    # outside, looking at a random point (part of the cloud out of view)
        th = rng.uniform(0, 2 * math.pi)
        r = rng.uniform(4.5, 12.0)
        pos = (float(r * math.cos(th)), float(r * math.sin(th)), float(rng.uniform(-3, 5)))
        target = tuple(float(v) for v in rng.uniform((-4, -4, -1), (4, 4, 1)))
    else:               # looking away from the cloud: (almost) everything culled
        pos = (float(rng.uniform(5, 8)), float(rng.uniform(5, 8)), 2.0)
        target = (float(pos[0] + rng.uniform(1, 3)), float(pos[1] + rng.uniform(1, 3)), float(rng.uniform(0, 4)))
    if abs(pos[0] - target[0]) + abs(pos[1] - target[1]) < 1e-3:
        target = (target[0] + 0.5, target[1], target[2])
    cam = syn.make_camera(W, H, focal, pos, target=target)
    op = g.opacities.clone()
    if rng.integers(0, 2):
        op[::5] = 1.0
        op[1::7] = 0.0
        op[2::11] = 1.0 / 255.0
    g = syn.SyntheticGaussians(g.xyz, g.scales, g.rotations, op, g.shs)
    bg = tuple(float(v) for v in rng.uniform(0, 1, 3))
    mod = float(rng.choice([1.0, 1.0, 0.3, 3.0]))
    cp = rng.uniform(0, 1, (P, 3)).astype(np.float32) if rng.integers(0, 4) == 0 else None
    cv = None
    if rng.integers(0, 4) == 0:
        A = rng.normal(0, hi, (P, 3, 3))
        S = A @ A.transpose(0, 2, 1) + 1e-6 * np.eye(3)
        cv = np.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], 1).astype(np.float32)
    desc = dict(seed=seed, P=P, W=W, H=H, focal=round(focal, 1), deg=deg, lo=lo, hi=hi, where=int(where), mod=mod,
                colors=cp is not None, cov=cv is not None)
    st = oracle_forward(g, cam, bg=bg, sh_degree=deg, scale_modifier=mod, colors_precomp=cp, cov3D_precomp=cv)
    desc["visible"] = int((st["radii"] > 0).sum())
    desc["R"] = int(st["num_rendered"]) if "num_rendered" in st else None
    dL = rng.standard_normal((3, H, W)).astype(np.float32)
    for cull in (1, 0):
        _lib.set_variant("CULL", cull)
        ctx = hip_forward_raw(g, cam, bg=bg, sh_degree=deg, scale_modifier=mod, colors_precomp=cp, cov3D_precomp=cv)
        assert_forward_bit_exact(ctx, st, colors_precomp=cp)
        check_backward(ctx, st, dL)
    _lib.set_variant("CULL", 1)
    return desc


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    first = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    out = {"scenes": 0, "failed": [], "visible_total": 0, "instances_total": 0, "by_where": [0, 0, 0]}
    t0 = time.time()
    for seed in range(first, first + n):
        try:
            d = one_scene(seed)
            out["scenes"] += 1
            out["visible_total"] += d["visible"]
            out["instances_total"] += d["R"] or 0
            out["by_where"][d["where"]] += 1
        except Exception:  # noqa: BLE001
            _lib.set_variant("CULL", 1)
            out["failed"].append({"seed": seed, "error": traceback.format_exc()[-1500:]})
            print("FAILED seed", seed, flush=True)
        if (seed - first) % 20 == 19:
            print(f"{seed - first + 1} scenes, {len(out['failed'])} failed, {time.time() - t0:.0f} s", flush=True)
    out["seconds"] = round(time.time() - t0, 1)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", os.environ.get("GSR_FUZZ_OUT", "r4_fuzz.json")), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "failed"}), "failures:", len(out["failed"]))
    for fl in out["failed"][:5]:
        print(fl["seed"], fl["error"][-600:])
    torch.cuda.synchronize()


if __name__ == "__main__":
    main()
