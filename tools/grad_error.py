#!/usr/bin/env python3
"""Measures the gradient error of the HIP backward on the benchmark scene (GPU box): tools/grad_error.py [out.json] [config]
cfg2, camera 0, the bench's N(0,1) dL/dcolor.  Per tensor: max |err|, the share of elements inside a plain 1e-4 absolute bar, and the
worst |err| / tolerance under the two-stage bar of tests/helpers.py (stage 1: blend sums vs the oracle's double sums; stage 2: chain
outputs vs the oracle's chain on the kernel's own sums).  `one_stage:*` = the chain outputs against the oracle's full backward (its
own sums), i.e. what north_star's "1e-4 abs" reads on without any conditioning -- reported, not asserted."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import CHAIN_KEYS, assert_forward_bit_exact, hip_backward_raw, hip_forward_raw, oracle_forward, two_stage_check  # noqa: E402
from oracle import oracle  # noqa: E402

from based_on_point_cloud_partitions_amd import synthetic as syn  # noqa: E402


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "grad_error.json")
    name = sys.argv[2] if len(sys.argv) > 2 else "cfg2"
    cfg = syn.CONFIGS[name]
    g = syn.make_gaussians(cfg["P"], seed=0)
    cam = syn.ring_cameras(cfg["cameras"], cfg["W"], cfg["H"], cfg["focal"])[0]
    dL = torch.randn(3, cfg["H"], cfg["W"], generator=torch.Generator().manual_seed(1)).numpy()
    st = oracle_forward(g, cam)
    ctx = hip_forward_raw(g, cam)
    assert_forward_bit_exact(ctx, st)
    got = hip_backward_raw(ctx, dL, capture=True)
    stats = two_stage_check(st, got, dL)
    want = oracle.backward(st, dL)
    for k in CHAIN_KEYS:
        e = np.abs(got[k].astype(np.float64) - want[k])
        stats["one_stage:" + k] = dict(max_abs_err=float(e.max()), frac_within_1e_4_abs=float(np.mean(e <= 1e-4)),
                                       frac_within_1e_4_abs_plus_1e_4_rel=float(np.mean(e <= 1e-4 + 1e-4 * np.abs(want[k]))),
                                       max_abs_value=float(np.abs(want[k]).max()), n=int(e.size))
    for v in stats.values():
        v.pop("worst", None)
    res = dict(workload=f"{name}: {cfg['P']} Gaussians, {cfg['W']}x{cfg['H']}, camera 0, N(0,1) dL/dcolor (seed 1)", R=int(ctx["R"]),
               bar="stage 1: 1e-4 + 1e-4|want| + 1e-6 cond; stage 2: bit-exact", tensors=stats)
    json.dump(res, open(out, "w"), indent=1, sort_keys=True)
    for k, v in stats.items():
        print(k, {a: (round(b, 8) if isinstance(b, float) else b) for a, b in v.items()})


if __name__ == "__main__":
    main()
