#!/usr/bin/env python3
"""CPU estimate (oracle data, cfg2 camera 0): how many (entry, 8x8 block) visits of the blend kernels a liveness test per 4x4 sub-block
would save.  Builds tools/r4_subblock_stat.c into tools/scratch/ and prints the counts.  usage: python tools/r4_subblock_stat.py"""
import ctypes
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import oracle_forward  # noqa: E402

from based_on_point_cloud_partitions_amd import synthetic as syn  # noqa: E402

so = os.path.join(ROOT, "tools", "scratch", "libsubstat.so")
os.makedirs(os.path.dirname(so), exist_ok=True)
subprocess.check_call(["gcc", "-O3", "-march=native", "-fopenmp", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tools", "r4_subblock_stat.c"), "-lm"])
cfg = syn.CONFIGS["cfg2"]
g = syn.make_gaussians(cfg["P"], seed=0)
cam = syn.ring_cameras(16, cfg["W"], cfg["H"], cfg["focal"])[0]
st = oracle_forward(g, cam)
L = ctypes.CDLL(so)
out = np.zeros(8, np.int64)
p = lambda a, t: a.ctypes.data_as(ctypes.POINTER(t))
L.subblock_stats(ctypes.c_int(cfg["W"]), ctypes.c_int(cfg["H"]), p(st["ranges"], ctypes.c_uint32), p(st["point_list"], ctypes.c_uint32),
                 p(st["means2D"], ctypes.c_float), p(st["conic_opacity"], ctypes.c_float), p(out, ctypes.c_longlong))
print(dict(fwd_visits=int(out[0]), fwd_visits_4x4_liveness=int(out[1]), fwd_blending_visits=int(out[2]), bwd_visits=int(out[3]),
           bwd_visits_4x4_liveness=int(out[4]), fwd_visits_8x4=int(out[6]), fwd_visits_4x8=int(out[7])))
print(f"forward: {100 * (1 - out[1] / out[0]):.1f} % of the visits saved (8x4: {100 * (1 - out[6] / out[0]):.1f} %, 4x8: {100 * (1 - out[7] / out[0]):.1f} %); "
      f"backward: {100 * (1 - out[4] / out[3]):.1f} %; {100 * out[2] / out[0]:.1f} % of the forward's visits blend a pixel")
