"""Timeline of the blend backward's log walk (a library built with -DGSR_LOG_PROFILE=3, selected through GSR_LIBRARY): start / end time stamp,
SIMD and list length of every wave of one launch -> how many waves are resident over the launch, when the SIMDs run dry, what the last
waves are.  usage: GSR_LIBRARY=tools/scratch/ab_<tag>_<n>/.../libgsrast_mi355x.so python tools/r5_wave_times.py [config] [camera]"""
import ctypes
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from based_on_point_cloud_partitions_amd import GaussianRasterizationSettings, GaussianRasterizer, _lib  # noqa: E402
from based_on_point_cloud_partitions_amd import synthetic as syn  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
cfg = syn.CONFIGS[name]
dev = torch.device("cuda", 0)
g = syn.make_gaussians(cfg["P"], seed=0).to(dev)
cams = syn.ring_cameras(cfg["cameras"], cfg["W"], cfg["H"], cfg["focal"])
dL = torch.randn(3, cfg["H"], cfg["W"], generator=torch.Generator().manual_seed(1)).to(dev)
bg = torch.zeros(3, device=dev)
params = [t.clone().requires_grad_(True) for t in (g.xyz, g.scales, g.rotations, g.opacities, g.shs)]
_lib.load()
raw = ctypes.CDLL(os.environ["GSR_LIBRARY"])
raw.gsr_debug_wave_times.argtypes = [ctypes.POINTER(ctypes.c_uint64), ctypes.c_int]
N = 4 * 32768
buf = (ctypes.c_uint64 * N)()


def step(c):
    r = GaussianRasterizer(GaussianRasterizationSettings(
        image_height=cfg["H"], image_width=cfg["W"], tanfovx=c.tanfovx, tanfovy=c.tanfovy, bg=bg, scale_modifier=1.0,
        viewmatrix=c.world_view_transform.to(dev), projmatrix=c.full_proj_transform.to(dev), sh_degree=3, campos=c.camera_center.to(dev),
        prefiltered=False, debug=False))
    img, _ = r(means3D=params[0], means2D=torch.zeros_like(params[0], requires_grad=True), opacities=params[3], shs=params[4],
               scales=params[1], rotations=params[2])
    img.backward(dL)


cam = cams[int(sys.argv[2]) if len(sys.argv) > 2 else 0]
for _ in range(6):
    step(cam)
torch.cuda.synchronize()
assert raw.gsr_debug_wave_times(buf, N) == 0
step(cam)
torch.cuda.synchronize()
assert raw.gsr_debug_wave_times(buf, N) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 4)
a = a[a[:, 1] != 0]
t0 = a[:, 0].astype(np.int64)
t1 = a[:, 1].astype(np.int64)
base = t0.min()
t0 = (t0 - base) / 100.0  # us (100 MHz)
t1 = (t1 - base) / 100.0
total = (a[:, 2] >> np.uint64(32)).astype(np.int64)
hi = (a[:, 3] >> np.uint64(32)).astype(np.int64)
hw = (a[:, 2] & np.uint64(0xFFFFFFFF)).astype(np.int64)
xcc = (a[:, 3] & np.uint64(0xF)).astype(np.int64)
simd = (hw >> 4) & 3
cu = (hw >> 8) & 15
se = (hw >> 13) & 7
sid = ((xcc * 8 + se) * 16 + cu) * 4 + simd
span = t1.max()
edges = np.linspace(0, span, 21)
resident = [int(((t0 < e) & (t1 > e)).sum()) for e in (edges[:-1] + edges[1:]) / 2]
# per SIMD: when its last wave ends, and the sum of its waves' resident time
ids = np.unique(sid)
last_end = np.array([t1[sid == i].max() for i in ids])
wave_time = np.array([(t1 - t0)[sid == i].sum() for i in ids])
order = np.argsort(t1)
out = {
    "config": name, "waves": int(len(a)), "simds_seen": int(len(ids)), "span_us": round(float(span), 1),
    "resident_waves_in_20_slices": resident,
    "wave_duration_us": {"mean": round(float((t1 - t0).mean()), 1), "p50": round(float(np.median(t1 - t0)), 1), "max": round(float((t1 - t0).max()), 1)},
    "simd_last_end_us": {"p05": round(float(np.percentile(last_end, 5)), 1), "p50": round(float(np.median(last_end)), 1),
                         "p95": round(float(np.percentile(last_end, 95)), 1), "max": round(float(last_end.max()), 1)},
    "simd_resident_wave_time_us": {"min": round(float(wave_time.min()), 1), "p50": round(float(np.median(wave_time)), 1), "max": round(float(wave_time.max()), 1)},
    "start_us_of_waves": {"p50": round(float(np.median(t0)), 1), "p90": round(float(np.percentile(t0, 90)), 1), "max": round(float(t0.max()), 1)},
    "last_10_waves": [{"end": round(float(t1[i]), 1), "start": round(float(t0[i]), 1), "list": int(total[i]), "walked": int(hi[i])} for i in order[-10:]],
    "longest_10_waves": [{"dur": round(float((t1 - t0)[i]), 1), "start": round(float(t0[i]), 1), "list": int(total[i]), "walked": int(hi[i])}
                         for i in np.argsort(t1 - t0)[-10:]],
    "corr_duration_walked": round(float(np.corrcoef(t1 - t0, hi)[0, 1]), 3),
}
print(json.dumps(out))
