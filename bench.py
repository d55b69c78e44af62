#!/usr/bin/env python3
"""Benchmark of the hot path: raster steps (GaussianRasterizer forward + backward) per second.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...)

One "step" = one pass of the hot path over one camera of the partition: frustum cull + covariance projection +
SH colour, tile binning (prefix sum, key emission, 64-bit radix sort, tile ranges), alpha compositing, and the
full backward for a fixed synthetic dL/dcolor -- what train.py:107 + :118 spend in the rasterizer.
Workload at N = 1: BASELINE.json configs[1] (one partition of 500,000 random-init Gaussians, 1920x1080, SH degree 3,
16 cameras on a ring); at N > 1 every rank owns its own 500k-Gaussian partition (weak scaling, no data-path
collective: the reference trains partitions independently, auto_train.py:22-51).  A rank's partition fills the same volume as the
N = 1 scene -- its origin box is 8/1.2 wide, so the box expanded by the reference's 10 % per side is 8 x 8 x 2 around its own camera
ring -- with its own seed: the per-GPU work is the same distribution at every N (R = 3.70 M instances for camera 0 at N = 1, 3.70 +-
0.01 M on the ranks of N = 8; with 8-wide origin boxes, as until round 4, a rank had 15 % fewer).
Inputs are resident in HBM before the timed region.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
PARTITION_CELL = (8.0 / 1.2, 8.0 / 1.2)  # origin box of a rank's partition at N > 1; expanded by 10 % per side it is the N = 1 scene's 8 x 8


def _requested_gpus(argv):
    for i, a in enumerate(argv):
        if a == "--gpus" and i + 1 < len(argv):
            return int(argv[i + 1])
        if a.startswith("--gpus="):
            return int(a.split("=", 1)[1])
    return 1


def _free_port():
    import socket

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks_if_needed(argv):
    """`python bench.py --gpus N` from a plain shell (the per-partition launcher of the reference, auto_train.py:22-51, is a loop of
    `os.system("python train.py ...")`): when N > 1 and no launcher has set RANK, start the N ranks with torch.distributed.run as a
    FRESH CHILD PROCESS -- before torch is imported here, so this parent never touches a GPU -- hand the child our stdout (rank 0's
    JSON line arrives unchanged) and exit with its return code."""
    n = _requested_gpus(argv)
    if n <= 1 or "RANK" in os.environ:
        return
    import subprocess

    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + list(argv)
    sys.exit(subprocess.run(cmd, env=env).returncode)


if __name__ == "__main__":
    launch_ranks_if_needed(sys.argv[1:])

import numpy as np  # noqa: E402
import torch  # noqa: E402

from based_on_point_cloud_partitions_amd import GaussianRasterizationSettings, GaussianRasterizer, _lib  # noqa: E402
from based_on_point_cloud_partitions_amd import synthetic as syn  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
VALU_PLAIN_CYCLES = 2.4  # SIMD cycles per plain wave64 fp32 add / mul / fma with 4-8 waves per SIMD, at a nominal 2.4 GHz (profiles/r01_valu_cost.txt)
VALU_GUIDE_CYCLES = 2.0  # MI355X_MICROARCH.md: wave64 v_fma_f32 = 2 cycles -> 1024 SIMDs x 2.4 GHz / 2 = 1,229 G wave-instructions/s
# Round 5 (tools/salu_probe.hip, profiles/r05_salu_probe.txt), ns per wave-instruction per SIMD with 8 waves resident: fp32 add / mul / fma 1.06;
# EVERY other VALU class measured (int32, compares, selects, SGPR operands, readlane / writelane, DPP) 1.75-1.9; every SALU instruction 1.75
# (not-taken branch ~0.4); v_exp / v_rcp / permlane swaps ~3.6 (r01).  SALU and VALU issue side by side (32 v_fma + 32 s_and per wave take
# the SALU time + 10 %), so a kernel has TWO issue budgets: its VALU time and its SALU time.
NS_SALU = 1.75
VALU_BOUND = ("render_backward", "render_forward")  # the VALU-issue-bound kernels


def issue_shares():
    """Static instruction shares of the blend kernels (profiles/blend_kernels_issue_shares.json, written by tools/r5_opcode_table.py): which
    part of a kernel's fp32 fma instructions reads three registers and which part of its fp32 adds carries a DPP modifier -- the class
    counters cannot tell, the issue costs differ by 1.7x."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "blend_kernels_issue_shares.json")))
    except Exception:
        return {}


def price_list_cycles(v, shares=None):
    """Mean SIMD cycles per VALU instruction of a kernel from its PMC instruction-class counts (profiles/pmc_traffic.json) and the
    measured price list of gfx950 (tools/salu_probe.hip; ns per wave-instruction per SIMD x 2.4 GHz): fp32 add / mul and fma with two
    register sources 2.4 cycles (0.95-1.0 ns); **fma reading three registers 4.0 (1.7 ns) and DPP adds 4.2 (1.76 ns)** -- the class counters
    count them as plain fma / add, so their shares come from the static listing of the template instance that ran (`kernel_name` in the
    PMC file, `shares`; without them everything counted as add / mul / fma is priced plain, as rounds 1-4 did); transcendentals (v_exp,
    v_rcp) 8.5; everything the class counters do not name -- compares, selects, lane reads, int32, converts, moves from SGPRs -- 4.3 (some
    of those are plain moves: an estimate of the kernel's issue time, not a bound).  None when the class counters are not on file."""
    need = ("sq_insts_valu", "sq_insts_valu_add_f32", "sq_insts_valu_mul_f32", "sq_insts_valu_fma_f32", "sq_insts_valu_int32",
            "sq_insts_valu_trans_f32")
    if not all(k in v for k in need):
        return None
    name = v.get("kernel_name")
    f3 = (shares or {}).get("three_register_share_of_fp32_fma", {}).get(name, 0.0)
    dpp = (shares or {}).get("dpp_share_of_fp32_add", {}).get(name, 0.0)
    add, mul, fma = v["sq_insts_valu_add_f32"], v["sq_insts_valu_mul_f32"], v["sq_insts_valu_fma_f32"]
    trans = v["sq_insts_valu_trans_f32"]
    other = max(v["sq_insts_valu"] - add - mul - fma - trans, 0.0)
    cycles = 2.4 * (add * (1.0 - dpp) + mul + fma * (1.0 - f3)) + 4.2 * add * dpp + 4.0 * fma * f3 + 8.5 * trans + 4.3 * other
    return cycles / v["sq_insts_valu"]
FP32_VALU_PEAK_TFLOPS = 157.3


def stage_bytes(P, V, R, N, key_bits, K=16):
    """ALGORITHMIC bytes per launch of each stage (SURVEY.md §8d / DESIGN.md): defined by the reference's data flow."""
    passes = (key_bits + 7) // 8
    return {
        "preprocess": 20 * P + (32 + 12 * K + 67) * V,
        "scan": 8 * P,
        "emit_keys": 8 * P + 12 * V + 12 * R,
        "radix_sort": 8 * R + 24 * R * passes,
        "tile_ranges": 8 * R,
        "render_forward": 40 * R + 20 * N,
        "render_backward": 76 * R + 20 * N,
        "gaussian_backward": 308 * P + (88 + 143 + 24 * K) * V,
    }


def per_rank_report(rows, steps, dominant):
    """rows[r] = the figures rank r measured itself (see `mine` in main) -> the `per_rank` / `slowest_rank` / `skew_ms` fields of the
    N > 1 line: which rank set the MAX-over-ranks time, how far apart the ranks left the opening barrier, and whether the slow rank had
    more work (V, R, listed instances) or just ran slower (the dominant stage's own average)."""
    keys = ("elapsed_s", "opening_barrier_wait_ms", "t0_wall", "V_mean", "R_mean", "instances_listed_mean", "tile_list_mean",
            "tile_list_max", "dominant_stage_avg_ms", "render_backward_avg_ms", "ms_per_step_cold", "closing_barrier_wait_ms",
            "spinup_last_batch_ms_per_step", "device_index", "dominant_stage_index")
    t_first = min(r[2] for r in rows)
    per = []
    for i, r in enumerate(rows):
        d = dict(zip(keys, r))
        per.append({"rank": i, "device_index": int(d["device_index"]), "ms_per_step": round(1e3 * d["elapsed_s"] / steps, 4),
                    "it_per_s": round(steps / d["elapsed_s"], 2), "opening_barrier_wait_ms": round(d["opening_barrier_wait_ms"], 3),
                    "start_offset_ms": round(1e3 * (d["t0_wall"] - t_first), 3), "closing_barrier_wait_ms": round(d["closing_barrier_wait_ms"], 3),
                    "V_mean": round(d["V_mean"], 1), "R_mean": round(d["R_mean"], 1), "instances_listed_mean": round(d["instances_listed_mean"], 1),
                    "tile_list_mean": round(d["tile_list_mean"], 1), "tile_list_max": int(d["tile_list_max"]),
                    # (every rank picks its own dominant stage from its own warm-up: the one it carried events around in the timed region)
                    "dominant_stage": (_lib.STAGES[int(d["dominant_stage_index"])] if "dominant_stage_index" in d else dominant),
                    "dominant_stage_avg_ms": round(d["dominant_stage_avg_ms"], 4),
                    "render_backward_avg_ms": round(d["render_backward_avg_ms"], 4), "ms_per_step_cold": round(d["ms_per_step_cold"], 4),
                    "spinup_last_batch_ms_per_step": round(d["spinup_last_batch_ms_per_step"], 4)})
    ms = [p_["ms_per_step"] for p_ in per]
    slow, fast = int(np.argmax(ms)), int(np.argmin(ms))
    return {"per_rank": per, "slowest_rank": slow, "fastest_rank": fast,
            "skew_ms": round(max(p_["start_offset_ms"] for p_ in per), 3),  # spread of the ranks' start stamps (one node, one clock)
            "ms_per_step_spread": round(ms[slow] - ms[fast], 4), "ms_per_step_mean_over_ranks": round(float(np.mean(ms)), 4),
            "sum_of_rank_rates_it_per_s": round(sum(p_["it_per_s"] for p_ in per), 2),
            "slowest_vs_fastest": {"ms_per_step": round(ms[slow] / ms[fast], 4),
                                   "instances_listed": round(per[slow]["instances_listed_mean"] / max(per[fast]["instances_listed_mean"], 1.0), 4),
                                   "render_backward_avg_ms": round(per[slow]["render_backward_avg_ms"] / max(per[fast]["render_backward_avg_ms"], 1e-9), 4)}}


def higher_msb(n):
    msb, step = 16, 16
    while step > 1:
        step //= 2
        msb = msb + step if (n >> msb) else msb - step
    if n >> msb:
        msb += 1
    return msb


def _oracle_step(g, cam, dL):
    from oracle import oracle

    t0 = time.perf_counter()
    st = oracle.forward(g.xyz.numpy(), g.opacities.numpy(), shs=g.shs.numpy(), scales=g.scales.numpy(),
                        rotations=g.rotations.numpy(), viewmatrix=cam.world_view_transform.numpy(),
                        projmatrix=cam.full_proj_transform.numpy(), campos=cam.camera_center.numpy(),
                        bg=np.zeros(3, np.float32), W=cam.image_width, H=cam.image_height, tanfovx=cam.tanfovx,
                        tanfovy=cam.tanfovy, sh_degree=3)
    t1 = time.perf_counter()
    oracle.backward(st, dL)
    return t1 - t0, time.perf_counter() - t1, st


def cpu_baseline(cfg, g, cams, dL, n_steps=16):
    """The oracle (a CPU port of the reference algorithm; the reference has no CPU path) on a bounded sample of the same workload:
    `n_steps` raster steps (cameras 0..n_steps-1) with every host thread, plus -- SURVEY.md §8(d) -- one step of the small
    reference-runnable case (cfg1: 10k Gaussians, 256x256) with one thread and with all of them, so the OpenMP scaling is visible."""
    from oracle import oracle
    from based_on_point_cloud_partitions_amd import synthetic as syn

    avail = os.cpu_count() or 1
    os.environ.setdefault("OMP_NUM_THREADS", str(avail))
    # the box may grant fewer CPUs than it shows (a container's CPU share): take the thread count that is fastest on one step
    tried = {}
    for threads in sorted({avail, min(avail, 64), min(avail, 32), min(avail, 16)}, reverse=True):
        oracle.set_threads(threads)
        f, b, _ = _oracle_step(g, cams[0], dL)
        tried[threads] = round(f + b, 3)
    cores = min(tried, key=tried.get)
    oracle.set_threads(cores)
    fwd = bwd = 0.0
    for i in range(n_steps):
        f, b, st = _oracle_step(g, cams[i % len(cams)], dL)
        fwd, bwd = fwd + f, bwd + b
        if i == 0:
            stats = {"pair_visits": int(st["stats"][0]), "blended_pairs": int(st["stats"][1])}  # camera 0, reference algorithm
    small_g = syn.make_gaussians(10_000, seed=0)
    small_cam = syn.ring_cameras(4, 256, 256, 160.0)[0]
    small_dL = np.random.default_rng(1).standard_normal((3, 256, 256)).astype(np.float32)
    small = {}
    for label, threads in (("1_thread", 1), (f"{cores}_threads", cores)):
        oracle.set_threads(threads)
        _oracle_step(small_g, small_cam, small_dL)  # warm
        f, b, _ = _oracle_step(small_g, small_cam, small_dL)
        small[label + "_it_per_s"] = round(1.0 / (f + b), 2)
    oracle.set_threads(cores)
    return {"value": n_steps / (fwd + bwd), "unit": "it/s", "cores": cores, "kind": "port",
            "build": "gcc -O3 -march=x86-64-v3 -fopenmp, OpenMP over Gaussians / tiles, parallel stable LSD radix sort (oracle/Makefile)",
            "sample": f"{n_steps} raster steps (cameras 0..{n_steps - 1} of the {cfg['P']}-Gaussian {cfg['W']}x{cfg['H']} workload): "
                      f"forward {fwd:.2f} s + backward {bwd:.2f} s in all, OpenMP over Gaussians/tiles; "
                      f"thread count chosen from one step each: {tried} s",
            "cfg1_10k_gaussians_256x256": small}, stats


def torch_l1_ssim(img, gt, lam, l1_image=None):
    """Stock-PyTorch evaluation of utils/loss_utils.py + train.py:116-117 (what the reference runs on the GPU), for timing.
    `l1_image`: the image the L1 term is taken on (train.py:112,116: the appearance-decoupled one); default img."""
    import torch.nn.functional as F
    from math import exp

    g1 = torch.tensor([exp(-(x - 5) ** 2 / float(2 * 1.5 ** 2)) for x in range(11)], device=img.device)
    g1 = (g1 / g1.sum()).unsqueeze(1)
    C = img.size(-3)
    window = g1.mm(g1.t()).float().unsqueeze(0).unsqueeze(0).expand(C, 1, 11, 11).contiguous()
    mu1 = F.conv2d(img, window, padding=5, groups=C)
    mu2 = F.conv2d(gt, window, padding=5, groups=C)
    mu1_sq, mu2_sq, mu1_mu2 = mu1.pow(2), mu2.pow(2), mu1 * mu2
    sigma1_sq = F.conv2d(img * img, window, padding=5, groups=C) - mu1_sq
    sigma2_sq = F.conv2d(gt * gt, window, padding=5, groups=C) - mu2_sq
    sigma12 = F.conv2d(img * gt, window, padding=5, groups=C) - mu1_mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    ssim_map = ((2 * mu1_mu2 + C1) * (2 * sigma12 + C2)) / ((mu1_sq + mu2_sq + C1) * (sigma1_sq + sigma2_sq + C2))
    return (1.0 - lam) * torch.abs((img if l1_image is None else l1_image) - gt).mean() + lam * (1.0 - ssim_map.mean())


def full_step_timing(args, rasterizers, g, dL, H, W, dev):
    """Context figures (not the headline): the training step from GaussianModel's RAW parameters, i.e. including the
    activations of scene/gaussian_model.py:102-122 and the loss of train.py:116-117, in the reference's data flow (torch
    activations + rasterizer [+ torch loss]) and with the fused raw-parameter operator [+ fused HIP loss]."""
    from based_on_point_cloud_partitions_amd import loss as hip_loss
    from based_on_point_cloud_partitions_amd.renderer import rasterize_gaussian_params

    leaf = lambda t: t.to(dev).contiguous().requires_grad_(True)
    xyz, f_dc, f_rest = leaf(g.xyz), leaf(g.shs[:, :1]), leaf(g.shs[:, 1:])
    op_raw, sc_raw = leaf(torch.logit(g.opacities.clamp(1e-6, 1 - 1e-6))), leaf(torch.log(g.scales))
    rot_raw = leaf(g.rotations * 1.5)
    params = [xyz, f_dc, f_rest, op_raw, sc_raw, rot_raw]
    gt = torch.rand(3, H, W, generator=torch.Generator().manual_seed(2)).to(dev)

    def render_reference(r):  # gaussian_renderer/__init__.py:52-93 with the getters of gaussian_model.py:102-122
        screenspace = torch.zeros_like(xyz, requires_grad=True)
        image, radii = r(means3D=xyz, means2D=screenspace, opacities=torch.sigmoid(op_raw), shs=torch.cat((f_dc, f_rest), dim=1),
                         scales=torch.exp(sc_raw), rotations=torch.nn.functional.normalize(rot_raw))
        return image, screenspace, radii

    def render_fused(r):
        screenspace = torch.zeros_like(xyz, requires_grad=True)
        image, radii = rasterize_gaussian_params(xyz, screenspace, f_dc, f_rest, op_raw, sc_raw, rot_raw, r.raster_settings)
        return image, screenspace, radii

    from based_on_point_cloud_partitions_amd.optim import Adam as HipAdam
    from based_on_point_cloud_partitions_amd.densify import add_densification_stats

    class Stats:  # the three statistics tensors of GaussianModel (gaussian_model.py:160-161, :57)
        pass
    stats = Stats()
    stats.max_radii2D = torch.zeros(xyz.shape[0], device=dev)
    stats.xyz_gradient_accum = torch.zeros(xyz.shape[0], 1, device=dev)
    stats.denom = torch.zeros(xyz.shape[0], 1, device=dev)

    def stats_reference(vsp, radii):  # train.py:140-142 + add_densification_stats (gaussian_model.py:433-435), verbatim
        vis = radii > 0
        stats.max_radii2D[vis] = torch.max(stats.max_radii2D[vis], radii[vis])
        stats.xyz_gradient_accum[vis] += torch.norm(vsp.grad[vis, :2], dim=-1, keepdim=True)
        stats.denom[vis] += 1

    def stats_fused(vsp, radii):
        add_densification_stats(stats, vsp, radii)

    names = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")
    # arguments/__init__.py defaults x 1e-3: same arithmetic, but the scene (and with it R) stays put over the timed steps
    lrs = tuple(1e-3 * v for v in (0.00016, 0.0025, 0.0025 / 20.0, 0.05, 0.005, 0.001))
    saved = [p.detach().clone() for p in params]
    groups = lambda: [{"params": [p], "lr": lr, "name": n} for p, lr, n in zip(params, lrs, names)]
    torch_adam = torch.optim.Adam(groups(), lr=0.0, eps=1e-15)  # scene/gaussian_model.py:175
    hip_adam = HipAdam(groups(), lr=0.0, eps=1e-15)
    hip_l = lambda im: hip_loss.l1_ssim_loss(im, gt, 0.2)
    # the reference fork's own loss (train.py:110-117): L1 on the appearance-decoupled image, SSIM on the rendered one
    from based_on_point_cloud_partitions_amd.appearance import decouple_appearance
    from based_on_point_cloud_partitions_amd.synthetic import decouple_appearance_torch, make_appearance_model
    app = make_appearance_model(n_cameras=len(rasterizers), seed=3, device=dev)
    app_params = list(app.appearance_network.parameters()) + [app._appearance_embeddings]
    app_groups = lambda: [{"params": list(app.appearance_network.parameters()), "lr": 1e-6, "name": "appearance_network"},
                          {"params": [app._appearance_embeddings], "lr": 1e-6, "name": "appearance_embeddings"}]
    torch_adam_app = torch.optim.Adam(groups() + app_groups(), lr=0.0, eps=1e-15)
    hip_adam_app = HipAdam(groups() + app_groups(), lr=0.0, eps=1e-15)
    view = [0]

    def app_torch_l(im):  # stock torch ops throughout
        dec, _ = decouple_appearance_torch(im, app, view[0])
        return torch_l1_ssim(im, gt, 0.2, l1_image=dec)

    def app_mixed_l(im):  # torch network, HIP loss
        dec, _ = decouple_appearance_torch(im, app, view[0])
        return hip_loss.l1_ssim_loss(im, gt, 0.2, l1_image=dec)

    def app_hip_l(im):  # HIP tail of the network + HIP loss
        dec, _ = decouple_appearance(im, app, view[0])
        return hip_loss.l1_ssim_loss(im, gt, 0.2, l1_image=dec)

    variants = (
        ("raster_from_raw_params_reference_dataflow", render_reference, None, None, None),
        ("raster_from_raw_params_fused", render_fused, None, None, None),
        ("step_with_loss_reference_dataflow_torch_loss", render_reference, lambda im: torch_l1_ssim(im, gt, 0.2), None, None),
        ("step_with_loss_reference_dataflow_hip_loss", render_reference, hip_l, None, None),
        ("step_with_loss_fused_params_hip_loss", render_fused, hip_l, None, None),
        ("train_iteration_reference_dataflow_torch_loss_torch_adam_torch_stats", render_reference,
         lambda im: torch_l1_ssim(im, gt, 0.2), torch_adam, stats_reference),
        ("train_iteration_hip_rasterizer_torch_activations_hip_loss_torch_adam_torch_stats", render_reference, hip_l, torch_adam,
         stats_reference),
        ("train_iteration_fused_params_hip_loss_hip_adam_hip_stats", render_fused, hip_l, hip_adam, stats_fused),
        ("train_iteration_with_appearance_reference_dataflow_all_torch", render_reference, app_torch_l, torch_adam_app, stats_reference),
        ("train_iteration_with_appearance_fused_params_torch_network_hip_loss_hip_adam_hip_stats", render_fused, app_mixed_l, hip_adam_app,
         stats_fused),
        ("train_iteration_with_appearance_fused_params_hip_tail_hip_loss_hip_adam_hip_stats", render_fused, app_hip_l, hip_adam_app,
         stats_fused),
    )
    res = {}
    for name, render, loss, opt, upd in variants:
        with torch.no_grad():
            for p, q in zip(params, saved):
                p.copy_(q)

        def one(i):
            view[0] = i % len(rasterizers)
            image, vsp, radii = render(rasterizers[i % len(rasterizers)])
            if loss is None:
                image.backward(dL)
            else:
                loss(image).backward()
            if upd is not None:
                with torch.no_grad():
                    upd(vsp, radii)
            if opt is not None:
                opt.step()  # train.py:155-157
                opt.zero_grad(set_to_none=True)
            else:
                for p in params + app_params:
                    p.grad = None
        for i in range(4):
            one(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            one(4 + i)
        torch.cuda.synchronize()
        res[name + "_ms"] = round(1e3 * (time.perf_counter() - t0) / args.steps, 4)
    # the optimizer step alone, on the gradients of the last iteration's shape
    for name, opt in (("adam_step_torch", torch_adam), ("adam_step_hip", hip_adam)):
        for p in params:
            p.grad = torch.full_like(p, 1e-4)
        for _ in range(3):
            opt.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            opt.step()
        torch.cuda.synchronize()
        res[name + "_ms"] = round(1e3 * (time.perf_counter() - t0) / args.steps, 4)
        for p in params:
            p.grad = None
    # the appearance decoupling alone (forward + backward of an L1 on its output), stock torch vs the HIP tail
    img = torch.rand(3, H, W, device=dev, requires_grad=True)
    for name, fn in (("appearance_decoupling_torch", decouple_appearance_torch), ("appearance_decoupling_hip_tail", decouple_appearance)):
        def one_app(i):
            torch.abs(fn(img, app, i % len(rasterizers))[0] - gt).mean().backward()
        for i in range(4):
            one_app(i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            one_app(i)
        torch.cuda.synchronize()
        res[name + "_ms"] = round(1e3 * (time.perf_counter() - t0) / args.steps, 4)
    res["note"] = ("per-step ms from GaussianModel's raw parameters (exp/sigmoid/normalize/cat included): raster_* = forward + "
                   "backward with a fixed dL/dcolor; step_with_loss_* add (1-0.2)*L1 + 0.2*(1-SSIM); train_iteration_* add the densification statistics "
                   "(train.py:140-142) and the Adam step over the six parameter groups (29.5 M parameters); *_with_appearance_* use the fork's loss of "
                   "train.py:110-117 (appearance network, L1 on the decoupled image) and step the network's parameters too; context only, not the headline")
    return res


STEP_SOURCES = ("api.hip", "preprocess.hip", "binning.hip", "radix_sort.hip", "render_forward.hip", "render_backward.hip",
                "gaussian_backward.hip", "common.h")


def kernel_source_hash():
    """Identifies the kernel build a measurement belongs to: sha1 over the HIP sources of the raster step (they travel with the repo
    snapshot; kNN / loss / Adam / appearance sources do not enter: their kernels are not part of the step the PMC profile describes)."""
    import hashlib
    h = hashlib.sha1()
    d = os.path.join(ROOT, "based_on_point_cloud_partitions_amd", "csrc")
    for name in STEP_SOURCES:
        h.update(name.encode())
        h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:12]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=256)
    ap.add_argument("--warmup", type=int, default=8)
    ap.add_argument("--config", default="cfg2", choices=sorted(syn.CONFIGS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--order", default="as-generated", choices=["as-generated", "morton"],
                    help="memory order of the Gaussians: the workload's own (random, BASELINE.md) or sorted by the Morton code of xyz "
                         "(what a trainer gets by re-ordering its model at densification time; not the headline workload)")
    ap.add_argument("--no-stage-timers", action="store_true", help="time the region without per-stage HIP events")
    ap.add_argument("--dist", default="auto", choices=["auto", "on", "off", "early"],
                    help="process group at N = 1: auto = bring up a world-size-1 RCCL group (continue without one if it fails), "
                         "on = required, off = none, early = required and BEFORE the warm-up, with every barrier of an N > 1 run "
                         "(a one-GPU rehearsal of the N > 1 ordering); N > 1 always has one")
    ap.add_argument("--spinup-ms", type=float, default=6000.0,
                    help="upper limit of the untimed spin-up before the warm-up: batches of 64 steps run until four in a row bring no "
                         "improvement of 0.5 %% (at least 0.5 s).  A fresh box's GPU reaches its steady clocks only after a second or more of "
                         "sustained load: same process, same box, 0.737-0.741 ms/step with no spin-up, 0.739 with 0.4 s, 0.724 with 2 s "
                         "(render_backward 0.333 -> 0.327 ms); 0 switches it off")
    ap.add_argument("--full-step", action="store_true",
                    help="also time the step from raw parameters (activations fused or not) and with the L1/SSIM loss (train.py:107-118)")
    args = ap.parse_args()

    # Exactly ONE line may reach stdout (the JSON of rank 0), but libraries write there too (librccl prints a version banner when its
    # first communicator comes up): file descriptor 1 is pointed at stderr for the whole run and the JSON goes to the saved original.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:  # `python bench.py --gpus N` starts its own ranks (launch_ranks_if_needed); this is a launcher with another count
        raise SystemExit(f"--gpus {args.gpus} but the launcher set WORLD_SIZE={world}")
    ndev = max(torch.cuda.device_count(), 1)
    dev = torch.device("cuda", local_rank % ndev)  # one rank per GPU; the modulo only matters for 1-GPU rehearsals
    torch.cuda.set_device(dev)
    # The collective library: RCCL (torch's "nccl" backend on ROCm).  Two ranks cannot share one GPU under RCCL, so a rehearsal of
    # N ranks on fewer GPUs falls back to gloo and says so (`collective_backend`).  At N = 1 a world-size-1 RCCL group is brought up
    # as well (--dist auto) -- the same librccl init / all_gather_into_tensor / all_reduce / barrier entry points the N > 1 run
    # uses execute on this GPU, through the same `dist` branches below -- but only AFTER the timed region: there is no other rank to
    # align with, and the first process on a fresh box that initialises RCCL runs 5-10 % slower for the next seconds (librccl's
    # code objects page in behind it: 0.697 ms/step in the spin-up, 0.735 / 0.768 in the timed region right after, two boxes).
    dist_state = {"dist": None, "error": None, "ranks_seen": [0]}
    backend = os.environ.get("GSR_DIST_BACKEND", "nccl" if ndev >= world else "gloo")

    def init_dist():
        import datetime

        import torch.distributed as dist_
        if "RANK" not in os.environ:  # world 1 from a plain shell: be our own rendezvous
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ["MASTER_PORT"] = str(_free_port())
        try:
            kw = dict(rank=rank, world_size=world, timeout=datetime.timedelta(seconds=300))
            if backend == "nccl":
                dist_.init_process_group("nccl", device_id=dev, **kw)
            else:
                dist_.init_process_group(backend, **kw)
        except Exception as exc:  # noqa: BLE001 -- only tolerable at world 1, where no exchange is needed
            if world > 1 or args.dist in ("on", "early"):
                raise
            dist_state["error"] = f"{type(exc).__name__}: {exc}"
            return None
        # what the collective library itself saw: every rank contributes its number through one all-gather
        cdev = dev if dist_.get_backend() == "nccl" else "cpu"
        seen = torch.full((world,), -1, dtype=torch.int64, device=cdev)
        dist_.all_gather_into_tensor(seen, torch.tensor([rank], dtype=torch.int64, device=cdev))
        dist_state["ranks_seen"] = sorted(int(v) for v in seen.tolist())
        dist_state["dist"] = dist_
        return dist_

    dist = init_dist() if world > 1 or args.dist == "early" else None

    cfg = syn.CONFIGS[args.config]
    P, W, H = cfg["P"], cfg["W"], cfg["H"]
    if world == 1:
        g = syn.make_gaussians(P, seed=0)
        centre = (0.0, 0.0, 0.0)
    else:
        g, centre, _box = syn.make_partition(rank, world, P, seed=0, cell=PARTITION_CELL)
    if args.order == "morton":
        q = ((g.xyz - g.xyz.min(0).values) / (g.xyz.max(0).values - g.xyz.min(0).values + 1e-9) * 1023.0).to(torch.int64)

        def spread(v):  # 10 bits -> every third bit
            v = (v | (v << 16)) & 0x030000FF
            v = (v | (v << 8)) & 0x0300F00F
            v = (v | (v << 4)) & 0x030C30C3
            return (v | (v << 2)) & 0x09249249
        perm = torch.argsort(spread(q[:, 0]) | (spread(q[:, 1]) << 1) | (spread(q[:, 2]) << 2), stable=True)
        g = syn.SyntheticGaussians(*[t[perm].contiguous() for t in g])
    cams = syn.ring_cameras(cfg["cameras"], W, H, cfg["focal"], center=centre)
    dL_cpu = torch.randn(3, H, W, generator=torch.Generator().manual_seed(1))
    dL = dL_cpu.to(dev)
    bg = torch.zeros(3, device=dev)
    xyz = g.xyz.to(dev).requires_grad_(True)
    scales = g.scales.to(dev).requires_grad_(True)
    rots = g.rotations.to(dev).requires_grad_(True)
    opac = g.opacities.to(dev).requires_grad_(True)
    shs = g.shs.to(dev).requires_grad_(True)
    params = [xyz, scales, rots, opac, shs]
    rasterizers = [GaussianRasterizer(GaussianRasterizationSettings(
        image_height=H, image_width=W, tanfovx=c.tanfovx, tanfovy=c.tanfovy, bg=bg, scale_modifier=1.0,
        viewmatrix=c.world_view_transform.to(dev), projmatrix=c.full_proj_transform.to(dev), sh_degree=3,
        campos=c.camera_center.to(dev), prefiltered=False, debug=False)) for c in cams]

    def step(i):
        r = rasterizers[i % len(rasterizers)]
        screenspace = torch.zeros_like(xyz, requires_grad=True)  # receives dL/dmeans2D (gaussian_renderer/__init__.py:26)
        image, radii = r(means3D=xyz, means2D=screenspace, opacities=opac, shs=shs, scales=scales, rotations=rots)
        image.backward(dL)
        for p in params:
            p.grad = None
        return radii

    # untimed statistics pass: V and R per camera (every byte / flop figure is a function of them)
    # The workload is described in the reference's terms (its tile lists: R instances, list lengths, n_contrib), so this pass
    # builds the reference's lists (GSR_CULL=0); `listed` is what the product's default lists hold (instances the splat cannot
    # reach are left out, api.hip gsr_cull_enabled).
    stats = []
    with torch.no_grad():
        from based_on_point_cloud_partitions_amd import rasterizer as rz
        e = torch.Tensor([])
        cull = _lib.get_variant("CULL", 1)
        for c, r in zip(cams, rasterizers):
            s = r.raster_settings
            per = {}
            for mode in (0, cull):
                _lib.set_variant("CULL", mode)
                R, _, radii, gb, bb, ib = rz.rasterize_gaussians_forward(s.bg, xyz, e, opac, scales, rots, 1.0, e, s.viewmatrix,
                                                                         s.projmatrix, s.tanfovx, s.tanfovy, H, W, shs, 3, s.campos,
                                                                         False, False)
                ex = rz.debug_export(P, R, W, H, gb, bb, ib)
                ln = (ex["ranges"][:, 1] - ex["ranges"][:, 0]).float()
                if mode == 0:
                    per = dict(V=int((radii > 0).sum()), R=int(R), tile_mean=float(ln.mean()), tile_max=int(ln.max()),
                               n_contrib_mean=float(ex["n_contrib"].float().mean()))
                per["listed"] = int(ln.sum())
                del gb, bb, ib, ex
            stats.append(per)
        _lib.set_variant("CULL", cull)
    torch.cuda.empty_cache()

    # Per-stage HIP events cost a few microseconds of stream time per pair (16 pairs per step added ~9 % to the step), so:
    #  - the untimed warm-up runs with all of them and tells which stage dominates;
    #  - the timed region carries events around that stage only (its duration feeds `roofline`);
    #  - an untimed pass after the timed region, all events on, fills in the per-stage table (`roofline_kernels`).
    use_timers = not args.no_stage_timers
    # the interpreter's cyclic collector is kept out of the timed region (a generation-2 pass over torch's heap is a multi-millisecond
    # host stall); everything is collected here, before the warm-up
    import gc
    gc.collect()
    _lib.profile_enable(False)
    _lib.profile_collect()
    _lib.profile_enable(use_timers)
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    _lib.profile_enable(False)
    warm = _lib.profile_collect() if use_timers else {}
    dominant = max(warm, key=lambda k: warm[k][0] / max(warm[k][1], 1)) if any(n for _, n in warm.values()) else "render_backward"
    # The same K steps timed cold, right behind the W warm-up steps and before any spin-up (`ms_per_step_cold`): what a caller who
    # just started sees.  They are untimed as far as the headline goes (counted in `untimed_steps`).
    torch.cuda.synchronize()
    tc0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    torch.cuda.synchronize()
    cold_ms = 1e3 * (time.perf_counter() - tc0) / args.steps
    untimed_steps = args.warmup + args.steps
    # Spin-up (untimed, beyond the W warm-up steps): a GPU that has idled -- a fresh box, but also the tens of milliseconds of host
    # work between two phases of this script -- runs its first ~30 ms of work at lower clocks.  Measured in one process: batches of 64
    # steps settle at 0.700 ms/step, a 48-step region timed right after an idle gap reads 0.735.  So batches of 64 steps run until four in
    # a row bring no improvement of 0.5 % (at least 0.5 s, at most --spinup-ms), and the timed region starts right behind the last one.
    spinup = None
    # N > 1: at least 2 s -- the process group came up before the warm-up, and the seconds behind librccl's first use are slow on a
    # fresh box (see init_dist above); the collectives the timed region uses are warmed here too
    min_spin = 0.5 if dist is None else min(2.0, 1e-3 * args.spinup_ms)
    if dist is not None:
        for _ in range(3):
            dist.barrier()
    if args.spinup_ms > 0:
        t_spin, k, best, flat, batches = time.perf_counter(), 0, float("inf"), 0, []
        while 1e3 * (time.perf_counter() - t_spin) < args.spinup_ms:
            tb0 = time.perf_counter()
            for _ in range(64):
                step(k)
                k += 1
            torch.cuda.synchronize()
            dt = (time.perf_counter() - tb0) / 64
            batches.append(round(1e3 * dt, 4))
            if dt < 0.995 * best:
                best, flat = dt, 0
            else:
                flat += 1
            if flat >= 4 and time.perf_counter() - t_spin > min_spin:
                break
        spinup = {"steps": k, "ms": round(1e3 * (time.perf_counter() - t_spin), 1), "first_batch_ms_per_step": batches[0],
                  "last_batch_ms_per_step": batches[-1]}
        untimed_steps += k
        if dist is not None:  # ranks leave their spin-up at different times: align, then 32 more steps so that nobody starts cold
            dist.barrier()
            for i in range(32):
                step(i)
            untimed_steps += 32
    # The timed region's own code paths once, untimed: it is the first place where the dominant stage alone carries HIP events, and in
    # the first process on a fresh box code that runs for the first time is paged in from the image (two of two collection runs that
    # started with this script read 0.70 ms/step behind spin-up batches at 0.648; later processes on the same box never did)
    warm_marks = []
    if use_timers:
        # (as many steps as the timed region: its 2 K events are alive at once, and the runtime's event / signal pools grow in chunks --
        # with 32 steps here the first process on a box still showed two host-side gaps of 1 and 6 ms inside the timed region)
        _lib.profile_enable(True, stages=[dominant])
        tw = time.perf_counter()
        for i in range(args.steps):
            step(i)
            if (i & 31) == 31:
                warm_marks.append(time.perf_counter())
        torch.cuda.synchronize()
        warm_marks = [round(1e3 * (b - a), 2) for a, b in zip([tw] + warm_marks[:-1], warm_marks)]
        _lib.profile_enable(False)
        _lib.profile_collect()
        for i in range(64):  # and back to steady state
            step(i)
        torch.cuda.synchronize()
        untimed_steps += args.steps + 64
    _lib.profile_enable(use_timers, stages=[dominant])
    gc.disable()
    # Timed region: K steps between (barrier + device sync) and (device sync + barrier).  Each rank reads its clock right after its own
    # device sync; the closing barrier follows and `elapsed` becomes the MAX over the ranks of those local times -- from the common
    # start to the moment the slowest rank's GPU finished.  The closing barrier's own latency (a host round trip through the collective
    # library, not work of the path) is reported beside it (`closing_barrier_ms`, `ms_per_step_incl_closing_barrier`).
    tb = time.perf_counter()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    t0_wall = time.time()  # one node: the ranks share this clock, so the spread of these stamps is the skew the opening barrier left
    opening_barrier_ms = 1e3 * (t0 - tb)
    host_marks = []  # host-side progress (no sync): where the enqueueing thread stood every 32 steps -- a stall shows as a jump
    for i in range(args.steps):
        step(args.warmup + i)
        if (i & 31) == 31:
            host_marks.append(time.perf_counter())
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        dist.barrier()
    elapsed_incl_barrier = time.perf_counter() - t0
    gc.enable()
    _lib.profile_enable(False)
    timed = _lib.profile_collect() if use_timers else {}
    stage = {}
    if use_timers:
        _lib.profile_enable(True)
        for i in range(min(args.steps, 16)):
            step(args.warmup + args.steps + i)
        torch.cuda.synchronize()
        _lib.profile_enable(False)
        stage = _lib.profile_collect()
        if timed.get(dominant, (0.0, 0))[1] > 0:
            stage[dominant] = timed[dominant]  # the dominant stage as measured inside the timed region
    # Per-rank figures (the reference's launcher reports per partition, auto_train.py:22-51): every rank's own elapsed time, its wait at
    # the opening barrier, its start stamp, its workload (V / R / listed instances, means over the cameras the timed steps used) and
    # the dominant stage's average -- all-gathered so that rank 0's line can say WHICH rank set the MAX and why.
    used_local = [stats[(args.warmup + i) % len(stats)] for i in range(args.steps)]
    def _stage_avg(name):  # as measured inside the timed region if the stage carried events there, else in the untimed all-stages pass
        for src in (timed, stage):
            ms, n = src.get(name, (0.0, 0))
            if n > 0:
                return ms, n
        return 0.0, 0
    dom_ms, dom_n = _stage_avg(dominant)
    bwd_ms, bwd_n = _stage_avg("render_backward")
    mine = [elapsed, opening_barrier_ms, t0_wall, float(np.mean([u["V"] for u in used_local])), float(np.mean([u["R"] for u in used_local])),
            float(np.mean([u["listed"] for u in used_local])), float(np.mean([u["tile_mean"] for u in used_local])),
            float(max(u["tile_max"] for u in used_local)), dom_ms / max(dom_n, 1), bwd_ms / max(bwd_n, 1), cold_ms,
            1e3 * (elapsed_incl_barrier - elapsed), float(spinup["last_batch_ms_per_step"]) if spinup else 0.0, float(dev.index or 0),
            float(_lib.STAGES.index(dominant))]
    all_ranks = [mine]
    if dist is not None:
        cdev = dev if dist.get_backend() == "nccl" else "cpu"
        t = torch.tensor([elapsed, elapsed_incl_barrier], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed, elapsed_incl_barrier = (float(v) for v in t.tolist())
        gathered = torch.zeros(world * len(mine), dtype=torch.float64, device=cdev)
        dist.all_gather_into_tensor(gathered, torch.tensor(mine, dtype=torch.float64, device=cdev))
        all_ranks = gathered.reshape(world, len(mine)).tolist()
    per_rank = per_rank_report(all_ranks, args.steps, dominant)

    # the same steps over the REFERENCE's tile lists (GSR_CULL=0: sort keys, point_list, ranges, n_contrib bit for bit as the
    # reference builds them; the default lists leave out the instances a splat cannot reach) -- reported beside the headline
    ref_lists_ms = None
    if _lib.get_variant("CULL", 1) and _lib.get_variant("BIN", 1):
        _lib.set_variant("CULL", 0)
        for i in range(min(args.warmup, 4)):
            step(i)
        torch.cuda.synchronize()
        tr = time.perf_counter()
        for i in range(args.steps):
            step(args.warmup + i)
        torch.cuda.synchronize()
        ref_lists_ms = 1e3 * (time.perf_counter() - tr) / args.steps
        _lib.set_variant("CULL", 1)

    # scene initialisation kernel of the path (distCUDA2, once per sub-scene): reported next to the step, not part of it
    from based_on_point_cloud_partitions_amd import distCUDA2
    knn_evals = torch.zeros(1, dtype=torch.int64, device=dev)
    _lib.knn_counter(knn_evals)  # one counted run: candidate distances evaluated (the work that bounds the kernel)
    distCUDA2(xyz.detach())
    torch.cuda.synchronize()
    _lib.knn_counter(None)
    knn_evals = int(knn_evals.item())
    tk = time.perf_counter()
    for _ in range(5):
        distCUDA2(xyz.detach())
    torch.cuda.synchronize()
    knn_ms = round(1e3 * (time.perf_counter() - tk) / 5, 3)

    if world == 1 and dist is None and (args.dist == "on" or (args.dist == "auto" and backend == "nccl")):
        # N = 1: the world-size-1 RCCL group comes up here -- behind the timed region and every other timing of this script -- and the
        # collectives of an N > 1 run follow through it: all-reduce (MAX) of the timing, merge all-gathers, evaluation all-reduce, barrier
        dist = init_dist()
        if dist is not None:
            t = torch.tensor([elapsed, elapsed_incl_barrier], dtype=torch.float64, device=dev if dist.get_backend() == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed, elapsed_incl_barrier = (float(v) for v in t.tolist())
            dist.barrier()
    # config 5 (BASELINE.json configs[4]): seamless merge of the partitions + evaluation of the merged scene, once, untimed w.r.t.
    # the headline.  Every rank crops its partition to its origin box (open sides where it has no neighbour), ONE all-gather
    # of the padded [maxN, 59] blocks (RCCL over xGMI at N > 1), np.unique-style dedup, then each rank renders its own
    # partition's camera 0 from the MERGED scene and scores it against the same view of its own un-cropped partition (what
    # "seamless" means: PSNR across the seams).  Reports what the collective saw, so a multi-GPU run documents itself.
    from based_on_point_cloud_partitions_amd import evaluate, merge
    boxes = []
    for r_ in range(world):
        bx0, bx1, by0, by1 = syn.partition_origin_box(r_, world, cell=PARTITION_CELL)
        boxes.append((bx0, by0, bx1, by1))
    with torch.no_grad():
        merge.dedup_first_lexicographic(torch.rand(4096, merge.ROW, device=dev))  # (loads the sort kernels: dedup_ms is the algorithm's time)
        raw_block = merge.pack_block(xyz.detach(), shs.detach()[:, :1], shs.detach()[:, 1:],
                                     torch.logit(opac.detach().clamp(1e-6, 1 - 1e-6)), torch.log(scales.detach()), rots.detach())
        views = [syn.ring_cameras(cfg["cameras"], W, H, cfg["focal"],
                                  center=(0.5 * (b[0] + b[2]), 0.5 * (b[1] + b[3]), 0.0) if world > 1 else centre)[0].to(dev)
                 for b in boxes]
        gts = [None] * world
        gts[rank] = evaluate.render_rows(raw_block, views[rank], bg).clamp(0.0, 1.0)
        ev = evaluate.seamless_merge_and_evaluate(raw_block, rank, boxes if world > 1 else [(None, None, None, None)], views, gts, bg)
    merged_rows = int(ev["merged"].shape[0])
    # the dedup again, three times, on this rank's own rows: ev["dedup_ms"] above is ONE call right behind the process group's first
    # collectives (at N = 1 right behind RCCL's init: host-side noise several times the algorithm's own time)
    with torch.no_grad():
        torch.cuda.synchronize()
        td = time.perf_counter()
        for _ in range(3):
            merge.dedup_first_lexicographic(raw_block)
        torch.cuda.synchronize()
        dedup_steady_ms = 1e3 * (time.perf_counter() - td) / 3
    del raw_block, gts

    full = None
    if args.full_step:
        full = full_step_timing(args, rasterizers, g, dL, H, W, dev)

    if rank == 0:
        used = [stats[(args.warmup + i) % len(stats)] for i in range(args.steps)]
        V = float(np.mean([u["V"] for u in used]))
        R = float(np.mean([u["R"] for u in used]))
        N = W * H
        key_bits = 32 + higher_msb(((W + 15) // 16) * ((H + 15) // 16))
        sb = stage_bytes(P, V, R, N, key_bits)
        step_bytes = float(sum(sb.values()))
        ms_per_step = 1e3 * elapsed / args.steps
        kernels = []
        for name, (ms, n) in stage.items():
            if n == 0:
                continue
            avg_ms = ms / n
            kernels.append({"kernel": name, "avg_ms": round(avg_ms, 4), "launches": n,
                            "algorithmic_MB": round(sb[name] / 1e6, 2),
                            # bytes the REFERENCE's data flow moves for this stage / this stage's time: an efficiency against the
                            # reference's algorithm, not a throughput of this chip (a stage that does the job with fewer bytes reads above
                            # the 6.29 TB/s copy rate here).  `counted_GBs` below is the throughput: PMC bytes / stage time
                            "reference_dataflow_GBs": round(sb[name] / 1e9 / (avg_ms * 1e-3), 1), "counted_GBs": None})
        # PMC figures (HBM bytes, VALU instruction counts) come from separate rocprofv3 --pmc passes recorded in
        # profiles/pmc_traffic.json together with the workload and the kernel-source hash they were measured on; they are merged
        # into this run's per-kernel lines only when both match, otherwise they are reported apart under `pmc_from_profile`.
        workload = (f"{args.config}: one partition of {P} random-init Gaussians per GPU, {W}x{H}, SH degree 3, "
                    f"{len(cams)} ring cameras, fixed N(0,1) dL/dcolor")
        build = kernel_source_hash()
        pmc, pmc_stale = {}, None
        try:
            pmc = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
            meta = pmc.pop("_meta", {})
            if meta.get("workload") != workload or meta.get("build") != build or not _lib.build_info().endswith(" ablate=0 bin_ablate=0 extra=[]"):
                pmc_stale = {"source": "profiles/pmc_traffic.json", "measured_on": meta,
                             "hbm_bytes": {k: v.get("hbm_bytes") for k, v in pmc.items()},
                             "valu_insts": {k: v.get("sq_insts_valu") for k, v in pmc.items() if "sq_insts_valu" in v}}
                pmc = {}
        except Exception:
            pass
        for k in kernels:
            v = pmc.get(k["kernel"], {})
            if "sq_insts_valu" in v:
                # SIMD cycles this kernel spends per VALU wave-instruction (1024 SIMDs, 2.4 GHz).  Measured issue costs on
                # gfx950 (tools/valu_cost.hip, profiles/r01_valu_cost.txt): plain VOP 2.4 cycles; v_cmp / v_cndmask / SGPR
                # operands / DPP 4.3; v_exp / v_rcp / v_permlane*_swap 8.4 -- a kernel whose average sits between the first
                # two is VALU-issue bound.
                k["valu_insts"] = int(v["sq_insts_valu"])
                k["simd_cycles_per_valu_inst"] = round(1024 * 2.4e9 * k["avg_ms"] * 1e-3 / v["sq_insts_valu"], 2)
                pl = price_list_cycles(v, issue_shares())
                if pl is not None:
                    k["price_list_cycles_per_valu_inst"] = round(pl, 2)
            if "hbm_bytes" in v:
                k["hbm_traffic_MB"] = round(v["hbm_bytes"] / 1e6, 1)
                k["counted_GBs"] = round(v["hbm_bytes"] / 1e9 / (k["avg_ms"] * 1e-3), 1)
                k["counted_frac_of_hbm_peak"] = round(k["counted_GBs"] / HBM_PEAK_GBS, 4)
        kernels.sort(key=lambda k: -k["avg_ms"])
        roofline = None
        if kernels:
            d = kernels[0]
            hbm = {"achieved": d["reference_dataflow_GBs"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(d["reference_dataflow_GBs"] / HBM_PEAK_GBS, 4),
                   "algorithmic_bytes": int(sb[d["kernel"]]), "traffic": pmc.get(d["kernel"], {}).get("hbm_bytes")}
            # `traffic` (and valu_insts) are NOT measured by this run: they are the PMC passes recorded in the profile file below,
            # attached only because its workload and kernel-source hash equal this run's
            src = (f"profiles/pmc_traffic.json (separate rocprofv3 --pmc passes; build {build}, same workload)"
                   if hbm["traffic"] is not None else None)
            if d["kernel"] in VALU_BOUND and "valu_insts" in d:
                # The blend kernels are pair loops limited by VALU instruction issue, not by HBM: the roofline they are
                # priced against is the chip's VALU issue rate.  `peak` = 1024 SIMDs x 2.4 GHz / 2 cycles per wave64 v_fma_f32
                # (MI355X_MICROARCH.md), `achieved` = VALU wave-instructions retired per second (PMC count / this run's average launch
                # duration).  `peak_measured_plain_fma` is what a loop of independent v_fma_f32 reaches here (2.4 cycles, tools/valu_cost.hip);
                # `frac_price_list` prices the kernel's measured instruction mix (price_list_cycles above): the share of the kernel's
                # SIMD time its VALU instructions need at their measured issue costs.  `salu` is the scalar pipe's share.
                rate = d["valu_insts"] / (d["avg_ms"] * 1e-3) / 1e9
                peak = 1024 * 2.4 / VALU_GUIDE_CYCLES        # the guide's figure (wave64 v_fma_f32 = 2 cycles)
                peak_meas = 1024 * 2.4 / VALU_PLAIN_CYCLES   # what a loop of independent v_fma_f32 reaches on this chip (2.4 cycles)
                pl = d.get("price_list_cycles_per_valu_inst")
                pv = pmc.get(d["kernel"], {})
                scalar = pv.get("sq_insts_salu", 0.0) + pv.get("sq_insts_branch", 0.0) + pv.get("sq_insts_smem", 0.0)
                roofline = {"kernel": d["kernel"], "bound": "valu_issue", "achieved": round(rate, 1), "peak": round(peak, 1),
                            "unit": "G wave-instr/s", "frac": round(rate / peak, 4),
                            "peak_measured_plain_fma": round(peak_meas, 1), "frac_of_measured_plain_peak": round(rate / peak_meas, 4),
                            "frac_price_list": round(rate / peak_meas * pl / VALU_PLAIN_CYCLES, 4) if pl else None,
                            "mean_cycles_per_inst_price_list": pl,
                            # the second issue budget: scalar instructions (SALU + branches + scalar memory) at the measured 1.75 ns each
                            "salu": ({"insts": int(scalar), "ns_per_inst": NS_SALU,
                                      "pipe_frac": round(scalar * NS_SALU * 1e-9 / 1024 / (d["avg_ms"] * 1e-3), 4)} if scalar else None),
                            "valu_insts": d["valu_insts"], "simd_cycles_per_valu_inst": d["simd_cycles_per_valu_inst"],
                            "traffic": hbm["traffic"], "traffic_source": src, "avg_ms": d["avg_ms"], "hbm": hbm,
                            "note": "dominant kernel by time; VALU-issue-bound pair loop -- its HBM figures are under `hbm` (low by "
                                    "construction); roofline_kernels has the streaming stages, roofline_step the whole step"}
            else:
                roofline = {"kernel": d["kernel"], "bound": "hbm", "achieved": hbm["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
                            "frac": hbm["frac"], "traffic": hbm["traffic"], "traffic_source": src, "avg_ms": d["avg_ms"],
                            "algorithmic_bytes": hbm["algorithmic_bytes"],
                            "note": ("dominant kernel by time" + ("; a VALU-issue-bound pair loop (DESIGN.md §5) whose instruction count is "
                                     "not on file for this build (profiles/pmc_traffic.json is from another one): HBM figures only"
                                     if d["kernel"] in VALU_BOUND else ""))}
        for k in kernels:
            k["bound"] = "valu_issue" if k["kernel"] in VALU_BOUND else "hbm"
        fwd_ms = sum(k["avg_ms"] for k in kernels if k["kernel"] in
                     ("preprocess", "scan", "emit_keys", "radix_sort", "tile_ranges", "render_forward"))
        out = {
            "metric": "train iters/sec (rasterizer forward+backward step), 500k-Gaussian partition @1080p",
            "value": round(world * args.steps / elapsed, 3), "unit": "it/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "build": build, "build_info": _lib.build_info(),
            "config": {"workload": workload + ("" if args.order == "as-generated" else f" [Gaussians in {args.order} order]"), "gaussians_per_gpu": P,
                       "image": [W, H], "sh_degree": 3, "parallelism": f"{world} independent partition(s), 1 per GPU"},
            "workload_stats": {"P": P, "V_mean": round(V, 1), "R_mean": round(R, 1), "N": N, "key_bits": key_bits,
                               "tile_list_mean": round(float(np.mean([u["tile_mean"] for u in used])), 1),
                               "tile_list_max": int(max(u["tile_max"] for u in used)),
                               "n_contrib_mean": round(float(np.mean([u["n_contrib_mean"] for u in used])), 1),
                               "instances_listed_mean": round(float(np.mean([u["listed"] for u in used])), 1),
                               "per_camera": [{"V": u["V"], "R": u["R"], "tile_list_mean": round(u["tile_mean"], 1),
                                               "tile_list_max": u["tile_max"], "listed": u["listed"]} for u in stats]},
            # everything that ran before the timed region: W warm-up steps, the K cold-timed steps, the spin-up (clock ramp of a GPU
            # that idled, see --spinup-ms) and, at N > 1, the 32 steps behind the aligning barrier
            "untimed_steps": untimed_steps, "spinup": spinup,
            "ms_per_step_cold": round(cold_ms, 4),  # the same K steps timed right after the W warm-up steps, no spin-up
            # host-side progress inside the timed region: ms the enqueueing thread needed for each 32 steps (no device sync)
            "timed_host_ms_per_32_steps": [round(1e3 * (b - a), 2) for a, b in zip([t0] + host_marks[:-1], host_marks)],
            "warm_pass_host_ms_per_32_steps": warm_marks,  # the same path run once, untimed, right before (see above)
            "ms_per_step_incl_closing_barrier": round(1e3 * elapsed_incl_barrier / args.steps, 4),
            "closing_barrier_ms": round(1e3 * (elapsed_incl_barrier - elapsed), 3), "opening_barrier_ms": round(opening_barrier_ms, 3),
            "ms_per_step_reference_lists": round(ref_lists_ms, 4) if ref_lists_ms is not None else None,
            # round 5: the forward logs which pixels blended each (list entry, quadrant) visit and the backward reads that instead of
            # re-deriving it (GSR_BLEND_LOG=0 switches it off, GSR_BLEND_LOG_LIST=n limits it to mean lists below n)
            "blend_log": {"GSR_BLEND_LOG": _lib.get_variant("BLEND_LOG", 1), "GSR_BLEND_LOG_LIST": _lib.get_variant("BLEND_LOG_LIST", 0)},
            "knn_init_ms": knn_ms,
            "knn_points_per_s": round(P / (knn_ms * 1e-3), 1),
            # the kNN is not bandwidth-bound (its ~120 B per point are read once; the candidate points come out of LDS): its work is the
            # candidate distances a query evaluates -- 8 flops + a 3-deep compare-and-swap each, from the boxes the query cannot reject --
            # so the figure is candidate evaluations per second against the fp32 vector peak, the HBM figure beside it
            "roofline_knn": {"bound": "valu", "candidate_evals": knn_evals, "candidate_evals_per_point": round(knn_evals / max(P, 1), 1),
                             "flops_per_eval": 8, "achieved": round(8 * knn_evals / (knn_ms * 1e-3) / 1e12, 3), "peak": FP32_VALU_PEAK_TFLOPS,
                             "unit": "TFLOP/s", "frac": round(8 * knn_evals / (knn_ms * 1e-3) / 1e12 / FP32_VALU_PEAK_TFLOPS, 4),
                             "hbm": {"algorithmic_bytes": 120 * P, "achieved": round(120 * P / 1e9 / (knn_ms * 1e-3), 1), "peak": HBM_PEAK_GBS,
                                     "unit": "GB/s", "frac": round(120 * P / 1e9 / (knn_ms * 1e-3) / HBM_PEAK_GBS, 4)},
                             "note": "whole distCUDA2 call (min/max, Morton codes, radix sort, gather, box AABBs, search); exact 3-NN: a wave is one "
                                     "64-point box of Morton-sorted queries, scans its own box first, then walks a four-level AABB tree outward "
                                     "(tops, groups, boxes, 16-point quarters) and scans from LDS what its lanes cannot reject (no workgroup barrier)"},
            **per_rank,
            "rccl_world_size": len(dist_state["ranks_seen"]), "rccl_ranks_seen": dist_state["ranks_seen"],
            "collective_backend": (dist.get_backend() if dist is not None else None), "collective_init_error": dist_state["error"],
            "collective_init": ("before the warm-up (N > 1: the timed region is bracketed by its barriers)" if world > 1 else
                                "before the warm-up (--dist early: the ordering of an N > 1 run on one GPU)" if args.dist == "early" else
                                "after the timed region (N = 1: no rank to align with; the later collectives -- all-reduce of the "
                                "timing, merge all-gather, evaluation all-reduce, barrier -- run through it)"),
            "merge": {"world_size": ev["world_size"], "rows_per_rank_after_crop": ev["counts"], "merged_rows": merged_rows,
                      "merge_allgather_ms": round(ev["merge_allgather_ms"], 3),
                      "allgather_bytes_per_rank": int(max(ev["counts"]) * merge.ROW * 4),
                      "dedup_ms": round(ev["dedup_ms"], 3), "dedup_ms_own_rows_steady": round(dedup_steady_ms, 3),
                      "views_scored": ev["views_scored"],
                      "seam_psnr_db": (round(ev["psnr"], 3) if np.isfinite(ev["psnr"]) else None), "seam_ssim": round(ev["ssim"], 6),
                      "note": "config 5, once after the timed region: crop -> all_gather_into_tensor of padded [maxN,59] blocks -> "
                              "first-occurrence lexicographic dedup; each rank scores its partition's camera 0 rendered from the merged "
                              "scene against its own un-cropped partition (PSNR None = identical images; at N > 1 the neighbour "
                              "partition is visible from that camera, so the seam figures mostly measure that, not the merge)"},
            "rendered_mpix_per_s_fwd": round(world * N / 1e6 / (fwd_ms * 1e-3), 1) if fwd_ms > 0 else None,
            "roofline": roofline,
            "roofline_step": {"bound": "hbm", "algorithmic_bytes": int(step_bytes),
                              "achieved": round(step_bytes / 1e9 / (ms_per_step * 1e-3), 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                              "frac": round(step_bytes / 1e9 / (ms_per_step * 1e-3) / HBM_PEAK_GBS, 4)},
            "roofline_kernels": kernels,
        }
        if pmc_stale is not None:
            out["pmc_from_profile"] = pmc_stale
        if full is not None:
            out["full_step"] = full
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"], pairs = cpu_baseline(cfg, g, cams, dL_cpu.numpy())
            # the blend kernels next to the fp32 vector peak.  Pair visits = (tile list entry, pixel) pairs the
            # reference's loop evaluates until the pixel is done (oracle count, camera 0); ~11 flops per visit (power) and ~30
            # more per blended pair in the forward, ~60 in the backward.
            by = {k["kernel"]: k for k in kernels}
            for name, per_visit, per_blend in (("render_forward", 11, 30), ("render_backward", 11, 60)):
                if name in by:
                    t = by[name]["avg_ms"] * 1e-3
                    fl = pairs["pair_visits"] * per_visit + pairs["blended_pairs"] * per_blend
                    by[name]["pair_loop"] = {"reference_pair_visits": pairs["pair_visits"], "blended_pairs": pairs["blended_pairs"],
                                             "reference_pair_visits_per_s": round(pairs["pair_visits"] / t, 1),
                                             "fp32_flops_est": int(fl), "fp32_vector_peak_frac": round(fl / t / 157.3e12, 4)}
        print(json.dumps(out), file=json_out, flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
