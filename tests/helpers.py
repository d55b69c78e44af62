"""Shared scene builders for the parity tests (oracle side = numpy, product side = torch)."""
import numpy as np
import torch

from based_on_point_cloud_partitions_amd import synthetic as syn


def scene(P, W, H, focal, seed=0, cam_index=0, n_cams=4, scale_lo=0.004, scale_hi=0.04, **kw):
    g = syn.make_gaussians(P, seed=seed, scale_lo=scale_lo, scale_hi=scale_hi, **kw)
    cam = syn.ring_cameras(n_cams, W, H, focal)[cam_index]
    return g, cam


def oracle_forward(g, cam, bg=(0.0, 0.0, 0.0), sh_degree=3, scale_modifier=1.0, colors_precomp=None, cov3D_precomp=None):
    from oracle import oracle

    kw = {}
    if colors_precomp is None:
        kw["shs"] = g.shs.numpy()
    else:
        kw["colors_precomp"] = np.asarray(colors_precomp)
    if cov3D_precomp is None:
        kw["scales"] = g.scales.numpy()
        kw["rotations"] = g.rotations.numpy()
    else:
        kw["cov3D_precomp"] = np.asarray(cov3D_precomp)
    return oracle.forward(g.xyz.numpy(), g.opacities.numpy(), viewmatrix=cam.world_view_transform.numpy(),
                          projmatrix=cam.full_proj_transform.numpy(), campos=cam.camera_center.numpy(),
                          bg=np.asarray(bg, np.float32), W=cam.image_width, H=cam.image_height, tanfovx=cam.tanfovx,
                          tanfovy=cam.tanfovy, sh_degree=sh_degree, scale_modifier=scale_modifier, **kw)


def snapshot_scene(fx):
    """Inputs stored in tests/golden/oracle_snapshot.npz -> (gaussians, camera, bg)."""
    g = syn.SyntheticGaussians(torch.tensor(fx["xyz"]), torch.tensor(fx["scales"]), torch.tensor(fx["rotations"]),
                               torch.tensor(fx["opacities"]), torch.tensor(fx["shs"].astype(np.float32)))
    W, H, tx, ty, fvx, fvy = fx["cam"]
    cam = syn.SyntheticCamera(int(W), int(H), float(fvx), float(fvy), float(tx), float(ty), torch.tensor(fx["viewmatrix"]),
                              torch.tensor(fx["projmatrix"]), torch.tensor(fx["campos"]))
    return g, cam, tuple(float(v) for v in fx["bg"])


def rel_err(a, b):
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-12)) if a.size else 0.0


# ---------------------------------------------------------------------------------------------
# product side (HIP through the C ABI).  Everything below needs a GPU.
# ---------------------------------------------------------------------------------------------
def hip_settings(cam, bg, sh_degree, scale_modifier=1.0, device="cuda", debug=False):
    from based_on_point_cloud_partitions_amd import GaussianRasterizationSettings

    return GaussianRasterizationSettings(
        image_height=cam.image_height, image_width=cam.image_width, tanfovx=cam.tanfovx, tanfovy=cam.tanfovy,
        bg=torch.tensor(bg, dtype=torch.float32, device=device), scale_modifier=scale_modifier,
        viewmatrix=cam.world_view_transform.to(device), projmatrix=cam.full_proj_transform.to(device), sh_degree=sh_degree,
        campos=cam.camera_center.to(device), prefiltered=False, debug=debug)


def hip_forward_raw(g, cam, bg=(0.0, 0.0, 0.0), sh_degree=3, scale_modifier=1.0, colors_precomp=None, cov3D_precomp=None,
                    device="cuda"):
    """Calls the `_C.rasterize_gaussians` counterpart and exports the intermediates."""
    from based_on_point_cloud_partitions_amd import rasterizer as rz

    e = torch.Tensor([])
    d = lambda t: t.to(device)
    st = hip_settings(cam, bg, sh_degree, scale_modifier, device)
    colors = e if colors_precomp is None else torch.as_tensor(colors_precomp, dtype=torch.float32).to(device)
    cov = e if cov3D_precomp is None else torch.as_tensor(cov3D_precomp, dtype=torch.float32).to(device)
    sh = d(g.shs) if colors_precomp is None else e
    scales = d(g.scales) if cov3D_precomp is None else e
    rots = d(g.rotations) if cov3D_precomp is None else e
    means3D, opac = d(g.xyz), d(g.opacities)
    R, color, radii, gb, bb, ib = rz.rasterize_gaussians_forward(
        st.bg, means3D, colors, opac, scales, rots, scale_modifier, cov, st.viewmatrix, st.projmatrix, st.tanfovx, st.tanfovy,
        st.image_height, st.image_width, sh, sh_degree, st.campos, False, False)
    ex = rz.debug_export(means3D.shape[0], R, cam.image_width, cam.image_height, gb, bb, ib)
    torch.cuda.synchronize()
    ctx = dict(st=st, means3D=means3D, colors=colors, opac=opac, scales=scales, rots=rots, cov=cov, sh=sh, R=R, radii=radii,
               gb=gb, bb=bb, ib=ib, color=color, ex=ex, sh_degree=sh_degree, scale_modifier=scale_modifier)
    return ctx


def hip_backward_raw(ctx, dL_dpix, capture=False):
    """gsr_backward through the operator's backward entry.  capture=True also returns, under "sums", the [P,12] per-Gaussian blend
    sums the kernel's own chain consumed (gsr_debug_capture_sums) -- the input of the second stage of the gradient check."""
    from based_on_point_cloud_partitions_amd import _lib
    from based_on_point_cloud_partitions_amd import rasterizer as rz

    st = ctx["st"]
    dL = torch.as_tensor(dL_dpix, dtype=torch.float32).to(ctx["means3D"].device)
    sums = torch.zeros(ctx["means3D"].shape[0], 12, device=dL.device) if capture else None
    _lib.capture_sums(sums)
    try:
        out = rz.rasterize_gaussians_backward(
            st.bg, ctx["means3D"], ctx["radii"], ctx["colors"], ctx["scales"], ctx["rots"], ctx["scale_modifier"], ctx["cov"],
            st.viewmatrix, st.projmatrix, st.tanfovx, st.tanfovy, dL, ctx["sh"], ctx["sh_degree"], st.campos, ctx["gb"], ctx["R"],
            ctx["bb"], ctx["ib"], False)
        torch.cuda.synchronize()
    finally:
        _lib.capture_sums(None)
    names = ["dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations"]
    res = {k: v.cpu().numpy() for k, v in zip(names, out)}
    if capture:
        res["sums"] = sums.cpu().numpy()
    return res


def reference_lists():
    """True if the product builds the reference's tile lists bit for bit (GSR_CULL=0); by default it leaves out the instances
    whose tile the splat's alpha >= 1/255 ellipse cannot reach (api.hip, gsr_cull_enabled)."""
    from based_on_point_cloud_partitions_amd import _lib

    return not (_lib.get_variant("CULL", 1) and _lib.get_variant("BIN", 1))


def assert_forward_bit_exact(ctx, st, colors_precomp=None):
    """HIP forward vs oracle state: every integer output and (thanks to the shared arithmetic contract) every float.
    With culled tile lists (the default) the lists are checked as what they are -- the reference's lists minus instances that
    cannot contribute: a subsequence in the same order that contains every entry the oracle blends into a pixel of the tile --
    and n_contrib through the position map between the two lists."""
    ex = {k: v.cpu().numpy() for k, v in ctx["ex"].items()}
    vis = st["radii"] > 0
    assert ctx["R"] == st["num_rendered"]
    assert np.array_equal(ctx["radii"].cpu().numpy(), st["radii"])
    assert np.array_equal(ex["tiles_touched"].view(np.uint32), st["tiles_touched"])
    assert np.array_equal(ex["point_offsets"].view(np.uint32), st["point_offsets"])
    assert np.array_equal(ex["depths"].view(np.uint32)[vis], st["depths"].view(np.uint32)[vis])
    assert np.array_equal(ex["means2D"].view(np.uint32)[vis], st["means2D"].view(np.uint32)[vis])
    assert np.array_equal(ex["conic_opacity"].view(np.uint32)[vis], st["conic_opacity"].view(np.uint32)[vis])
    want_rgb = st["rgb"] if colors_precomp is None else np.asarray(colors_precomp, np.float32)  # precomputed colours are carried through
    assert np.array_equal(ex["rgb"].view(np.uint32)[vis], want_rgb.view(np.uint32)[vis])
    if reference_lists():
        assert np.array_equal(ex["keys_sorted"].view(np.uint64), st["keys_sorted"])
        assert np.array_equal(ex["point_list"].view(np.uint32), st["point_list"])
        assert np.array_equal(ex["ranges"].view(np.uint32), st["ranges"])
        assert np.array_equal(ex["n_contrib"].view(np.uint32), st["n_contrib"])
    else:
        _assert_culled_lists(ex, st)
    assert np.array_equal(ex["final_T"].view(np.uint32), st["final_T"].view(np.uint32))
    assert np.array_equal(ctx["color"].cpu().numpy().view(np.uint32), st["out_color"].view(np.uint32))
    return ex


def _assert_culled_lists(ex, st):
    from oracle import oracle

    W, H = st["W"], st["H"]
    gx, gy = (W + 15) // 16, (H + 15) // 16
    ref_rg, ref_pl, ref_keys = st["ranges"].astype(np.int64), st["point_list"], st["keys_sorted"]
    got_rg = ex["ranges"].view(np.uint32).astype(np.int64).reshape(-1, 2)
    got_pl, got_keys = ex["point_list"].view(np.uint32), ex["keys_sorted"].view(np.uint64)
    got_nc, ref_nc = ex["n_contrib"].view(np.uint32), st["n_contrib"]
    needed = oracle.block_flags(st) != 0  # list entries that are blended into at least one pixel of their tile
    kept_total = 0
    for tile in range(gx * gy):
        r0, r1 = ref_rg[tile]
        g0, g1 = got_rg[tile]
        n_ref, n_got = int(r1 - r0), int(g1 - g0)
        assert n_got <= n_ref
        if n_got == 0:
            assert not needed[r0:r1].any(), tile
            continue
        ids_ref, ids_got = ref_pl[r0:r1], got_pl[g0:g1]
        # ids are unique within a tile: position of every kept entry in the reference's list
        order = np.argsort(ids_ref, kind="stable")
        pos = order[np.searchsorted(ids_ref[order], ids_got)]
        assert np.array_equal(ids_ref[pos], ids_got), tile                     # a subset ...
        assert np.all(np.diff(pos) > 0), tile                                   # ... in the reference's order
        assert np.array_equal(got_keys[g0:g1], ref_keys[r0:r1][pos]), tile      # same (tile, depth) keys
        keep = np.zeros(n_ref, bool)
        keep[pos] = True
        assert not (needed[r0:r1] & ~keep).any(), tile                          # nothing that contributes was left out
        # n_contrib counts list positions: translate the culled list's positions into the reference's
        ty, tx = divmod(tile, gx)
        y0, x0 = ty * 16, tx * 16
        a = got_nc[y0:y0 + 16, x0:x0 + 16].astype(np.int64)
        b = ref_nc[y0:y0 + 16, x0:x0 + 16].astype(np.int64)
        assert a.max(initial=0) <= n_got
        mapped = np.where(a > 0, pos[np.maximum(a, 1) - 1] + 1, 0)
        assert np.array_equal(mapped, b), tile
        kept_total += n_got
    assert kept_total == int(got_rg[:, 1].max(initial=0))  # the culled ranges are contiguous: they end at the number of kept entries


# Gradient bar (BASELINE.json north_star: "within 1e-4 abs on rendered RGB/grad"), ELEMENT-WISE and in TWO STAGES (round 4; the
# one-stage bar of round 3 needed a `noise` term sampled from the oracle's own fp32 chain, i.e. it certified "within the oracle's own
# instability" on needle-shaped splats):
#   stage 1, the blend backward (K8): the nine per-Gaussian sums the HIP kernels reduce -- dL_dmean2D.xy, dL_dconic.xyw, dL_dopacity,
#     dL_dcolor.rgb; the conic sums are read through gsr_debug_capture_sums, the others are outputs -- against the oracle's sums
#     accumulated in double:      |got - want| <= GRAD_ATOL + GRAD_RTOL * |want| + GRAD_COND * cond
#     cond = the sum of the magnitudes of the element's terms (orc_render_backward_abs).  The HIP kernels sum in fp32 (a 64-lane tree,
#     the waves of a tile, then a Gaussian's tiles in order) over terms whose factors carry the hardware v_exp / v_rcp (<= 1e-6
#     relative), so their result may differ from a double sum by ~2e-6 * sum|terms| whatever the order;
#   stage 2, the per-Gaussian chain (K9 + K10: conic -> cov2D -> cov3D -> scale / rotation, mean projection, SH): the HIP outputs
#     against oracle.chain_backward evaluated ON THE HIP KERNEL'S OWN SUMS.  Same inputs, same fp32 expression order, no FMA contraction
#     on either side, correctly rounded division / sqrt on both: the outputs are BIT-IDENTICAL -- measured over the whole GPU suite
#     (292 tests: cfg2 / cfg3 / cfg4, 500:1 needles, the raw-parameter operator; gpurun_out/grad_worst.json) -- and the bar says so:
#     CHAIN_RTOL = 0, i.e. got == want for every element, 500:1 needles included.
# Measured (profiles/r04_grad_error.json, cfg2 camera 0; gpurun_out/grad_worst.json, whole suite): stage 1's worst |err| / tolerance was
# 0.076 on cfg2 and 0.124 over the suite with GRAD_COND = 4e-6; the coefficient is now 1e-6 (2x the measured worst if the tolerance of
# the worst element were its cond term alone).  No per-tensor scale, no blanket exception, no noise term.
GRAD_ATOL = 1e-4
GRAD_RTOL = 1e-4
GRAD_COND = 1e-6
CHAIN_RTOL = 0.0
GRAD_TOL = GRAD_ATOL
BLEND_KEYS = ("dL_dmeans2D", "dL_dconic", "dL_dopacity", "dL_dcolors")
CHAIN_KEYS = ("dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales", "dL_drotations")


def unpack_sums(sums12):
    """[P,12] rows of gsr_debug_capture_sums -> the oracle's blend-sum tensors (dL_dmeans2D [P,3], dL_dconic [P,4] = x, y, -, w, ...)."""
    s = np.asarray(sums12, np.float32)
    z = np.zeros(s.shape[0], np.float32)
    return dict(dL_dmeans2D=np.stack([s[:, 0], s[:, 1], z], 1), dL_dconic=np.stack([s[:, 2], s[:, 3], z, s[:, 4]], 1),
                dL_dopacity=np.ascontiguousarray(s[:, 5:6]), dL_dcolors=np.ascontiguousarray(s[:, 6:9]))


def _bar_stats(a, b, tol):
    """-> (worst |err| / tol, stats) for one tensor; raises through the caller's assert."""
    a, b, tol = a.astype(np.float64).reshape(-1), b.astype(np.float64).reshape(-1), np.asarray(tol, np.float64).reshape(-1)
    assert np.array_equal(np.isfinite(a), np.isfinite(b)), "non-finite values differ"
    fin = np.isfinite(b)
    err = np.abs(a[fin] - b[fin])
    ratio = err / tol[fin]
    i = int(np.argmax(ratio)) if ratio.size else 0
    st = dict(max_abs_err=float(err.max()) if err.size else 0.0, worst_ratio=float(ratio[i]) if ratio.size else 0.0,
              frac_within_1e_4_abs=float(np.mean(err <= 1e-4)) if err.size else 1.0, n=int(err.size),
              outside=int((ratio > 1).sum()), worst=(float(err[i]), float(tol[fin][i]), float(b[fin][i])) if ratio.size else None)
    return st


def assert_blend_sums_close(got, want, cond, atol=GRAD_ATOL, rtol=GRAD_RTOL, ccoef=GRAD_COND):
    """Stage 1: `got` / `want` / `cond` dicts over (a subset of) BLEND_KEYS.  Returns {tensor: stats}."""
    out = {}
    for k in got:
        if got[k].size == 0:
            continue
        b, c = want[k].astype(np.float64), np.asarray(cond[k], np.float64)
        assert got[k].shape == b.shape == c.shape, (k, got[k].shape, b.shape, c.shape)
        out[k] = _bar_stats(got[k], b, atol + rtol * np.abs(b) + ccoef * c)
        cover = np.mean(ccoef * c > atol + rtol * np.abs(b))  # share of elements whose bar is mostly the conditioning term
        out[k]["frac_cond_term_dominates"] = float(cover)
        e, t, w = out[k]["worst"]
        assert out[k]["outside"] == 0, (f"blend sums {k}: {out[k]['outside']} of {out[k]['n']} elements outside the bar; worst |err| {e:.3e} "
                                        f"vs tolerance {t:.3e} (want {w:.3e})")
    return out


def assert_chain_close(got, want, rtol=CHAIN_RTOL):
    """Stage 2: per-Gaussian chain outputs against the oracle's chain on the SAME sums.  Returns {tensor: stats}."""
    out = {}
    for k in got:
        if got[k].size == 0:
            continue
        b = want[k].astype(np.float64)
        assert got[k].shape == b.shape, (k, got[k].shape, b.shape)
        row = np.abs(np.where(np.isfinite(b), b, 0.0)).reshape(b.shape[0], -1).max(1).reshape((-1,) + (1,) * (b.ndim - 1))
        out[k] = _bar_stats(got[k], b, rtol * (np.abs(np.where(np.isfinite(b), b, 0.0)) + row) + 1e-30)
        e, t, w = out[k]["worst"]
        assert out[k]["outside"] == 0, (f"chain output {k}: {out[k]['outside']} of {out[k]['n']} elements outside {rtol:g} relative; worst "
                                        f"|err| {e:.3e} vs tolerance {t:.3e} (want {w:.3e})")
    return out


WORST = {}  # {stage:tensor: worst |err| / tolerance seen in this process}; tests/conftest.py writes it out at the end of a GPU session


def two_stage_check(st, got, dL, chain_of=None, want_blend=None):
    """The full check for one backward call.  `got`: the HIP outputs incl. "sums" (hip_backward_raw(..., capture=True)).
    chain_of(sums dict) -> dict of expected chain outputs keyed like `got` (default: oracle.chain_backward on the activated operator);
    want_blend: expected blend sums (default: the oracle's, in double).  Returns {tensor: stats} of both stages."""
    from oracle import oracle

    hs = unpack_sums(got["sums"])
    # the blend-level outputs ARE those sums (dL_dopacity only where the operator returns it before the sigmoid chain)
    for k in ("dL_dmeans2D", "dL_dcolors", "dL_dopacity"):
        if k in got and chain_of is None:
            assert np.array_equal(got[k], hs[k]), k
    want = oracle.blend_backward(st, dL) if want_blend is None else want_blend
    cond = oracle.blend_backward(st, dL, abs_terms=True)
    stats = {"blend:" + k: v for k, v in assert_blend_sums_close({k: hs[k] for k in want}, want, cond).items()}
    chain = oracle.chain_backward(st, hs) if chain_of is None else chain_of(hs)
    keys = [k for k in got if k in chain and k != "sums" and (chain_of is not None or k in CHAIN_KEYS)]
    stats.update({"chain:" + k: v for k, v in assert_chain_close({k: got[k] for k in keys}, chain).items()})
    for k, v in stats.items():
        WORST[k] = max(WORST.get(k, 0.0), v["worst_ratio"])
    return stats


def check_backward(ctx, st, dL, keys=None):
    """HIP backward vs oracle backward for one forward state, at the two-stage element-wise bar.  Returns {stage:tensor: worst
    |err| / tolerance}."""
    got = hip_backward_raw(ctx, dL, capture=True)
    stats = two_stage_check(st, got, dL)
    return {k: v["worst_ratio"] for k, v in stats.items()}
