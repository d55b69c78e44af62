"""ctypes/numpy front-end of the CPU oracle (oracle/gs_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
`cpu_baseline` leg of bench.py -- never by the product package.  Mirrors the
two-phase shape of the reference's Rasterizer::forward / ::backward
(submodules/diff-gaussian-rasterization/cuda_rasterizer/rasterizer_impl.cu:198-434)
and of simple-knn's distCUDA2 (submodules/simple-knn/spatial.cu:15-26), and
returns every intermediate so the HIP path can be compared stage by stage.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_f = ctypes.POINTER(ctypes.c_float)
_i = ctypes.POINTER(ctypes.c_int)
_u8 = ctypes.POINTER(ctypes.c_uint8)
_u32 = ctypes.POINTER(ctypes.c_uint32)
_u64 = ctypes.POINTER(ctypes.c_uint64)


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "gs_oracle.c")
    if force or not os.path.exists(so) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        _LIB = ctypes.CDLL(so)
        _LIB.orc_preprocess.restype = ctypes.c_long
        for fn in (_LIB.gs_expf, _LIB.gs_expf_blend):
            fn.restype = ctypes.c_float
            fn.argtypes = [ctypes.c_float]
        _LIB.orc_getHigherMsb.restype = ctypes.c_uint32
        _LIB.orc_getHigherMsb.argtypes = [ctypes.c_uint32]
    return _LIB


def _p(a, ty):
    if a is None:
        return ctypes.cast(None, ty)
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(ty)


def _f32(a):
    if a is None:
        return None
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float32))
    return a if a.size else None


def gs_expf(x, blend=False):
    """The two fully specified exp functions of the arithmetic contract: gs_expf (activations) or gs_expf_blend (alpha blending)."""
    L = lib()
    fn = L.gs_expf_blend if blend else L.gs_expf
    x = np.asarray(x, dtype=np.float32)
    out = np.empty_like(x)
    for k, v in enumerate(x.ravel()):
        out.ravel()[k] = fn(ctypes.c_float(float(v)))
    return out


def forward(means3D, opacities, *, shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None,
            viewmatrix, projmatrix, campos, bg, W, H, tanfovx, tanfovy, sh_degree=0, scale_modifier=1.0):
    """Full forward (K1..K7).  Returns a dict with out_color, radii and every intermediate."""
    L = lib()
    means3D = _f32(means3D).reshape(-1, 3) if np.asarray(means3D).size else np.zeros((0, 3), np.float32)
    P = means3D.shape[0]
    shs = _f32(shs); colors_precomp = _f32(colors_precomp); scales = _f32(scales)
    rotations = _f32(rotations); cov3D_precomp = _f32(cov3D_precomp)
    opacities = _f32(opacities)
    M = shs.reshape(P, -1, 3).shape[1] if shs is not None else 0
    vm = _f32(viewmatrix).reshape(16); pm = _f32(projmatrix).reshape(16)
    cp = _f32(campos).reshape(3); bgc = _f32(bg).reshape(3)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    st = dict(P=P, W=W, H=H, M=M, D=sh_degree)
    st["radii"] = np.zeros(P, np.int32)
    st["means2D"] = np.zeros((P, 2), np.float32)
    st["depths"] = np.zeros(P, np.float32)
    st["cov3D"] = np.zeros((P, 6), np.float32)
    st["rgb"] = np.zeros((P, 3), np.float32)
    st["conic_opacity"] = np.zeros((P, 4), np.float32)
    st["clamped"] = np.zeros((P, 3), np.uint8)
    st["tiles_touched"] = np.zeros(P, np.uint32)
    st["point_offsets"] = np.zeros(P, np.uint32)
    st["out_color"] = np.zeros((3, H, W), np.float32)
    st["final_T"] = np.zeros((H, W), np.float32)
    st["n_contrib"] = np.zeros((H, W), np.uint32)
    st["ranges"] = np.zeros((gx * gy, 2), np.uint32)
    R = 0
    if P > 0:
        R = L.orc_preprocess(
            ctypes.c_int(P), ctypes.c_int(sh_degree), ctypes.c_int(M), _p(means3D, _f), _p(scales, _f),
            ctypes.c_float(scale_modifier), _p(rotations, _f), _p(opacities, _f), _p(shs, _f), _p(cov3D_precomp, _f),
            _p(colors_precomp, _f), _p(vm, _f), _p(pm, _f), _p(cp, _f), ctypes.c_int(W), ctypes.c_int(H),
            ctypes.c_float(tanfovx), ctypes.c_float(tanfovy), _p(st["radii"], _i), _p(st["means2D"], _f),
            _p(st["depths"], _f), _p(st["cov3D"], _f), _p(st["rgb"], _f), _p(st["conic_opacity"], _f),
            _p(st["clamped"], _u8), _p(st["tiles_touched"], _u32), _p(st["point_offsets"], _u32))
    st["num_rendered"] = int(R)
    n = max(R, 1)
    st["keys_unsorted"] = np.zeros(n, np.uint64)[:R]
    st["values_unsorted"] = np.zeros(n, np.uint32)[:R]
    st["keys_sorted"] = np.zeros(n, np.uint64)[:R]
    st["point_list"] = np.zeros(n, np.uint32)[:R]
    st["stats"] = np.zeros(8, np.int64)  # visited, blended, 16x4 strips, touched entries, 8x8 heavy, sum max quarter, sum quarters, sum max half
    if P > 0:
        L.orc_binning(ctypes.c_int(P), ctypes.c_long(R), ctypes.c_int(W), ctypes.c_int(H), _p(st["radii"], _i),
                      _p(st["means2D"], _f), _p(st["depths"], _f), _p(st["point_offsets"], _u32),
                      _p(st["keys_unsorted"], _u64), _p(st["values_unsorted"], _u32), _p(st["keys_sorted"], _u64),
                      _p(st["point_list"], _u32), _p(st["ranges"], _u32))
        feats = colors_precomp if colors_precomp is not None else st["rgb"]
        L.orc_render_forward(ctypes.c_int(W), ctypes.c_int(H), _p(st["ranges"], _u32), _p(st["point_list"], _u32),
                             _p(st["means2D"], _f), _p(feats, _f), _p(st["conic_opacity"], _f), _p(bgc, _f),
                             _p(st["final_T"], _f), _p(st["n_contrib"], _u32), _p(st["out_color"], _f),
                             st["stats"].ctypes.data_as(ctypes.POINTER(ctypes.c_longlong)))
    # the reference fills out_color with 0 when P == 0 (rasterize_points.cu:68, :82)
    st["_inputs"] = dict(means3D=means3D, opacities=opacities, shs=shs, colors_precomp=colors_precomp, scales=scales,
                         rotations=rotations, cov3D_precomp=cov3D_precomp, vm=vm, pm=pm, cp=cp, bg=bgc,
                         tanfovx=tanfovx, tanfovy=tanfovy, scale_modifier=scale_modifier)
    return st


def block_flags(st):
    """uint8 per list entry (aligned with st["point_list"]): bit q set iff the entry is blended into a pixel of quadrant q
    (q = 2 * (y half) + (x half)) of its 16x16 tile -- the per-8x8-block work lists of the backward blend."""
    L = lib()
    R = st["num_rendered"]
    flags = np.zeros(max(R, 1), np.uint8)
    if R > 0:
        L.orc_block_flags(ctypes.c_int(st["W"]), ctypes.c_int(st["H"]), _p(st["ranges"], _u32), _p(st["point_list"], _u32),
                          _p(st["means2D"], _f), _p(st["conic_opacity"], _f), _p(flags, _u8))
    return flags[:R]


def blend_backward(st, dL_dpix, abs_terms=False):
    """K8 alone: the per-Gaussian blend-level sums dL_dmeans2D [P,3], dL_dconic [P,4] (x, y, -, w), dL_dopacity [P,1],
    dL_dcolors [P,3] of a state returned by forward(), accumulated in double.  abs_terms=True: the same sums with every term
    replaced by its magnitude (orc_render_backward_abs) -- the condition number of each sum for the tests' tolerance."""
    L = lib()
    P, W, H = st["P"], st["W"], st["H"]
    inp = st["_inputs"]
    dL_dpix = _f32(dL_dpix).reshape(3, H, W)
    g = dict(dL_dmeans2D=np.zeros((P, 3), np.float32), dL_dconic=np.zeros((P, 4), np.float32),
             dL_dopacity=np.zeros((P, 1), np.float32), dL_dcolors=np.zeros((P, 3), np.float32))
    if P == 0:
        return g
    colors = inp["colors_precomp"] if inp["colors_precomp"] is not None else st["rgb"]
    fn = L.orc_render_backward_abs if abs_terms else L.orc_render_backward
    fn(ctypes.c_int(P), ctypes.c_int(W), ctypes.c_int(H), _p(st["ranges"], _u32), _p(st["point_list"], _u32), _p(inp["bg"], _f),
       _p(st["means2D"], _f), _p(st["conic_opacity"], _f), _p(colors, _f), _p(st["final_T"], _f), _p(st["n_contrib"], _u32),
       _p(dL_dpix, _f), _p(g["dL_dmeans2D"], _f), _p(g["dL_dconic"], _f), _p(g["dL_dopacity"], _f), _p(g["dL_dcolors"], _f))
    return g


def backward(st, dL_dpix):
    """Full backward (K8..K10) for a state returned by forward()."""
    M = st["M"]
    P = st["P"]
    g = blend_backward(st, dL_dpix)
    if P == 0:
        g.update(dL_dmeans3D=np.zeros((P, 3), np.float32), dL_dcov3D=np.zeros((P, 6), np.float32),
                 dL_dsh=np.zeros((P, M, 3), np.float32), dL_dscales=np.zeros((P, 3), np.float32),
                 dL_drotations=np.zeros((P, 4), np.float32))
        return g
    return chain_backward(st, g)


def chain_backward(st, g):
    """K9 + K10 (the per-Gaussian chain) on given blend sums: g holds dL_dmeans2D [P,3], dL_dconic [P,4], dL_dcolors [P,3]
    (+ dL_dopacity, passed through); fills and returns the chain's outputs in a copy of g."""
    L = lib()
    P, W, H, M, D = st["P"], st["W"], st["H"], st["M"], st["D"]
    inp = st["_inputs"]
    g = {k: np.ascontiguousarray(v, dtype=np.float32) for k, v in g.items()}
    g.update(dL_dmeans3D=np.zeros((P, 3), np.float32), dL_dcov3D=np.zeros((P, 6), np.float32),
             dL_dsh=np.zeros((P, M, 3), np.float32), dL_dscales=np.zeros((P, 3), np.float32),
             dL_drotations=np.zeros((P, 4), np.float32))
    if P == 0:
        return g
    cov3D = inp["cov3D_precomp"] if inp["cov3D_precomp"] is not None else st["cov3D"]
    L.orc_preprocess_backward(
        ctypes.c_int(P), ctypes.c_int(D), ctypes.c_int(M), _p(inp["means3D"], _f), _p(st["radii"], _i),
        _p(inp["shs"], _f), _p(st["clamped"], _u8), _p(inp["scales"], _f), _p(inp["rotations"], _f),
        ctypes.c_float(inp["scale_modifier"]), _p(cov3D, _f), _p(inp["vm"], _f), _p(inp["pm"], _f), ctypes.c_int(W),
        ctypes.c_int(H), ctypes.c_float(inp["tanfovx"]), ctypes.c_float(inp["tanfovy"]), _p(inp["cp"], _f),
        _p(g["dL_dmeans2D"], _f), _p(g["dL_dconic"], _f), _p(g["dL_dmeans3D"], _f), _p(g["dL_dcolors"], _f),
        _p(g["dL_dcov3D"], _f), _p(g["dL_dsh"], _f), _p(g["dL_dscales"], _f), _p(g["dL_drotations"], _f))
    return g


def cov3d(scales, rotations, scale_modifier=1.0):
    """computeCov3D (forward.cu:118-152) for a batch: [P,6] upper triangle of R S S^T R^T (the quaternion is used as given)."""
    L = lib()
    sc, rot = _f32(scales).reshape(-1, 3), _f32(rotations).reshape(-1, 4)
    out = np.zeros((sc.shape[0], 6), np.float32)
    L.orc_cov3d(ctypes.c_int(sc.shape[0]), _p(sc, _f), ctypes.c_float(scale_modifier), _p(rot, _f), _p(out, _f))
    return out


def cov3d_backward(scales, rotations, dL_dcov3D, scale_modifier=1.0):
    """computeCov3D backward (backward.cu:278-341): (dL_dscales [P,3] w.r.t. scale_modifier * scale, dL_drotations [P,4] w.r.t. the
    quaternion as given, i.e. without a normalisation in front)."""
    L = lib()
    sc, rot, g = _f32(scales).reshape(-1, 3), _f32(rotations).reshape(-1, 4), _f32(dL_dcov3D).reshape(-1, 6)
    P = sc.shape[0]
    ds, dr = np.zeros((P, 3), np.float32), np.zeros((P, 4), np.float32)
    L.orc_cov3d_backward(ctypes.c_int(P), _p(sc, _f), ctypes.c_float(scale_modifier), _p(rot, _f), _p(g, _f), _p(ds, _f), _p(dr, _f))
    return ds, dr


def sh_backward(means3D, campos, shs, clamped, dL_dcolor, sh_degree):
    """computeColorFromSH backward (backward.cu:20-139) alone: (dL_dsh [P,M,3], the view-direction part of dL_dmeans3D [P,3])."""
    L = lib()
    m, sh, g = _f32(means3D).reshape(-1, 3), _f32(shs), _f32(dL_dcolor).reshape(-1, 3)
    P = m.shape[0]
    sh = sh.reshape(P, -1, 3)
    cl = np.ascontiguousarray(np.asarray(clamped, np.uint8).reshape(P, 3))
    dsh, dm = np.zeros_like(sh), np.zeros((P, 3), np.float32)
    L.orc_sh_backward(ctypes.c_int(P), ctypes.c_int(sh_degree), ctypes.c_int(sh.shape[1]), _p(m, _f), _p(_f32(campos).reshape(3), _f),
                      _p(sh, _f), _p(cl, _u8), _p(g, _f), _p(dm, _f), _p(dsh, _f))
    return dsh, dm


def activate(features_dc, features_rest, opacity_raw, scaling_raw, rotation_raw):
    """GaussianModel getters (scene/gaussian_model.py:102-122): returns dict(shs, opacities, scales, rotations)."""
    L = lib()
    f_dc = np.ascontiguousarray(np.asarray(features_dc, np.float32))
    P = f_dc.reshape(-1, 3).shape[0]
    f_rest = np.ascontiguousarray(np.asarray(features_rest, np.float32)).reshape(P, -1, 3)
    Mrest = f_rest.shape[1]
    o = np.ascontiguousarray(np.asarray(opacity_raw, np.float32)).reshape(P, 1)
    sc = np.ascontiguousarray(np.asarray(scaling_raw, np.float32)).reshape(P, 3)
    rot = np.ascontiguousarray(np.asarray(rotation_raw, np.float32)).reshape(P, 4)
    out = dict(shs=np.zeros((P, 1 + Mrest, 3), np.float32), opacities=np.zeros((P, 1), np.float32),
               scales=np.zeros((P, 3), np.float32), rotations=np.zeros((P, 4), np.float32))
    if P:
        L.orc_activate(ctypes.c_int(P), ctypes.c_int(Mrest), _p(f_dc, _f), _p(f_rest if Mrest else None, _f), _p(o, _f),
                       _p(sc, _f), _p(rot, _f), _p(out["shs"], _f), _p(out["opacities"], _f), _p(out["scales"], _f),
                       _p(out["rotations"], _f))
    return out


def activate_backward(opacity_raw, scaling_raw, rotation_raw, g):
    """Chain rule through activate() for the gradient dict `g` of backward(): gradients w.r.t. the raw parameters."""
    L = lib()
    P, M = g["dL_dsh"].shape[0], g["dL_dsh"].shape[1]
    Mrest = M - 1
    o = np.ascontiguousarray(np.asarray(opacity_raw, np.float32)).reshape(P, 1)
    sc = np.ascontiguousarray(np.asarray(scaling_raw, np.float32)).reshape(P, 3)
    rot = np.ascontiguousarray(np.asarray(rotation_raw, np.float32)).reshape(P, 4)
    out = dict(dL_dfeatures_dc=np.zeros((P, 1, 3), np.float32), dL_dfeatures_rest=np.zeros((P, Mrest, 3), np.float32),
               dL_dopacity_raw=np.zeros((P, 1), np.float32), dL_dscaling_raw=np.zeros((P, 3), np.float32),
               dL_drotation_raw=np.zeros((P, 4), np.float32), dL_dxyz=g["dL_dmeans3D"], dL_dmeans2D=g["dL_dmeans2D"])
    if P:
        L.orc_activate_backward(ctypes.c_int(P), ctypes.c_int(Mrest), _p(o, _f), _p(sc, _f), _p(rot, _f),
                                _p(np.ascontiguousarray(g["dL_dsh"]), _f), _p(np.ascontiguousarray(g["dL_dopacity"]), _f),
                                _p(np.ascontiguousarray(g["dL_dscales"]), _f), _p(np.ascontiguousarray(g["dL_drotations"]), _f),
                                _p(out["dL_dfeatures_dc"], _f), _p(out["dL_dfeatures_rest"] if Mrest else None, _f),
                                _p(out["dL_dopacity_raw"], _f), _p(out["dL_dscaling_raw"], _f), _p(out["dL_drotation_raw"], _f))
    return out


def adam_step(p, g, m, v, lr, beta1=0.9, beta2=0.999, eps=1e-15, step=1):
    """One torch.optim.Adam update of a float32 array, in place on (p, m, v); `step` is the count after the update."""
    L = lib()
    for a in (p, g, m, v):
        assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"] and a.size == p.size
    L.orc_adam_step(ctypes.c_long(p.size), _p(p, _f), _p(g, _f), _p(m, _f), _p(v, _f), ctypes.c_double(lr), ctypes.c_double(beta1),
                    ctypes.c_double(beta2), ctypes.c_double(eps), ctypes.c_long(step))


def densification_stats(radii, viewspace_grad, max_radii2D, xyz_gradient_accum, denom):
    """train.py:140-142 + GaussianModel.add_densification_stats (scene/gaussian_model.py:433-435), numpy restatement, in place."""
    vis = np.asarray(radii) > 0
    g = np.asarray(viewspace_grad, np.float32)
    max_radii2D[vis] = np.maximum(max_radii2D[vis], np.asarray(radii)[vis].astype(np.float32))
    n = np.sqrt((g[:, 0] * g[:, 0] + g[:, 1] * g[:, 1]).astype(np.float32)).astype(np.float32)
    acc = xyz_gradient_accum.reshape(-1)
    acc[vis] = (acc[vis] + n[vis]).astype(np.float32)
    d = denom.reshape(-1)
    d[vis] = d[vis] + np.float32(1.0)


def mark_visible(means3D, viewmatrix, projmatrix):
    L = lib()
    means3D = _f32(means3D)
    P = 0 if means3D is None else means3D.reshape(-1, 3).shape[0]
    out = np.zeros(P, np.uint8)
    if P:
        L.orc_mark_visible(ctypes.c_int(P), _p(means3D, _f), _p(_f32(viewmatrix).reshape(16), _f),
                           _p(_f32(projmatrix).reshape(16), _f), _p(out, _u8))
    return out.astype(bool)


def knn_mean_dist2(points, bruteforce=False):
    """distCUDA2: mean squared distance to the 3 nearest other points."""
    L = lib()
    pts = _f32(points)
    P = 0 if pts is None else pts.reshape(-1, 3).shape[0]
    out = np.zeros(P, np.float32)
    if P:
        fn = L.orc_knn_bruteforce if bruteforce else L.orc_knn_mean_dist2
        fn(ctypes.c_int(P), _p(pts, _f), _p(out, _f))
    return out


def l1_ssim_loss(img1, img2, lambda_dssim=0.2, grad=True):
    """utils/loss_utils.py (l1_loss, ssim) combined as train.py:116-117. Returns (loss, l1, ssim, dloss/dimg1 or None)."""
    L = lib()
    a = _f32(img1); b = _f32(img2)
    C, H, W = a.shape
    out = np.zeros(3, np.float32)
    g = np.zeros_like(a) if grad else None
    L.orc_l1_ssim_loss(ctypes.c_int(C), ctypes.c_int(H), ctypes.c_int(W), _p(a, _f), _p(b, _f), ctypes.c_float(lambda_dssim),
                       _p(out, _f), _p(g, _f))
    return float(out[0]), float(out[1]), float(out[2]), g


# ------------------------------------------------------------------------------------------------------------------
# Appearance-decoupling network, full-resolution tail: scene/ptgs/appearance_network.py:52-60 from the
# bilinear upsampling on, and `transformed_image = mapping_image * image` (:15).  numpy fp32; pinned by
# tests/golden/ref_appearance.npz, which the reference's own module produced (values + autograd gradients).
# ------------------------------------------------------------------------------------------------------------------
def _bilinear_coords(n_in, n_out):
    """torch's upsample_bilinear2d with align_corners=True (appearance_network.py:54): source index, neighbour, weight."""
    scale = np.float32(n_in - 1) / np.float32(n_out - 1) if n_out > 1 else np.float32(0)
    src = scale * np.arange(n_out, dtype=np.float32)
    i0 = src.astype(np.int32)
    i1 = i0 + (i0 < n_in - 1)
    l1 = (src - i0.astype(np.float32)).astype(np.float32)
    return i0, i1, (np.float32(1) - l1).astype(np.float32), l1


def bilinear_upsample(x, H, W):
    y0, y1, wy0, wy1 = _bilinear_coords(x.shape[1], H)
    x0, x1, wx0, wx1 = _bilinear_coords(x.shape[2], W)
    top = wx0[None, None, :] * x[:, y0][:, :, x0] + wx1[None, None, :] * x[:, y0][:, :, x1]
    bot = wx0[None, None, :] * x[:, y1][:, :, x0] + wx1[None, None, :] * x[:, y1][:, :, x1]
    return (wy0[None, :, None] * top + wy1[None, :, None] * bot).astype(np.float32)


def _bilinear_upsample_adjoint(g, h, w):
    C, H, W = g.shape
    y0, y1, wy0, wy1 = _bilinear_coords(h, H)
    x0, x1, wx0, wx1 = _bilinear_coords(w, W)
    out = np.zeros((C, h, w), np.float64)
    for (yi, wy) in ((y0, wy0), (y1, wy1)):
        for (xi, wx) in ((x0, wx0), (x1, wx1)):
            contrib = g.astype(np.float64) * wy[None, :, None] * wx[None, None, :]
            np.add.at(out, (slice(None), yi[:, None], xi[None, :]), contrib)
    return out.astype(np.float32)


def _conv3x3(x, w, b):
    """Conv2d(k=3, stride 1, padding 1): x [Ci,H,W], w [Co,Ci,3,3], b [Co]."""
    Ci, H, W = x.shape
    xp = np.zeros((Ci, H + 2, W + 2), np.float32)
    xp[:, 1:-1, 1:-1] = x
    out = np.zeros((w.shape[0], H, W), np.float32)
    for dy in range(3):
        for dx in range(3):
            out += np.einsum("oc,chw->ohw", w[:, :, dy, dx], xp[:, dy:dy + H, dx:dx + W], optimize=True).astype(np.float32)
    return out + b[:, None, None]


def _conv3x3_backward(x, w, gz):
    """Gradients of _conv3x3 w.r.t. x, w, b given gz = dL/dout."""
    Ci, H, W = x.shape
    xp = np.zeros((Ci, H + 2, W + 2), np.float32)
    xp[:, 1:-1, 1:-1] = x
    gp = np.zeros((gz.shape[0], H + 2, W + 2), np.float32)
    gp[:, 1:-1, 1:-1] = gz
    gw = np.zeros_like(w)
    gx = np.zeros((Ci, H, W), np.float32)
    for dy in range(3):
        for dx in range(3):
            gw[:, :, dy, dx] = np.einsum("ohw,chw->oc", gz.astype(np.float64), xp[:, dy:dy + H, dx:dx + W].astype(np.float64))
            gx += np.einsum("oc,ohw->chw", w[:, :, dy, dx], gp[:, 2 - dy:2 - dy + H, 2 - dx:2 - dx + W], optimize=True).astype(np.float32)
    return gx, gw, gz.astype(np.float64).sum(axis=(1, 2)).astype(np.float32)


def appearance_tail_forward(feat, w2, b2, w3, b3, image):
    """feat = output of `up4` [16,h,w]; returns (mapping_image, transformed_image, saved) -- appearance_network.py:54-60,15."""
    feat, w2, b2, w3, b3, image = (np.ascontiguousarray(t, np.float32) for t in (feat, w2, b2, w3, b3, image))
    H, W = image.shape[1:]
    u = bilinear_upsample(feat, H, W)
    a = np.maximum(_conv3x3(u, w2, b2), np.float32(0))
    z = _conv3x3(a, w3, b3)
    m = (np.float32(1) / (np.float32(1) + np.exp(-z))).astype(np.float32)
    return m, (m * image).astype(np.float32), dict(u=u, a=a, m=m, feat_shape=feat.shape)


def appearance_tail_backward(saved, w2, w3, image, g_transformed, g_mapping=None):
    """Returns dict of gradients: feat, w2, b2, w3, b3, image (the direct `mapping * g` term only: the path through the
    down-sampled input of the network is upstream of `feat`)."""
    u, a, m = saved["u"], saved["a"], saved["m"]
    g = np.ascontiguousarray(g_transformed, np.float32)
    dm = g * image if g_mapping is None else g * image + g_mapping
    dz3 = (dm * m * (np.float32(1) - m)).astype(np.float32)
    da, gw3, gb3 = _conv3x3_backward(a, w3, dz3)
    dz2 = np.where(a > 0, da, np.float32(0)).astype(np.float32)
    du, gw2, gb2 = _conv3x3_backward(u, w2, dz2)
    _, h, w = saved["feat_shape"]
    return dict(feat=_bilinear_upsample_adjoint(du, h, w), w2=gw2, b2=gb2, w3=gw3, b3=gb3, image=(g * m).astype(np.float32))


def pixel_shuffle2(x):
    """torch.nn.PixelShuffle(2): [4c, h, w] -> [c, 2h, 2w], out[c][2y + i][2x + j] = x[4c + 2i + j][y][x]."""
    C4, h, w = x.shape
    return np.ascontiguousarray(x.reshape(C4 // 4, 2, 2, h, w).transpose(0, 3, 1, 4, 2).reshape(C4 // 4, 2 * h, 2 * w))


def _pixel_unshuffle2(g):
    C, H, W = g.shape
    return np.ascontiguousarray(g.reshape(C, H // 2, 2, W // 2, 2).transpose(0, 2, 4, 1, 3).reshape(4 * C, H // 2, W // 2))


def up_block_forward(x, w, b):
    """UpsampleBlock.forward (scene/ptgs/appearance_network.py:20-29): PixelShuffle(2) -> Conv2d(3x3, padding 1) -> ReLU."""
    x, w, b = (np.ascontiguousarray(t, np.float32) for t in (x, w, b))
    ps = pixel_shuffle2(x)
    return np.maximum(_conv3x3(ps, w, b), np.float32(0)), ps


def up_block_backward(ps, w, y, g):
    """Gradients of up_block_forward w.r.t. x, w, b given g = dL/dy (y = its output, for the ReLU mask)."""
    dz = np.where(y > 0, np.ascontiguousarray(g, np.float32), np.float32(0)).astype(np.float32)
    dps, gw, gb = _conv3x3_backward(ps, w, dz)
    return _pixel_unshuffle2(dps), gw, gb


def set_threads(n):
    """Thread count of the C restatement's OpenMP loops; returns the previous maximum."""
    prev = int(lib().orc_max_threads())
    lib().orc_set_threads(int(n))
    return prev

