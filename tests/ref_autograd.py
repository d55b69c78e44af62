"""Independent float64 re-derivation of the splatting equations with torch autograd.

Written from the maths (EWA splatting + front-to-back alpha compositing), NOT from
the oracle's code: used to check that the oracle's hand-written backward
(which follows backward.cu) really is the derivative of its forward.  The
discrete decisions (sorted per-tile lists, n_contrib) are taken from the oracle
run; everything differentiable is recomputed here in float64.
"""
import math

import torch

C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435]


def sh_basis(d, deg):
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    b = [torch.full_like(x, C0)]
    if deg > 0:
        b += [-C1 * y, C1 * z, -C1 * x]
    if deg > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        b += [C2[0] * xy, C2[1] * yz, C2[2] * (2 * zz - xx - yy), C2[3] * xz, C2[4] * (xx - yy)]
    if deg > 2:
        b += [C3[0] * y * (3 * xx - yy), C3[1] * xy * z, C3[2] * y * (4 * zz - xx - yy),
              C3[3] * z * (2 * zz - 3 * xx - 3 * yy), C3[4] * x * (4 * zz - xx - yy), C3[5] * z * (xx - yy),
              C3[6] * x * (xx - 3 * yy)]
    return torch.stack(b, dim=1)  # [P, K]


def quat_to_R(q):
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1).reshape(-1, 3, 3)
    return R


def render(xyz, scales, rots, opac, shs, ndc_off, *, viewmatrix, projmatrix, campos, bg, W, H, tanfovx, tanfovy, deg,
           scale_modifier, ranges, point_list, n_contrib, colors_precomp=None, cov3D_precomp=None):
    """All tensor args float64.  viewmatrix/projmatrix are the reference's transposed 4x4 matrices."""
    P = xyz.shape[0]
    V = viewmatrix.t()  # W2C
    F = projmatrix.t()  # full projection
    ones = torch.ones(P, 1, dtype=xyz.dtype)
    ph = torch.cat([xyz, ones], 1) @ F.t()
    pw = 1.0 / (ph[:, 3] + 0.0000001)
    ndc = ph[:, :2] * pw[:, None] + ndc_off
    pix = torch.stack([((ndc[:, 0] + 1.0) * W - 1.0) * 0.5, ((ndc[:, 1] + 1.0) * H - 1.0) * 0.5], 1)
    t = (torch.cat([xyz, ones], 1) @ V.t())[:, :3]
    if cov3D_precomp is None:
        R = quat_to_R(rots)
        S = torch.diag_embed(scale_modifier * scales)
        Sigma = R @ S @ S @ R.transpose(1, 2)
    else:
        c = cov3D_precomp
        Sigma = torch.stack([c[:, 0], c[:, 1], c[:, 2], c[:, 1], c[:, 3], c[:, 4], c[:, 2], c[:, 4], c[:, 5]], 1).reshape(-1, 3, 3)
    fx, fy = W / (2.0 * tanfovx), H / (2.0 * tanfovy)
    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    tz = t[:, 2]
    # backward.cu:166-172,259-264: outside the clamp the reference treats the clamped t.x / t.y as constants
    # (x_grad_mul = 0 and no d(t.x)/d(t.z) term), which is what .detach() reproduces here.
    inx = (t[:, 0] / tz).abs() <= limx
    iny = (t[:, 1] / tz).abs() <= limy
    tx = torch.where(inx, t[:, 0], (torch.clamp(t[:, 0] / tz, -limx, limx) * tz).detach())
    ty = torch.where(iny, t[:, 1], (torch.clamp(t[:, 1] / tz, -limy, limy) * tz).detach())
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tz, zero, -(fx * tx) / (tz * tz), zero, fy / tz, -(fy * ty) / (tz * tz)], 1).reshape(-1, 2, 3)
    Wr = V[:3, :3]
    Tm = J @ Wr
    cov = Tm @ Sigma @ Tm.transpose(1, 2)
    a = cov[:, 0, 0] + 0.3
    b = cov[:, 0, 1]
    c = cov[:, 1, 1] + 0.3
    det = a * c - b * b
    conic = torch.stack([c / det, -b / det, a / det], 1)
    if colors_precomp is None:
        d = xyz - campos[None]
        d = d / d.norm(dim=1, keepdim=True)
        K = (deg + 1) ** 2
        B = sh_basis(d, deg)
        rgb = torch.clamp_min((B[:, :, None] * shs[:, :K, :]).sum(1) + 0.5, 0.0)
    else:
        rgb = colors_precomp
    gx = (W + 15) // 16
    out = torch.zeros(3, H, W, dtype=xyz.dtype)
    rows = []
    for py in range(H):
        for px in range(W):
            tile = (py // 16) * gx + (px // 16)
            r0 = int(ranges[tile, 0])
            n = int(n_contrib[py, px])
            if n == 0:
                rows.append(bg.clone())
                continue
            ids = torch.as_tensor(point_list[r0:r0 + n].astype("int64"))
            dxy = pix[ids] - torch.tensor([float(px), float(py)], dtype=xyz.dtype)
            co = conic[ids]
            power = -0.5 * (co[:, 0] * dxy[:, 0] ** 2 + co[:, 2] * dxy[:, 1] ** 2) - co[:, 1] * dxy[:, 0] * dxy[:, 1]
            alpha = torch.clamp_max(opac[ids, 0] * torch.exp(power), 0.99)
            keep = (power <= 0) & (alpha >= 1.0 / 255.0)
            alpha = torch.where(keep, alpha, torch.zeros_like(alpha))
            Tcum = torch.cumprod(torch.cat([torch.ones(1, dtype=xyz.dtype), 1 - alpha]), 0)
            col = (rgb[ids] * (alpha * Tcum[:-1])[:, None]).sum(0) + Tcum[-1] * bg
            rows.append(col)
    img = torch.stack(rows, 0).reshape(H, W, 3).permute(2, 0, 1)
    return img
