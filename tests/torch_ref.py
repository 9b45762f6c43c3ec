"""Independent differentiable restatement of SURVEY.md Appendix A in plain torch (fp64), used ONLY to
cross-check the C oracle's forward and its explicit backward with torch autograd on small cases.

It is deliberately written differently from the oracle (vectorised over pixels, one global
depth-ordered loop over Gaussians, autograd instead of hand-derived gradients).  The three places
where the reference's gradient deviates from naive autograd (A.9: alpha clamp not differentiated;
A.10: clamped tx/tz contributes no gradient; `1/(den^2+1e-7)`) are mirrored with detach() tricks so
the comparison can be tight.
"""
from __future__ import annotations

import math

import torch

C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
      1.445305721320277, -0.5900435899266435]


def eval_sh_basis(D, d):
    x, y, z = d[:, 0], d[:, 1], d[:, 2]
    b = [torch.full_like(x, C0)]
    if D > 0:
        b += [-C1 * y, C1 * z, -C1 * x]
    if D > 1:
        xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
        b += [C2[0] * xy, C2[1] * yz, C2[2] * (2 * zz - xx - yy), C2[3] * xz, C2[4] * (xx - yy)]
    if D > 2:
        b += [C3[0] * y * (3 * xx - yy), C3[1] * xy * z, C3[2] * y * (4 * zz - xx - yy),
              C3[3] * z * (2 * zz - 3 * xx - 3 * yy), C3[4] * x * (4 * zz - xx - yy), C3[5] * z * (xx - yy),
              C3[6] * x * (xx - 3 * yy)]
    return torch.stack(b, dim=1)  # [P, K]


def quat_to_rot(q):
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                     2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                     2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1)
    return R.view(-1, 3, 3)


def render_autograd(*, image_height, image_width, tanfovx, tanfovy, bg, scale_modifier, viewmatrix, projmatrix,
                    sh_degree, campos, means3D, opacities, shs=None, colors_precomp=None, scales=None,
                    rotations=None, cov3D_precomp=None, tile_rows=None):
    """Returns (color[3,H,W], radii[P], means2D_proxy[P,2]).  `means2D_proxy` is a zero tensor added to the
    NDC position so that its autograd gradient is the reference's `means2D.grad[:, :2]`."""
    dt = torch.float64
    H, W = int(image_height), int(image_width)
    Gx, Gy = (W + 15) // 16, (H + 15) // 16
    ty0, ty1 = (0, Gy) if tile_rows is None else tile_rows
    V, PV = viewmatrix.to(dt), projmatrix.to(dt)
    P = means3D.shape[0]
    ones = torch.ones(P, 1, dtype=dt)
    ph = torch.cat([means3D, ones], 1)
    pv = ph @ V
    hom = ph @ PV
    visible = pv[:, 2] > 0.2
    pw = 1.0 / (hom[:, 3] + 1e-7)
    means2D_proxy = torch.zeros(P, 2, dtype=dt, requires_grad=True)
    ndc = hom[:, :2] * pw[:, None] + means2D_proxy
    fx, fy = W / (2 * tanfovx), H / (2 * tanfovy)

    if cov3D_precomp is None:
        R = quat_to_rot(rotations)
        S = torch.diag_embed(scales * scale_modifier)
        L = R @ S
        Sigma = L @ L.transpose(1, 2)
    else:
        c = cov3D_precomp
        Sigma = torch.stack([c[:, 0], c[:, 1], c[:, 2], c[:, 1], c[:, 3], c[:, 4], c[:, 2], c[:, 4], c[:, 5]], 1).view(-1, 3, 3)

    limx, limy = 1.3 * tanfovx, 1.3 * tanfovy
    tz = pv[:, 2]
    txtz, tytz = pv[:, 0] / tz, pv[:, 1] / tz
    cx, cy = (txtz < -limx) | (txtz > limx), (tytz < -limy) | (tytz > limy)
    tx = torch.where(cx, (txtz.clamp(-limx, limx) * tz).detach(), pv[:, 0])
    ty = torch.where(cy, (tytz.clamp(-limy, limy) * tz).detach(), pv[:, 1])
    zero = torch.zeros_like(tz)
    J = torch.stack([fx / tz, zero, -fx * tx / (tz * tz), zero, fy / tz, -fy * ty / (tz * tz)], 1).view(-1, 2, 3)
    Wm = V[:3, :3].t()
    T = J @ Wm
    cov2 = T @ Sigma @ T.transpose(1, 2)
    a, b, c_ = cov2[:, 0, 0] + 0.3, cov2[:, 0, 1], cov2[:, 1, 1] + 0.3
    det = a * c_ - b * b
    # mirror of k = 1/(den^2 + 1e-7): d(conic)/d(cov) is scaled by den^2/(den^2+1e-7)
    conA, conB, conC = c_ / det, -b / det, a / det
    kfix = (det * det / (det * det + 1e-7)).detach()

    def rescale_grad(v):
        return v.detach() + kfix * (v - v.detach())
    conA, conB, conC = rescale_grad(conA), rescale_grad(conB), rescale_grad(conC)
    mid = 0.5 * (a + c_)
    lam = mid + torch.sqrt(torch.clamp(mid * mid - det, min=0.1))
    radius = torch.ceil(3 * torch.sqrt(lam)).detach()
    px = ((ndc[:, 0] + 1) * W - 1) * 0.5
    py = ((ndc[:, 1] + 1) * H - 1) * 0.5

    def trunc_clamp(v, hi):
        return torch.clamp(torch.trunc(v), 0, hi).to(torch.int64)
    rx0 = trunc_clamp((px.detach() - radius) / 16, Gx); rx1 = trunc_clamp((px.detach() + radius + 15) / 16, Gx)
    ry0 = trunc_clamp((py.detach() - radius) / 16, Gy); ry1 = trunc_clamp((py.detach() + radius + 15) / 16, Gy)
    visible = visible & (det != 0) & ((rx1 - rx0) * (ry1 - ry0) > 0)
    radii = torch.where(visible, radius, torch.zeros_like(radius)).to(torch.int32)

    if colors_precomp is None:
        d = means3D - campos.to(dt)[None]
        d = d / d.norm(dim=1, keepdim=True)
        K = (sh_degree + 1) ** 2
        basis = eval_sh_basis(sh_degree, d)
        rgb = torch.einsum("pk,pkc->pc", basis, shs[:, :K, :]) + 0.5
        rgb = torch.clamp_min(rgb, 0.0)
    else:
        rgb = colors_precomp

    ys, xs = torch.meshgrid(torch.arange(H, dtype=dt), torch.arange(W, dtype=dt), indexing="ij")
    tyy, txx = (ys // 16).to(torch.int64), (xs // 16).to(torch.int64)
    in_slab = (tyy >= ty0) & (tyy < ty1)
    Tacc = torch.ones(H, W, dtype=dt)
    Cacc = torch.zeros(3, H, W, dtype=dt)
    done = ~in_slab
    depth32 = pv[:, 2].detach().to(torch.float32)
    order = sorted([i for i in range(P) if bool(visible[i])], key=lambda i: (float(depth32[i]), i))
    for i in order:
        m = (txx >= rx0[i]) & (txx < rx1[i]) & (tyy >= max(int(ry0[i]), ty0)) & (tyy < min(int(ry1[i]), ty1)) & ~done
        if not bool(m.any()):
            continue
        dx, dy = px[i] - xs, py[i] - ys
        power = -0.5 * (conA[i] * dx * dx + conC[i] * dy * dy) - conB[i] * dx * dy
        araw = opacities.view(-1)[i] * torch.exp(power)
        alpha = araw + (torch.clamp(araw, max=0.99) - araw).detach()   # clamp not differentiated (A.9 i)
        m = m & (power <= 0) & (alpha >= 1.0 / 255.0)
        test_T = Tacc * (1 - alpha)
        stop = m & (test_T < 1e-4)
        done = done | stop
        m = m & ~stop
        w = torch.where(m, alpha * Tacc, torch.zeros_like(Tacc))
        Cacc = Cacc + rgb[i][:, None, None] * w[None]
        Tacc = torch.where(m, test_T, Tacc)
    color = Cacc + Tacc[None] * bg.to(dt)[:, None, None]
    color = torch.where(in_slab[None], color, torch.zeros_like(color))
    return color, radii, means2D_proxy, Tacc
