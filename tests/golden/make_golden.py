#!/usr/bin/env python3
"""Generate the committed golden vectors from the reference's OWN Python (run in the authoring
container only; /root/reference does not exist on the GPU box and is never read by tests/bench).

What it pins: the in-tree twins of the rasterizer's sub-steps —
  utils/sh_utils.py:57-118        eval_sh, RGB2SH, SH2RGB, C0        -> sh_eval.npz
  utils/graphics_utils.py:22-77   view/projection matrices, point xf -> camera.npz
  utils/loss_utils.py:17-63, utils/image_utils.py:14-19  losses      -> loss.npz
  utils/general_utils.py:29-62    get_expon_lr_func                   -> lr.npz
  utils/general_utils.py:64-110   build_rotation, build_scaling_rotation, strip_symmetric (A.3: quaternion -> R,
                                  L = R S, Sigma = L L^T packed [xx,xy,xz,yy,yz,zz]) and the covariance builder of
                                  scene/gaussian_model.py:25-29                                              -> cov3d.npz
                                  (those functions hard-code device="cuda"; HERE ONLY the torch factory they call is
                                  wrapped to drop that keyword so that they run on the CPU of the authoring container)
  arguments/__init__.py:47-95     default hyper-parameters            -> params.json
Only inputs and expected outputs are stored (data, not source).

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py
"""
import json
import math
import os
import sys
from argparse import ArgumentParser

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)

from utils.sh_utils import eval_sh, RGB2SH, SH2RGB, C0  # noqa: E402
from utils.graphics_utils import (getWorld2View2, getProjectionMatrix, geom_transform_points,  # noqa: E402
                                  fov2focal, focal2fov)
from utils.loss_utils import l1_loss, ssim  # noqa: E402
from utils.image_utils import psnr  # noqa: E402
from utils.general_utils import get_expon_lr_func  # noqa: E402


def sh_vectors():
    g = torch.Generator().manual_seed(1234)
    P = 96
    campos = torch.tensor([0.3, -0.2, 0.1], dtype=torch.float64)
    out = dict(campos=campos.numpy(), C0=np.float64(C0))
    pos = torch.randn(P, 3, generator=g, dtype=torch.float64) * 0.6 + torch.tensor([0.0, 0.0, 3.0], dtype=torch.float64)
    sh = torch.randn(P, 16, 3, generator=g, dtype=torch.float64)
    sh[:, 0] -= 0.6          # push some channels below zero so the clamp mask is exercised
    up = torch.rand(P, 3, generator=g, dtype=torch.float64) * 2 - 1
    out.update(pos=pos.numpy(), sh=sh.numpy(), upstream=up.numpy())
    for D in range(4):
        p = pos.clone().requires_grad_(True)
        s = sh.clone().requires_grad_(True)
        # exactly the reference's Python colour path, gaussian_renderer/__init__.py:73-78
        shs_view = s.transpose(1, 2).view(-1, 3, 16)
        dir_pp = p - campos.repeat(P, 1)
        dir_n = dir_pp / dir_pp.norm(dim=1, keepdim=True)
        raw = eval_sh(D, shs_view, dir_n)
        col = torch.clamp_min(raw + 0.5, 0.0)
        (col * up).sum().backward()
        out[f"raw_D{D}"] = raw.detach().numpy()
        out[f"color_D{D}"] = col.detach().numpy()
        out[f"grad_sh_D{D}"] = s.grad.numpy()
        out[f"grad_pos_D{D}"] = np.zeros((P, 3)) if p.grad is None else p.grad.numpy()
    rgb = torch.rand(8, 3, generator=g, dtype=torch.float64)
    out["rgb"] = rgb.numpy(); out["rgb2sh"] = RGB2SH(rgb).numpy(); out["sh2rgb"] = SH2RGB(rgb).numpy()
    np.savez_compressed(os.path.join(OUT, "sh_eval.npz"), **out)


def camera_vectors():
    g = torch.Generator().manual_seed(99)
    cams = []
    specs = [(np.eye(3), np.zeros(3), 640, 480, 0.5), (None, None, 1920, 1080, 0.5), (None, None, 800, 800, 0.7),
             (None, None, 48, 80, 0.35)]
    rs = np.random.RandomState(7)
    out = {}
    for i, (R, t, W, H, tanfovy) in enumerate(specs):
        if R is None:
            q = rs.randn(4); q /= np.linalg.norm(q)
            r, x, y, z = q
            R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)],
                          [2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)],
                          [2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)]])
            t = rs.randn(3) * 0.5 + np.array([0, 0, 2.0])
        tanfovx = tanfovy * W / H
        fovx, fovy = 2 * math.atan(tanfovx), 2 * math.atan(tanfovy)
        # exactly scene/cameras.py:54-57 (minus .cuda())
        wvt = torch.tensor(getWorld2View2(R, t, np.array([0.0, 0.0, 0.0]), 1.0)).transpose(0, 1)
        proj = getProjectionMatrix(znear=0.01, zfar=100.0, fovX=fovx, fovY=fovy).transpose(0, 1)
        full = (wvt.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0)
        center = wvt.inverse()[3, :3]
        pts = torch.randn(32, 3, generator=g) * 0.8 + torch.tensor([0.0, 0.0, 0.5])
        out[f"R{i}"] = R; out[f"t{i}"] = t; out[f"WH{i}"] = np.array([W, H]); out[f"tanfovy{i}"] = np.float64(tanfovy)
        out[f"wvt{i}"] = wvt.numpy(); out[f"proj{i}"] = proj.numpy(); out[f"full{i}"] = full.numpy()
        out[f"center{i}"] = center.numpy(); out[f"pts{i}"] = pts.numpy()
        out[f"pts_view{i}"] = geom_transform_points(pts, wvt).numpy()
        out[f"pts_ndc{i}"] = geom_transform_points(pts, full).numpy()
        out[f"focal{i}"] = np.array([fov2focal(fovx, W), fov2focal(fovy, H)])
        out[f"fov_rt{i}"] = np.array([focal2fov(fov2focal(fovx, W), W), fovx])
        cams.append(i)
    out["n"] = np.int64(len(cams))
    np.savez_compressed(os.path.join(OUT, "camera.npz"), **out)


def loss_vectors():
    g = torch.Generator().manual_seed(4321)
    out = {}
    for i, (H, W) in enumerate([(40, 56), (64, 64)]):
        a = torch.rand(3, H, W, generator=g).requires_grad_(True)
        b = torch.rand(3, H, W, generator=g)
        l1 = l1_loss(a, b)
        s = ssim(a, b)
        loss = (1.0 - 0.2) * l1 + 0.2 * (1.0 - s)          # train.py:104-105, lambda_dssim = 0.2
        loss.backward()
        out[f"a{i}"] = a.detach().numpy(); out[f"b{i}"] = b.numpy()
        out[f"l1_{i}"] = l1.item(); out[f"ssim_{i}"] = s.item(); out[f"loss_{i}"] = loss.item()
        out[f"grad_a{i}"] = a.grad.numpy(); out[f"psnr_{i}"] = psnr(a.detach(), b).numpy()
        # each function's own gradient (the drop-in l1_loss / ssim autograd ops are checked one by one)
        a2 = a.detach().clone().requires_grad_(True)
        l1_loss(a2, b).backward()
        out[f"grad_l1_a{i}"] = a2.grad.numpy()
        a3 = a.detach().clone().requires_grad_(True)
        ssim(a3, b).backward()
        out[f"grad_ssim_a{i}"] = a3.grad.numpy()
    np.savez_compressed(os.path.join(OUT, "loss.npz"), **out)


def cov3d_vectors():
    """A.3 from the reference's own functions.  utils/general_utils.py:65,83,102 create their outputs with
    device="cuda"; in this container (no GPU) that keyword is dropped by wrapping torch.zeros for the duration of the
    calls - nothing else about the functions is touched."""
    import utils.general_utils as gu
    real_zeros = torch.zeros

    def zeros_anywhere(*a, **k):
        if str(k.get("device", "")).startswith("cuda"):
            k.pop("device")
        return real_zeros(*a, **k)
    g = torch.Generator().manual_seed(2024)
    P = 64
    q = torch.randn(P, 4, generator=g, dtype=torch.float32)            # un-normalised: build_rotation normalises itself
    s = torch.exp(torch.rand(P, 3, generator=g) * 4.0 - 5.0)           # scales over two decades
    mods = (1.0, 1.7)
    out = dict(rotation=q.numpy(), scaling=s.numpy(), modifiers=np.array(mods))
    torch.zeros = zeros_anywhere
    try:
        out["R"] = gu.build_rotation(q).numpy()
        for j, m in enumerate(mods):
            L = gu.build_scaling_rotation(m * s, q)                   # scene/gaussian_model.py:25-29
            cov = L @ L.transpose(1, 2)
            out[f"L_{j}"] = L.numpy()
            out[f"cov6_{j}"] = gu.strip_symmetric(cov).numpy()
    finally:
        torch.zeros = real_zeros
    np.savez_compressed(os.path.join(OUT, "cov3d.npz"), **out)


def lr_vectors():
    f = get_expon_lr_func(lr_init=0.00016, lr_final=0.0000016, lr_delay_mult=0.01, max_steps=30000)
    steps = np.array([0, 1, 100, 15000, 30000])
    np.savez_compressed(os.path.join(OUT, "lr.npz"), steps=steps, lr=np.array([f(int(s)) for s in steps]))


def param_defaults():
    from arguments import ModelParams, PipelineParams, OptimizationParams
    parser = ArgumentParser()
    groups = dict(model=ModelParams(parser), pipeline=PipelineParams(parser), optimization=OptimizationParams(parser))
    dump = {k: {n.lstrip("_"): v for n, v in vars(g).items()} for k, g in groups.items()}
    with open(os.path.join(OUT, "params.json"), "w") as fh:
        json.dump(dump, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    only = set(sys.argv[1:])
    for name, fn in (("sh", sh_vectors), ("camera", camera_vectors), ("loss", loss_vectors), ("lr", lr_vectors),
                     ("params", param_defaults), ("cov3d", cov3d_vectors)):
        if not only or name in only:
            fn()
    print("golden vectors written to", OUT)
