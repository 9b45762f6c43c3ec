"""Shared helpers for the test-suite (scene → kwargs plumbing, tolerant comparisons)."""
import math

import numpy as np
import torch

import scene_synth as S


def raster_kwargs(scene: S.Scene, cam: S.Camera, bg=(0.0, 0.0, 0.0), scale_modifier=1.0, as_numpy=True,
                  colors_precomp=None, cov3D_precomp=None):
    """kwargs named like GaussianRasterizationSettings + GaussianRasterizer.forward."""
    a = scene.activated()
    kw = dict(image_height=cam.image_height, image_width=cam.image_width,
              tanfovx=math.tan(cam.FoVx * 0.5), tanfovy=math.tan(cam.FoVy * 0.5),
              bg=torch.tensor(bg, dtype=torch.float32), scale_modifier=scale_modifier,
              viewmatrix=cam.world_view_transform, projmatrix=cam.full_proj_transform,
              sh_degree=scene.sh_degree, campos=cam.camera_center,
              means3D=a["means3D"], opacities=a["opacities"])
    if colors_precomp is None:
        kw["shs"] = a["shs"]
    else:
        kw["colors_precomp"] = colors_precomp
    if cov3D_precomp is None:
        kw["scales"], kw["rotations"] = a["scales"], a["rotations"]
    else:
        kw["cov3D_precomp"] = cov3D_precomp
    if as_numpy:
        kw = {k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else v) for k, v in kw.items()}
    return kw


def cov3d_from(scales: torch.Tensor, rotations: torch.Tensor, mod: float = 1.0) -> torch.Tensor:
    """Packed [xx,xy,xz,yy,yz,zz] covariance = R S^2 R^T (scene/gaussian_model.py:25-29 restated)."""
    r, x, y, z = rotations.unbind(1)
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                     2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                     2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).view(-1, 3, 3)
    L = R @ torch.diag_embed(scales * mod)
    Sg = L @ L.transpose(1, 2)
    return torch.stack([Sg[:, 0, 0], Sg[:, 0, 1], Sg[:, 0, 2], Sg[:, 1, 1], Sg[:, 1, 2], Sg[:, 2, 2]], 1).contiguous()


def assert_close_masked(got, want, mask_strict, atol, rtol, what, loose_atol=None):
    """|got-want| <= atol + rtol*|want| on mask_strict elements; optional loose bound elsewhere."""
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    err = np.abs(got - want)
    bound = atol + rtol * np.abs(want)
    bad = (err > bound) & mask_strict
    assert not bad.any(), (f"{what}: {bad.sum()} / {mask_strict.sum()} strict elements out of tolerance; "
                           f"max err {err[mask_strict].max():.3e}")
    if loose_atol is not None and (~mask_strict).any():
        assert err[~mask_strict].max() <= loose_atol, f"{what}: non-strict max err {err[~mask_strict].max():.3e}"


LOG2E = 1.4426950408889634


def unscale_records(rec):
    """Device splat records store the conic and opacity pre-scaled (csrc/gsr_math.h Splat): return a copy with
    columns 2..5 = conic A, B, C and opacity as the oracle reports them."""
    out = np.array(rec, dtype=np.float32, copy=True)
    out[:, 2] = rec[:, 2] * np.float32(-2.0 / LOG2E)
    out[:, 3] = rec[:, 3] * np.float32(-1.0 / LOG2E)
    out[:, 4] = rec[:, 4] * np.float32(-2.0 / LOG2E)
    out[:, 5] = np.exp2(rec[:, 5].astype(np.float64)).astype(np.float32)
    return out
