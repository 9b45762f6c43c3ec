"""render(): host-side mirror of the reference's only caller of the rasterizer
(gaussian_renderer/__init__.py:18-100).  Same arguments, same returned dict, same optional-input
selection (`pipe.compute_cov3D_python`, `pipe.convert_SHs_python`, `override_color`); `pc` is anything
with the reference GaussianModel's getters (get_xyz, get_opacity, get_scaling, get_rotation,
get_features, get_covariance, active_sh_degree, max_sh_degree).

Extension: `pipe.fused_activations` (not a reference flag).  True: render from the model's raw parameters
`_xyz, _features_dc, _features_rest, _opacity, _scaling, _rotation` with the activations fused into the HIP kernels
(GaussianRasterizer.forward_raw) — same image, same gradients on the parameters, without the ~30 torch kernels of the getters
and their backward.  Unset (None, the default): that path for this package's scene.GaussianModel (whose SH coefficients are
one interleaved leaf `_features`: raw mode 2), the getters for any other store.  False: always the getters.
"""
import math

import torch

from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer

_C0 = 0.28209479177387814
_C1 = 0.4886025119029199
_C2 = (1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396)
_C3 = (-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154, -0.4570457994644658,
       1.445305721320277, -0.5900435899266435)


def eval_sh(deg: int, sh: torch.Tensor, dirs: torch.Tensor) -> torch.Tensor:
    """SH colour in torch for the `convert_SHs_python` branch; sh [..., C, K], dirs [..., 3]
    (same basis as utils/sh_utils.py:57-112; pinned by tests/golden/sh_eval.npz)."""
    assert 0 <= deg <= 3
    res = _C0 * sh[..., 0]
    if deg > 0:
        x, y, z = dirs[..., 0:1], dirs[..., 1:2], dirs[..., 2:3]
        res = res - _C1 * y * sh[..., 1] + _C1 * z * sh[..., 2] - _C1 * x * sh[..., 3]
        if deg > 1:
            xx, yy, zz, xy, yz, xz = x * x, y * y, z * z, x * y, y * z, x * z
            res = (res + _C2[0] * xy * sh[..., 4] + _C2[1] * yz * sh[..., 5] + _C2[2] * (2.0 * zz - xx - yy) * sh[..., 6]
                   + _C2[3] * xz * sh[..., 7] + _C2[4] * (xx - yy) * sh[..., 8])
            if deg > 2:
                res = (res + _C3[0] * y * (3 * xx - yy) * sh[..., 9] + _C3[1] * xy * z * sh[..., 10]
                       + _C3[2] * y * (4 * zz - xx - yy) * sh[..., 11] + _C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * sh[..., 12]
                       + _C3[4] * x * (4 * zz - xx - yy) * sh[..., 13] + _C3[5] * z * (xx - yy) * sh[..., 14]
                       + _C3[6] * x * (xx - 3 * yy) * sh[..., 15])
    return res


_ZERO_POINTS = {}


def _zero_points(like: torch.Tensor) -> torch.Tensor:
    """One block of zeros shaped like `like` per device (re-made when the model's size or dtype changes)."""
    key = (like.device.type, like.device.index)
    z = _ZERO_POINTS.get(key)
    if z is None or z.shape != like.shape or z.dtype != like.dtype:
        z = _ZERO_POINTS[key] = torch.zeros_like(like, requires_grad=False)
    return z


def render(viewpoint_camera, pc, pipe, bg_color: torch.Tensor, scaling_modifier=1.0, override_color=None):
    """Render the scene.  Background tensor (bg_color) must be on the GPU."""
    xyz = pc.get_xyz
    # (the reference builds this as zeros_like(...) + 0 and retain_grad()s the non-leaf result; a leaf with requires_grad keeps
    # its .grad by itself.  Its VALUES are never read, by the rasterizer or by the training loop, only its .grad: every frame
    # gets a fresh leaf over one shared block of zeros per device instead of a 12 P-byte fill of its own)
    screenspace_points = _zero_points(xyz).detach().requires_grad_(True)

    raster_settings = GaussianRasterizationSettings(
        image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width),
        tanfovx=math.tan(viewpoint_camera.FoVx * 0.5), tanfovy=math.tan(viewpoint_camera.FoVy * 0.5),
        bg=bg_color, scale_modifier=scaling_modifier, viewmatrix=viewpoint_camera.world_view_transform,
        projmatrix=viewpoint_camera.full_proj_transform, sh_degree=pc.active_sh_degree,
        campos=viewpoint_camera.camera_center, prefiltered=False, debug=pipe.debug)
    rasterizer = GaussianRasterizer(raster_settings=raster_settings)

    # the fused path hands the RAW parameters to the kernels and so bypasses the getters — and with them the reference's
    # freeze flags (scene/gaussian_model.py:104-125 detach() the getter's result): a model with any of them set takes the
    # plain path, where a frozen parameter receives no gradient
    frozen = any(getattr(pc, f, False) for f in ("freeze_means", "freeze_scales", "freeze_rotations", "freeze_opacities"))
    packed = bool(getattr(pc, "packed_features", False))          # scene.GaussianModel: get_features is ONE interleaved leaf [P,M,3]
    fused = getattr(pipe, "fused_activations", None)
    if fused is None:            # not set: this package's own model renders from its raw leaves, any other store through its getters
        fused = packed
    if (fused and override_color is None and not pipe.compute_cov3D_python and not pipe.convert_SHs_python and not frozen
            and (packed or hasattr(pc, "_features_rest"))):
        if packed:               # exp / normalize / sigmoid inside the kernels, the SH table as it is: no getter runs at all
            rendered_image, radii = rasterizer.forward_raw(pc._xyz, screenspace_points, pc._features, None,
                                                           pc._opacity, pc._scaling, pc._rotation)
        else:
            rendered_image, radii = rasterizer.forward_raw(pc._xyz, screenspace_points, pc._features_dc, pc._features_rest,
                                                           pc._opacity, pc._scaling, pc._rotation)
        return {"render": rendered_image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0,
                "radii": radii}

    scales = rotations = cov3D_precomp = None
    if pipe.compute_cov3D_python:
        cov3D_precomp = pc.get_covariance(scaling_modifier)
    else:
        scales, rotations = pc.get_scaling, pc.get_rotation

    shs = colors_precomp = None
    if override_color is None:
        if pipe.convert_SHs_python:
            feats = pc.get_features
            shs_view = feats.transpose(1, 2).view(-1, 3, (int(pc.max_sh_degree) + 1) ** 2)
            dir_pp = xyz - viewpoint_camera.camera_center.repeat(feats.shape[0], 1)
            dir_pp = dir_pp / dir_pp.norm(dim=1, keepdim=True)
            colors_precomp = torch.clamp_min(eval_sh(int(pc.active_sh_degree), shs_view, dir_pp) + 0.5, 0.0)
        else:
            shs = pc.get_features
    else:
        colors_precomp = override_color

    rendered_image, radii = rasterizer(means3D=xyz, means2D=screenspace_points, shs=shs, colors_precomp=colors_precomp,
                                       opacities=pc.get_opacity, scales=scales, rotations=rotations,
                                       cov3D_precomp=cov3D_precomp)
    return {"render": rendered_image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0,
            "radii": radii}
