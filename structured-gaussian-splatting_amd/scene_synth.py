"""Seeded synthetic scenes and closed-form cameras.

Camera maths mirrors the reference's conventions:
  * ``world_view_transform = getWorld2View2(R, T).T``           (scene/cameras.py:54, utils/graphics_utils.py:38-49)
  * ``projection = getProjectionMatrix(znear, zfar, fovX, fovY).T`` (scene/cameras.py:55, utils/graphics_utils.py:51-71)
  * ``full_proj_transform = world_view_transform @ projection``  (scene/cameras.py:56)
  * ``camera_center = inverse(world_view_transform)[3, :3]``     (scene/cameras.py:57)
with znear = 0.01, zfar = 100 (scene/cameras.py:48-49).  Everything is generated on the host with a
CPU generator (fp32) and then moved to the requested device, so a seed names the same scene everywhere.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional

import numpy as np
import torch

ZNEAR, ZFAR = 0.01, 100.0
SH_C0 = 0.28209479177387814  # utils/sh_utils.py:26


def world_to_view(R: np.ndarray, t: np.ndarray) -> np.ndarray:
    """4x4 world-to-camera matrix for column vectors (utils/graphics_utils.py:38-49 with
    translate = 0, scale = 1): rotation block R^T, translation t."""
    Rt = np.zeros((4, 4), np.float64)
    Rt[:3, :3] = np.asarray(R, np.float64).T
    Rt[:3, 3] = np.asarray(t, np.float64)
    Rt[3, 3] = 1.0
    return Rt.astype(np.float32)


def projection_matrix(znear: float, zfar: float, fovx: float, fovy: float) -> torch.Tensor:
    """OpenGL-style perspective matrix with z_sign = +1 (utils/graphics_utils.py:51-71)."""
    ty, tx = math.tan(fovy / 2), math.tan(fovx / 2)
    top, right = ty * znear, tx * znear
    Pm = torch.zeros(4, 4)
    Pm[0, 0] = 2.0 * znear / (2 * right)
    Pm[1, 1] = 2.0 * znear / (2 * top)
    Pm[3, 2] = 1.0
    Pm[2, 2] = zfar / (zfar - znear)
    Pm[2, 3] = -(zfar * znear) / (zfar - znear)
    return Pm


@dataclass
class Camera:
    """The fields render() reads from a reference Camera/MiniCam (scene/cameras.py:17-70)."""
    image_width: int
    image_height: int
    FoVx: float
    FoVy: float
    world_view_transform: torch.Tensor   # [4,4], transposed (row-vector convention)
    full_proj_transform: torch.Tensor    # [4,4]
    camera_center: torch.Tensor          # [3]
    znear: float = ZNEAR
    zfar: float = ZFAR

    def to(self, device) -> "Camera":
        return Camera(self.image_width, self.image_height, self.FoVx, self.FoVy,
                      self.world_view_transform.to(device), self.full_proj_transform.to(device),
                      self.camera_center.to(device), self.znear, self.zfar)


def make_camera(width: int, height: int, R: Optional[np.ndarray] = None, t: Optional[np.ndarray] = None,
                tanfovy: float = 0.5, tanfovx: Optional[float] = None) -> Camera:
    """Camera with square pixels by default: tanfovx = tanfovy * W / H (Appendix B)."""
    R = np.eye(3) if R is None else R
    t = np.zeros(3) if t is None else t
    tanfovx = tanfovy * width / height if tanfovx is None else tanfovx
    fovx, fovy = 2 * math.atan(tanfovx), 2 * math.atan(tanfovy)
    wvt = torch.tensor(world_to_view(R, t)).transpose(0, 1).contiguous()
    proj = projection_matrix(ZNEAR, ZFAR, fovx, fovy).transpose(0, 1)
    full = (wvt.unsqueeze(0).bmm(proj.unsqueeze(0))).squeeze(0).contiguous()
    center = wvt.inverse()[3, :3].contiguous()
    return Camera(width, height, fovx, fovy, wvt, full, center)


def arc_cameras(width: int, height: int, n: int, radius: float = 0.35, tanfovy: float = 0.5):
    """n cameras on a small arc around the origin, all looking down +z (training-loop bench)."""
    cams = []
    for i in range(n):
        a = (i / max(n - 1, 1) - 0.5) * 0.5
        Rm = np.array([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]])
        c2w_pos = np.array([radius * math.sin(a) * 2, 0.0, 0.0])
        t = -Rm.T @ c2w_pos
        cams.append(make_camera(width, height, Rm, t, tanfovy))
    return cams


@dataclass
class Scene:
    means3D: torch.Tensor      # [P,3]
    log_scales: torch.Tensor   # [P,3]  raw (exp -> scales)
    raw_rotations: torch.Tensor  # [P,4] raw (normalize -> rotations)
    opacity_logits: torch.Tensor  # [P,1] raw (sigmoid -> opacities)
    shs: torch.Tensor          # [P,M,3]
    sh_degree: int

    @property
    def P(self):
        return self.means3D.shape[0]

    def activated(self):
        """Activations exactly as the reference's getters (scene/gaussian_model.py:101-125)."""
        return dict(means3D=self.means3D, scales=torch.exp(self.log_scales),
                    rotations=torch.nn.functional.normalize(self.raw_rotations),
                    opacities=torch.sigmoid(self.opacity_logits), shs=self.shs)

    def to(self, device) -> "Scene":
        return Scene(self.means3D.to(device), self.log_scales.to(device), self.raw_rotations.to(device),
                     self.opacity_logits.to(device), self.shs.to(device), self.sh_degree)


def make_scene(P: int, width: int, height: int, sh_degree: int, seed: int, tanfovy: float = 0.5,
               scale_lo: float = 0.003, scale_hi: float = 0.03, zmax: float = 6.0, zmin: float = 0.0) -> Scene:
    """Appendix B Gaussians: z ~ U(zmin = 0, 6); x,y inside 1.1x the frustum; log-scales ~ U(ln .003, ln .03);
    raw quaternion ~ N(0,1); opacity logit ~ N(0, 1.5^2); SH DC ~ N(0,1)*0.25/C0, rest ~ N(0, 0.1^2)."""
    g = torch.Generator(device="cpu").manual_seed(int(seed))
    tanfovx = tanfovy * width / height
    z = torch.rand(P, generator=g) * (zmax - zmin) + zmin
    u = torch.rand(P, generator=g) * 2 - 1
    v = torch.rand(P, generator=g) * 2 - 1
    means = torch.stack([u * 1.1 * tanfovx * z, v * 1.1 * tanfovy * z, z], dim=1)
    log_scales = torch.rand(P, 3, generator=g) * (math.log(scale_hi) - math.log(scale_lo)) + math.log(scale_lo)
    rots = torch.randn(P, 4, generator=g)
    opac = torch.randn(P, 1, generator=g) * 1.5
    M = (sh_degree + 1) ** 2
    shs = torch.randn(P, M, 3, generator=g) * 0.1
    shs[:, 0, :] = torch.randn(P, 3, generator=g) * 0.25 / SH_C0
    return Scene(means.contiguous(), log_scales, rots, opac, shs.contiguous(), sh_degree)


def make_grad_image(width: int, height: int, seed: int) -> torch.Tensor:
    """Seeded dL/dcolor ~ U(-1, 1) [3,H,W] (Appendix B, kernel-only parity/bench)."""
    g = torch.Generator(device="cpu").manual_seed(int(seed) + 7919)
    return torch.rand(3, height, width, generator=g) * 2 - 1


# The BASELINE.json configurations (SURVEY Appendix B seeds).
CONFIGS = {
    "cfg1": dict(P=10_000, W=256, H=256, D=0, seed=1),
    "cfg2": dict(P=100_000, W=800, H=800, D=3, seed=2),
    "cfg3": dict(P=1_000_000, W=1920, H=1080, D=3, seed=3),
    "cfg5": dict(P=5_000_000, W=3840, H=2160, D=3, seed=5),
    # Not a BASELINE config: the cfg3 generator with the depth range moved off the camera (z ~ U(2, 6)) and the scales
    # halved, so that splats are the "1-10 px" SURVEY Appendix B describes (sigma 0.3-8 px at 1080p) and R/P lands in its
    # expected 3-6 (measured 3.55).  98.5 % of the visible Gaussians receive a gradient and half the pixels never reach
    # the transmittance cut-off: the ordinary, NON-saturating case next to cfg3's degenerate one (bench.py `secondary`).
    "cfg3n": dict(P=1_000_000, W=1920, H=1080, D=3, seed=3, zmin=2.0, scale_mul=0.5),
    "cfg3b": dict(P=1_000_000, W=1920, H=1080, D=3, seed=33),      # the cfg3 generator, another seed
    # BASELINE configs[4] ("5M Gaussians, 4K, SH 3 - HBM-bound stress") with the cfg3n recipe: cfg5's own generator saturates (4 656 of
    # its 5 000 000 Gaussians are ever binned), this one does not, so the streaming kernels (preprocess, colours, the dense geometry
    # backward, the radix sorts) run at the size the config names (bench.py `secondary_4k`).
    "cfg5n": dict(P=5_000_000, W=3840, H=2160, D=3, seed=5, zmin=2.0, scale_mul=0.5),
}


def make_config(name: str):
    c = CONFIGS[name]
    m = c.get("scale_mul", 1.0)
    return (make_scene(c["P"], c["W"], c["H"], c["D"], c["seed"], scale_lo=0.003 * m, scale_hi=0.03 * m, zmin=c.get("zmin", 0.0)),
            make_camera(c["W"], c["H"]))
