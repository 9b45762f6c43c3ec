"""The oracle against (i) an independent torch fp64 restatement + autograd and (ii) analytic
known-answer cases.  The reference ships no test or golden vector for the rasterizer proper
("parity unpinned"), so these self-consistency checks are what holds the oracle's
forward and explicit backward (Appendix A.8-A.10) in place.
"""
import math

import numpy as np
import pytest
import torch

import oracle
import scene_synth as S
from torch_ref import render_autograd
from util import cov3d_from, raster_kwargs


def _small_scene(P, W, H, D, seed, zmax=4.0):
    return S.make_scene(P, W, H, D, seed, scale_lo=0.02, scale_hi=0.25, zmax=zmax), S.make_camera(W, H)


def _to_t64(kw):
    out = {}
    for k, v in kw.items():
        out[k] = torch.tensor(v, dtype=torch.float64) if isinstance(v, np.ndarray) else v
    return out


CASES = [
    dict(P=40, W=48, H=32, D=3, seed=11, mode="sh+scale"),
    dict(P=40, W=40, H=56, D=1, seed=12, mode="sh+scale"),
    dict(P=32, W=48, H=48, D=2, seed=13, mode="color+cov"),
    dict(P=24, W=33, H=47, D=0, seed=14, mode="sh+scale", bg=(1.0, 0.5, 0.25)),
]


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"P{c['P']}_{c['W']}x{c['H']}_D{c['D']}_{c['mode']}")
def test_oracle_f64_matches_torch_autograd(case):
    scene, cam = _small_scene(case["P"], case["W"], case["H"], case["D"], case["seed"])
    a = scene.activated()
    extra = {}
    if case["mode"] == "color+cov":
        g = torch.Generator().manual_seed(5)
        extra = dict(colors_precomp=torch.rand(scene.P, 3, generator=g),
                     cov3D_precomp=cov3d_from(a["scales"], a["rotations"]))
    kw = raster_kwargs(scene, cam, bg=case.get("bg", (0, 0, 0)), **extra)
    fr = oracle.rasterize(dtype=np.float64, **kw)

    tk = _to_t64(kw)
    leaves = {}
    for name in ("means3D", "opacities", "shs", "colors_precomp", "scales", "rotations", "cov3D_precomp"):
        if name in tk:
            tk[name] = tk[name].clone().requires_grad_(True)
            leaves[name] = tk[name]
    color, radii, proxy, _ = render_autograd(**tk)
    np.testing.assert_array_equal(radii.numpy(), fr.radii)
    assert np.abs(color.detach().numpy() - fr.color).max() < 1e-12

    gimg = S.make_grad_image(case["W"], case["H"], case["seed"]).double()
    (color * gimg).sum().backward()
    got = fr.backward(gimg.numpy())
    for name, leaf in leaves.items():
        want = leaf.grad.numpy().reshape(got[name].shape)
        scale = max(np.abs(want).max(), 1e-12)
        err = np.abs(got[name] - want).max()
        assert err <= 1e-9 * scale + 1e-12, f"{name}: {err:.3e} vs scale {scale:.3e}"
    want2d = proxy.grad.numpy()
    assert np.abs(got["means2D"][:, :2] - want2d).max() <= 1e-9 * max(np.abs(want2d).max(), 1e-12)
    assert np.all(got["means2D"][:, 2] == 0)


def _raw_settings(kw):
    return {k: kw[k] for k in ("image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix",
                               "projmatrix", "sh_degree", "campos")}


@pytest.mark.parametrize("case", CASES[:2] + CASES[3:], ids=lambda c: f"P{c['P']}_{c['W']}x{c['H']}_D{c['D']}")
def test_raw_leaves_entry_matches_torch_autograd_through_the_getters(case):
    """oracle.rasterize_raw = the getters of scene/gaussian_model.py:101-125 (exp / normalize / sigmoid / cat) in binary64 in
    front of the oracle and their chain rule behind its explicit backward: against torch autograd through the same torch
    getters and the independent restatement (tests/torch_ref.py), on the raw leaves."""
    scene, cam = _small_scene(case["P"], case["W"], case["H"], case["D"], case["seed"])
    kw = raster_kwargs(scene, cam, bg=case.get("bg", (0, 0, 0)))
    raw = dict(means3D=scene.means3D.numpy(), features=scene.shs.numpy(), opacity_logits=scene.opacity_logits.numpy(),
               log_scales=scene.log_scales.numpy(), raw_rotations=scene.raw_rotations.numpy())
    fr = oracle.rasterize_raw(**raw, **_raw_settings(kw))
    leaves = {k: torch.tensor(v, dtype=torch.float64).requires_grad_(True) for k, v in raw.items()}
    tk = _to_t64(_raw_settings(kw))
    color, radii, proxy, _ = render_autograd(
        means3D=leaves["means3D"], shs=torch.cat((leaves["features"][:, :1], leaves["features"][:, 1:]), dim=1),
        opacities=torch.sigmoid(leaves["opacity_logits"]), scales=torch.exp(leaves["log_scales"]),
        rotations=torch.nn.functional.normalize(leaves["raw_rotations"]), **tk)
    np.testing.assert_array_equal(radii.numpy(), fr.radii)
    assert np.abs(color.detach().numpy() - fr.color).max() < 1e-12
    gimg = S.make_grad_image(case["W"], case["H"], case["seed"]).double()
    (color * gimg).sum().backward()
    got = fr.backward(gimg.numpy())
    for leaf, name in (("means3D", "_xyz"), ("features", "_features"), ("opacity_logits", "_opacity"),
                       ("log_scales", "_scaling"), ("raw_rotations", "_rotation")):
        want = leaves[leaf].grad.numpy().reshape(got[name].shape)
        scale = max(np.abs(want).max(), 1e-12)
        err = np.abs(got[name] - want).max()
        assert scale > 1e-9 and err <= 1e-9 * scale + 1e-12, f"{name}: {err:.3e} vs scale {scale:.3e}"
    want2d = proxy.grad.numpy()
    assert np.abs(got["means2D"][:, :2] - want2d).max() <= 1e-9 * max(np.abs(want2d).max(), 1e-12)


def test_oracle_slab_split_sums_to_full():
    """Tile-row slabs (multi-GPU sharding): slab renders tile the image, slab gradients sum."""
    scene, cam = _small_scene(60, 64, 80, 2, 21)
    kw = raster_kwargs(scene, cam, bg=(0.2, 0.3, 0.4))
    full = oracle.rasterize(dtype=np.float64, **kw)
    gimg = S.make_grad_image(64, 80, 3).double().numpy()
    gfull = full.backward(gimg)
    color = np.zeros_like(full.color)
    gsum = None
    for rows in ((0, 2), (2, 3), (3, 5)):
        fr = oracle.rasterize(dtype=np.float64, tile_rows=rows, **kw)
        np.testing.assert_array_equal(fr.radii, full.radii)
        color += fr.color
        g = fr.backward(gimg)
        gsum = g if gsum is None else {k: gsum[k] + g[k] for k in g}
    assert np.abs(color - full.color).max() == 0
    for k in gfull:
        assert np.abs(gsum[k] - gfull[k]).max() <= 1e-12 * max(1.0, np.abs(gfull[k]).max()), k


def _single(opacity, scale, W=32, H=32, pos=(0.0, 0.0, 2.0), bg=(0, 0, 0), color=(0.8, 0.4, 0.2), dtype=np.float64):
    cam = S.make_camera(W, H)
    return oracle.rasterize(
        dtype=dtype, image_height=H, image_width=W, tanfovx=math.tan(cam.FoVx / 2), tanfovy=math.tan(cam.FoVy / 2),
        bg=np.array(bg), scale_modifier=1.0, viewmatrix=cam.world_view_transform.numpy(),
        projmatrix=cam.full_proj_transform.numpy(), sh_degree=0, campos=cam.camera_center.numpy(),
        means3D=np.array([pos]), opacities=np.array([[opacity]]), colors_precomp=np.array([color]),
        scales=np.array([[scale] * 3]), rotations=np.array([[1.0, 0, 0, 0]])), cam


def test_known_answer_single_isotropic_splat():
    """One isotropic Gaussian on the optical axis: closed-form alpha / T / colour at every pixel."""
    W = H = 32
    op, s, z = 0.6, 0.1, 2.0
    fr, cam = _single(op, s, W, H, (0, 0, z), bg=(0.1, 0.2, 0.3))
    focal = H / (2 * 0.5)
    var = (s * focal / z) ** 2 + 0.3                     # isotropic 2D variance + dilation (A.4)
    cx, cy = (W - 1) / 2.0, (H - 1) / 2.0                # ndc 0 -> pixel centre coordinate (A.2)
    assert fr.radii[0] == math.ceil(3 * math.sqrt(var))
    np.testing.assert_allclose(fr.xy[0], [cx, cy], atol=1e-9)
    ys, xs = np.mgrid[0:H, 0:W]
    alpha = np.minimum(0.99, op * np.exp(-0.5 * ((xs - cx) ** 2 + (ys - cy) ** 2) / var))
    alpha = np.where(alpha < 1 / 255, 0.0, alpha)
    for ch, (c, b) in enumerate(zip((0.8, 0.4, 0.2), (0.1, 0.2, 0.3))):
        np.testing.assert_allclose(fr.color[ch], c * alpha + (1 - alpha) * b, atol=1e-9)
    np.testing.assert_allclose(fr.final_T, 1 - alpha, atol=1e-12)


def test_known_answer_culling_and_clamps():
    # behind the near cut (A.1): invisible, image is background
    fr, _ = _single(0.9, 0.1, pos=(0, 0, 0.2), bg=(0.3, 0.3, 0.3))
    assert fr.radii[0] == 0 and fr.num_rendered == 0 and np.all(fr.color == 0.3)
    fr, _ = _single(0.9, 0.1, pos=(0, 0, 0.2001))
    assert fr.radii[0] > 0
    # opacity 1.0: alpha clamps at 0.99 at the centre (A.8)
    fr, _ = _single(1.0, 5.0, pos=(0.0, 0.0, 2.0), color=(1, 1, 1))
    assert abs(fr.color[0].max() - 0.99) < 1e-12
    # far off-screen: empty rect -> invisible (A.5)
    fr, _ = _single(0.9, 0.01, pos=(50.0, 0, 2.0))
    assert fr.radii[0] == 0


def test_known_answer_ordering_and_cutoff():
    """Two splats: nearer one composited first; a stack of opaque splats stops at T < 1e-4 (A.8)."""
    W = H = 16
    cam = S.make_camera(W, H)
    base = dict(image_height=H, image_width=W, tanfovx=math.tan(cam.FoVx / 2), tanfovy=math.tan(cam.FoVy / 2),
                bg=np.zeros(3), scale_modifier=1.0, viewmatrix=cam.world_view_transform.numpy(),
                projmatrix=cam.full_proj_transform.numpy(), sh_degree=0, campos=cam.camera_center.numpy())
    n = 6
    zs = np.array([3.0, 1.0, 2.0, 5.0, 4.0, 6.0])
    means = np.stack([np.zeros(n), np.zeros(n), zs], 1)
    cols = np.eye(3)[np.arange(n) % 3]
    fr = oracle.rasterize(dtype=np.float64, means3D=means, opacities=np.full((n, 1), 0.95), colors_precomp=cols,
                          scales=np.full((n, 3), 2.0), rotations=np.tile([1.0, 0, 0, 0], (n, 1)), **base)
    order = fr.point_list[:n]
    assert list(order) == list(np.argsort(zs, kind="stable"))
    # scales are huge -> alpha ~= 0.95 everywhere; T: 1, .05, .0025, 1.25e-4, then 6.25e-6 < 1e-4 stops
    a = 0.95 * math.exp(0.0)
    mid = fr.n_contrib[8, 8]
    assert mid == 3
    T = (1 - fr.conic_opacity[0, 3]) ** 3
    assert abs(fr.final_T[8, 8] - T) < 0.1 * T and a > 0   # alpha slightly below 0.95 off-centre


def test_stable_sort_ties_resolve_by_index():
    """Equal depth in the same tile: ascending Gaussian index (A.7)."""
    W = H = 16
    cam = S.make_camera(W, H)
    n = 5
    means = np.tile([0.0, 0.0, 2.5], (n, 1))
    fr = oracle.rasterize(dtype=np.float32, image_height=H, image_width=W, tanfovx=0.5, tanfovy=0.5, bg=np.zeros(3),
                          scale_modifier=1.0, viewmatrix=cam.world_view_transform.numpy(),
                          projmatrix=cam.full_proj_transform.numpy(), sh_degree=0, campos=np.zeros(3),
                          means3D=means, opacities=np.full((n, 1), 0.1), colors_precomp=np.random.rand(n, 3),
                          scales=np.full((n, 3), 0.05), rotations=np.tile([1.0, 0, 0, 0], (n, 1)))
    assert list(fr.point_list) == [0, 1, 2, 3, 4]


def test_oracle_f32_close_to_f64_outside_fragile_pixels():
    scene, cam = S.make_scene(4000, 128, 128, 3, 31, scale_lo=0.005, scale_hi=0.06), S.make_camera(128, 128)
    kw = raster_kwargs(scene, cam)
    f32, f64 = oracle.rasterize(dtype=np.float32, **kw), oracle.rasterize(dtype=np.float64, **kw)
    strict = f64.fragile_px == 0
    assert strict.mean() > 0.98
    err = np.abs(f32.color.astype(np.float64) - f64.color).max(0)
    assert err[strict].max() < 1e-5
    np.testing.assert_array_equal(f32.radii, f64.radii)
