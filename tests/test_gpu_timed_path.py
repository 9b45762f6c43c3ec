"""The path bench.py TIMES, held to the strict oracle bound.

`bench.py`'s step renders this package's scene.GaussianModel through render()'s default branch for that class: the RAW leaves
(_xyz, _features [P,M,3], _opacity, _scaling, _rotation) go into the kernels (GaussianRasterizer.forward_raw, ABI raw mode 2:
exp / normalize / sigmoid inside k_preprocess, their chain rule inside the geometry backward).  The oracle's raw-leaves entry
(oracle.rasterize_raw: the getters of scene/gaussian_model.py:101-125 in binary64 in front of the oracle, their chain rule behind
its explicit backward) is the checker: pixels 1e-5 on non-fragile pixels, EVERY Gaussian's gradient on EVERY raw leaf at
1e-4 (test_gpu_parity._forward_backward_strict's bound), dL/dcolor zero on the oracle's fragile pixels on both sides.

Cases: the small fixtures, cfg2, cfg3 at full size, cfg3n at full size (the non-saturating workload: > 800 k Gaussians carry a
gradient), and one 1M / 1080p frame of the training loop's arc cameras (an uncovered corner: chunk merge + live filter at scale).
"""
import numpy as np
import pytest
import torch

import oracle
import scene_synth as S
from test_gpu_parity import DEV, GRAD_ATOL_REL, GRAD_RTOL, _check_forward, _check_grads, _strict_pixels
from util import raster_kwargs

pytestmark = pytest.mark.gpu

RAW_NAMES = ("_xyz", "_features", "_opacity", "_scaling", "_rotation", "means2D")
SETTINGS = ("image_height", "image_width", "tanfovx", "tanfovy", "bg", "scale_modifier", "viewmatrix", "projmatrix", "sh_degree", "campos")


def _oracle_raw(scene, cam, bg=(0.0, 0.0, 0.0), parallel=True, dtype=np.float64):
    kw = raster_kwargs(scene, cam, bg=bg)
    return oracle.rasterize_raw(dtype=dtype, means3D=scene.means3D.numpy(), features=scene.shs.numpy(), opacity_logits=scene.opacity_logits.numpy(),
                                log_scales=scene.log_scales.numpy(), raw_rotations=scene.raw_rotations.numpy(), parallel=parallel,
                                **{k: kw[k] for k in SETTINGS})


def _render_timed_path(scene, cam, bg, grad_img):
    """What bench.py's step() does for the rasterizer: render(cam, scene.GaussianModel, Pipe(), bg) -> backward.  Asserts that
    the call really took the raw-leaves branch (no getter ran)."""
    import diff_gaussian_rasterization as dgr
    from gaussian_params import Pipe
    from gaussian_renderer import render
    from scene import GaussianModel
    model = GaussianModel(scene.sh_degree)
    model.adopt_scene(scene, device=DEV)
    seen = []
    orig = dgr.GaussianRasterizer.forward_raw

    def spy(self, *a, **k):
        seen.append(a[3] is None)                     # features_rest=None <=> raw mode 2 (one interleaved table)
        return orig(self, *a, **k)
    dgr.GaussianRasterizer.forward_raw = spy
    try:
        out = render(cam.to(DEV), model, Pipe(), torch.as_tensor(bg, dtype=torch.float32, device=DEV))
    finally:
        dgr.GaussianRasterizer.forward_raw = orig
    assert seen == [True], "render() did not take the raw-leaves (raw mode 2) branch bench.py times"
    color, radii = out["render"], out["radii"]
    g = grad_img(color.detach().cpu().numpy(), radii.cpu().numpy())
    color.backward(torch.as_tensor(np.ascontiguousarray(g), dtype=torch.float32, device=DEV))
    torch.cuda.synchronize()
    grads = {"_xyz": model._xyz.grad, "_features": model._features.grad, "_opacity": model._opacity.grad,
             "_scaling": model._scaling.grad, "_rotation": model._rotation.grad, "means2D": out["viewspace_points"].grad}
    return color.detach().cpu().numpy(), radii.cpu().numpy(), {k: v.detach().cpu().numpy() for k, v in grads.items()}


def _timed_path_strict(scene, cam, gimg, bg=(0.0, 0.0, 0.0), label="", parallel=True, deep=False):
    fr = _oracle_raw(scene, cam, bg, parallel)
    fr32 = _oracle_raw(scene, cam, bg, parallel, dtype=np.float32) if deep else None       # see _check_forward
    state = {}

    def masked(color, radii):
        # fp32 exp / sigmoid / normalize in the kernel against binary64 in the oracle: a radius may round across a ceil()
        state["strict"] = _strict_pixels(fr, radii, exact_radii=False)
        return np.where(state["strict"][None], gimg, 0.0).astype(np.float32)
    color, radii, grads = _render_timed_path(scene, cam, bg, masked)
    _check_forward(None, fr, color, radii, exact_radii=False, fr32=fr32)
    want = fr.backward(np.where(state["strict"][None], gimg, 0.0).astype(np.float64), parallel=parallel)
    live, strict_live = _check_grads(fr, want, grads, list(RAW_NAMES), masked=True)
    assert strict_live == live
    print(f"{label} [timed path, raw leaves]: {live} Gaussians with a non-zero gradient, {strict_live} held to "
          f"{GRAD_ATOL_REL:g}*scale + {GRAD_RTOL:g}*|w| on every raw leaf; {int((~state['strict']).sum())} of {state['strict'].size} pixels masked")
    return live


@pytest.mark.parametrize("P,W,H,D,seed,bg", [(1, 32, 32, 0, 101, (0, 0, 0)), (64, 48, 80, 2, 103, (0, 0, 0)),
                                             (64, 80, 48, 3, 104, (1.0, 1.0, 1.0)), (2048, 128, 128, 3, 105, (0, 0, 0)),
                                             (2048, 100, 60, 1, 107, (0.2, 0.4, 0.6)), (5000, 256, 192, 3, 109, (0, 0, 0))])
def test_timed_path_fixtures_vs_raw_oracle(P, W, H, D, seed, bg):
    lo, hi = (0.01, 0.2) if P <= 64 else (0.005, 0.06)
    scene, cam = S.make_scene(P, W, H, D, seed, scale_lo=lo, scale_hi=hi), S.make_camera(W, H)
    if P == 1:
        scene.means3D[:] = torch.tensor([[0.05, -0.03, 2.0]])
    live = _timed_path_strict(scene, cam, S.make_grad_image(W, H, seed).numpy(), bg, label=f"fixture P={P} {W}x{H} D={D}", parallel=False)
    assert live >= 1


@pytest.mark.parametrize("workload,min_live", [("cfg2", 1000), ("cfg3", 1000), ("cfg3n", 800_000)])
def test_timed_path_full_size_vs_raw_oracle(workload, min_live):
    """BASELINE configs[1] / configs[2] scenes and the non-saturating cfg3n at FULL size through the timed path."""
    c = S.CONFIGS[workload]
    scene, cam = S.make_config(workload)
    live = _timed_path_strict(scene, cam, S.make_grad_image(c["W"], c["H"], c["seed"]).numpy(), label=workload, deep=workload == "cfg3n")
    assert live >= min_live, live


def test_timed_path_training_loop_arc_frame_full_size():
    """One 1M-Gaussian / 1080p frame as the training loop of bench.py sees it (scene_synth.arc_cameras: the camera off the
    axis, a corner no splat covers -> those tiles never close, every planned chunk runs: chunk merge and live filter at scale)."""
    c = S.CONFIGS["cfg3"]
    scene, _ = S.make_config("cfg3")
    cam = S.arc_cameras(c["W"], c["H"], 8)[0]
    live = _timed_path_strict(scene, cam, S.make_grad_image(c["W"], c["H"], 11).numpy(), label="cfg3 arc camera 0", deep=True)
    assert live > 1000, live
