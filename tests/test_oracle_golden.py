"""The oracle (and the host-side camera maths) against golden vectors produced by the reference's
own Python twins (tests/golden/make_golden.py).  These are the steps of the path whose
parity IS pinned by the reference: SH colour (+ its backward), camera matrices, point projection.
"""
import math
import os

import numpy as np
import pytest

import oracle
import scene_synth as S

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _sh_frame(D, gold, dtype):
    """Render a frame whose per-Gaussian colours are the golden SH evaluation: identity view,
    Gaussians at `pos`, camera centre (for the view direction only) at the golden campos."""
    cam = S.make_camera(64, 64)
    pos = gold["pos"]
    P = pos.shape[0]
    return oracle.rasterize(
        dtype=dtype, image_height=64, image_width=64, tanfovx=0.5, tanfovy=0.5, bg=np.zeros(3), scale_modifier=1.0,
        viewmatrix=cam.world_view_transform.numpy(), projmatrix=cam.full_proj_transform.numpy(), sh_degree=D,
        campos=gold["campos"], means3D=pos, opacities=np.full((P, 1), 0.5), shs=gold["sh"],
        scales=np.full((P, 3), 0.05), rotations=np.tile([1.0, 0, 0, 0], (P, 1)))


@pytest.mark.parametrize("D", [0, 1, 2, 3])
def test_sh_colour_matches_reference_eval_sh(D):
    gold = np.load(os.path.join(G, "sh_eval.npz"))
    fr = _sh_frame(D, gold, np.float64)
    vis = fr.radii > 0
    assert vis.sum() > 60
    np.testing.assert_allclose(fr.rgb[vis], gold[f"color_D{D}"][vis], atol=1e-13)
    np.testing.assert_array_equal(fr.clamped[vis].astype(bool), (gold[f"raw_D{D}"][vis] + 0.5) < 0)
    fr32 = _sh_frame(D, gold, np.float32)
    np.testing.assert_allclose(fr32.rgb[vis], gold[f"color_D{D}"][vis], atol=3e-6)


@pytest.mark.parametrize("D", [0, 1, 2, 3])
def test_sh_backward_matches_reference_autograd(D):
    """dL/dsh and the view-direction term of dL/dmeans3D (A.10) = autograd of the reference's
    eval_sh + clamp_min path (gaussian_renderer/__init__.py:73-78)."""
    gold = np.load(os.path.join(G, "sh_eval.npz"))
    fr = _sh_frame(D, gold, np.float64)
    vis = fr.radii > 0
    screen = np.zeros((fr.P, 9))
    screen[:, 6:9] = gold["upstream"]
    got = fr.backward_geom(screen)
    np.testing.assert_allclose(got["shs"][vis], gold[f"grad_sh_D{D}"][vis], atol=1e-13)
    np.testing.assert_allclose(got["means3D"][vis], gold[f"grad_pos_D{D}"][vis], atol=1e-12)
    assert np.all(got["shs"][~vis] == 0)


def test_sh_dc_constants():
    gold = np.load(os.path.join(G, "sh_eval.npz"))
    assert float(gold["C0"]) == S.SH_C0
    np.testing.assert_allclose((gold["rgb"] - 0.5) / S.SH_C0, gold["rgb2sh"], atol=1e-15)
    np.testing.assert_allclose(gold["rgb"] * S.SH_C0 + 0.5, gold["sh2rgb"], atol=1e-15)


def test_camera_matrices_match_reference():
    gold = np.load(os.path.join(G, "camera.npz"))
    for i in range(int(gold["n"])):
        W, H = (int(v) for v in gold[f"WH{i}"])
        cam = S.make_camera(W, H, gold[f"R{i}"], gold[f"t{i}"], tanfovy=float(gold[f"tanfovy{i}"]))
        np.testing.assert_allclose(cam.world_view_transform.numpy(), gold[f"wvt{i}"], atol=1e-7)
        np.testing.assert_allclose(cam.full_proj_transform.numpy(), gold[f"full{i}"], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose(cam.camera_center.numpy(), gold[f"center{i}"], atol=1e-6)
        fx, fy = gold[f"focal{i}"]
        assert abs(fx - W / (2 * math.tan(cam.FoVx / 2))) < 1e-9 * fx and abs(fy - H / (2 * math.tan(cam.FoVy / 2))) < 1e-9 * fy


def test_oracle_projection_matches_reference_point_transform():
    """xy/depth of the oracle's preprocess = geom_transform_points with the reference's matrices (A.0-A.2)."""
    gold = np.load(os.path.join(G, "camera.npz"))
    checked = 0
    for i in range(int(gold["n"])):
        W, H = (int(v) for v in gold[f"WH{i}"])
        tanfovy = float(gold[f"tanfovy{i}"])
        pts = gold[f"pts{i}"].astype(np.float64)
        P = pts.shape[0]
        # move the cloud in front of THIS camera: use camera-space points mapped back to world
        wvt = gold[f"wvt{i}"].astype(np.float64)
        fr = oracle.rasterize(
            dtype=np.float64, image_height=H, image_width=W, tanfovx=tanfovy * W / H, tanfovy=tanfovy, bg=np.zeros(3),
            scale_modifier=1.0, viewmatrix=wvt, projmatrix=gold[f"full{i}"].astype(np.float64), sh_degree=0,
            campos=gold[f"center{i}"].astype(np.float64), means3D=pts, opacities=np.full((P, 1), 0.5),
            colors_precomp=np.full((P, 3), 0.5), scales=np.full((P, 3), 0.05), rotations=np.tile([1.0, 0, 0, 0], (P, 1)))
        vis = fr.radii > 0
        view, ndc = gold[f"pts_view{i}"], gold[f"pts_ndc{i}"]
        # geom_transform_points divides view-space coordinates by (w + 1e-7) with w = 1
        np.testing.assert_allclose(fr.depth[vis], view[vis, 2] * (1 + 1e-7), rtol=2e-6)
        px = ((ndc[:, 0] + 1) * W - 1) * 0.5
        py = ((ndc[:, 1] + 1) * H - 1) * 0.5
        np.testing.assert_allclose(fr.xy[vis, 0], px[vis], rtol=2e-5, atol=2e-3)
        np.testing.assert_allclose(fr.xy[vis, 1], py[vis], rtol=2e-5, atol=2e-3)
        assert not np.any(vis & (view[:, 2] < 0.19))
        checked += int(vis.sum())
    assert checked > 20


def test_cov3d_matches_reference_build_scaling_rotation():
    """A.3 pinned by the reference's own functions (tests/golden/cov3d.npz from utils/general_utils.py:64-110 and the
    covariance builder of scene/gaussian_model.py:25-29): quaternion -> R, Sigma = (R S)(R S)^T, packing
    [xx, xy, xz, yy, yz, zz] — against the oracle's cov3D (both precisions), the product's gsr_math.h (through the
    g++ harness) and the host-side GaussianModel / GaussianParams.get_covariance."""
    import ctypes as C
    import subprocess
    import tempfile

    import torch
    gold = np.load(os.path.join(G, "cov3d.npz"))
    q_raw, s = gold["rotation"].astype(np.float64), gold["scaling"].astype(np.float64)
    qn = q_raw / np.linalg.norm(q_raw, axis=1, keepdims=True)          # the caller normalises (scene/gaussian_model.py:106-109)
    P = qn.shape[0]
    cam = S.make_camera(64, 64)
    means = np.tile([0.0, 0.0, 3.0], (P, 1)) + np.random.default_rng(0).normal(0, 0.2, (P, 3))
    for j, mod in enumerate(gold["modifiers"]):
        want = gold[f"cov6_{j}"].astype(np.float64)
        for dtype, tol in ((np.float64, 2e-6), (np.float32, 2e-6)):      # the golden itself is the reference's float32 result
            fr = oracle.rasterize(dtype=dtype, image_height=64, image_width=64, tanfovx=0.5, tanfovy=0.5, bg=np.zeros(3),
                                  scale_modifier=float(mod), viewmatrix=cam.world_view_transform.numpy(),
                                  projmatrix=cam.full_proj_transform.numpy(), sh_degree=0, campos=np.zeros(3), means3D=means,
                                  opacities=np.full((P, 1), 0.5), colors_precomp=np.full((P, 3), 0.5), scales=s, rotations=qn)
            vis = fr.radii > 0
            assert vis.sum() > P // 2
            scale = np.abs(want[vis]).max(1, keepdims=True)
            assert (np.abs(fr.cov3D[vis] - want[vis]) <= tol * scale + 1e-12).all()
        # host-side builders (torch): util.cov3d_from (used by the parity tests), GaussianParams / GaussianModel.get_covariance
        from util import cov3d_from
        got = cov3d_from(torch.tensor(s, dtype=torch.float32), torch.tensor(qn, dtype=torch.float32), float(mod)).numpy()
        assert (np.abs(got - want) <= 3e-6 * np.abs(want).max(1, keepdims=True) + 1e-12).all()
        from scene import GaussianModel
        gm = GaussianModel(0)
        gm._t["scaling"] = torch.log(torch.tensor(s, dtype=torch.float32))
        gm._t["rotation"] = torch.tensor(q_raw, dtype=torch.float32)   # raw: the model's getter path normalises like build_rotation
        got = gm.get_covariance(float(mod)).numpy()
        assert (np.abs(got - want) <= 5e-6 * np.abs(want).max(1, keepdims=True) + 1e-12).all()
    # rotation matrices and the un-packed L = R S
    r, x, y, z = qn.T
    R = np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y), 2 * (x * y + r * z), 1 - 2 * (x * x + z * z),
                  2 * (y * z - r * x), 2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).reshape(P, 3, 3)
    np.testing.assert_allclose(R, gold["R"], atol=2e-6)
    np.testing.assert_allclose(R * s[:, None, :], gold["L_0"], rtol=1e-5, atol=1e-7 * float(np.abs(gold["L_0"]).max()))      # float32 golden
    # the product's header, compiled for the host
    here = os.path.dirname(os.path.abspath(__file__))
    with tempfile.TemporaryDirectory() as d:
        so = os.path.join(d, "libhh.so")
        subprocess.check_call(["g++", "-O2", "-fPIC", "-shared", "-std=c++17", "-o", so, os.path.join(here, "host_harness.cpp")])
        hh = C.CDLL(so)
        out = np.zeros((P, 6), np.float32)
        s32, q32 = np.ascontiguousarray(s, np.float32), np.ascontiguousarray(qn, np.float32)
        for j, mod in enumerate(gold["modifiers"]):
            hh.hh_cov3d(P, s32.ctypes.data_as(C.c_void_p), q32.ctypes.data_as(C.c_void_p), C.c_float(float(mod)),
                        out.ctypes.data_as(C.c_void_p))
            want = gold[f"cov6_{j}"]
            assert (np.abs(out - want) <= 3e-6 * np.abs(want).max(1, keepdims=True) + 1e-12).all()
