"""CPU check of the product's per-Gaussian maths header (csrc/gsr_math.h — the functions the HIP
preprocess / geometry-backward kernels wrap) against the oracle, via a g++-compiled test harness
(tests/host_harness.cpp).  Host-logic coverage for `-m "not gpu"`; the kernels themselves are
checked on the GPU in test_gpu_parity.py.
"""
import ctypes as C
import math
import os
import subprocess

import numpy as np
import pytest
import torch

import oracle
import scene_synth as S
from util import cov3d_from, raster_kwargs, unscale_records

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def hh(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("hh") / "libhost_harness.so")
    subprocess.check_call(["g++", "-O2", "-fPIC", "-shared", "-std=c++17", "-o", so, os.path.join(HERE, "host_harness.cpp")])
    return C.CDLL(so)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, np.float32)


def _run(hh, kw, slab=(0, 0)):
    P = kw["means3D"].shape[0]
    shs = _f32(kw.get("shs")); M = 0 if shs is None else shs.shape[1]
    W, H, D = kw["image_width"], kw["image_height"], kw["sh_degree"]
    V, PV, cam = _f32(kw["viewmatrix"]), _f32(kw["projmatrix"]), _f32(kw["campos"])
    means, sc, ro = _f32(kw["means3D"]), _f32(kw.get("scales")), _f32(kw.get("rotations"))
    cp, op, col = _f32(kw.get("cov3D_precomp")), _f32(kw["opacities"]).reshape(-1), _f32(kw.get("colors_precomp"))
    radii = np.zeros(P, np.int32); tiles = np.zeros(P, np.uint32); cl = np.zeros(P, np.uint8); rec = np.zeros((P, 12), np.float32)
    hh.hh_preprocess(P, D, M, W, H, C.c_float(kw["tanfovx"]), C.c_float(kw["tanfovy"]), C.c_float(kw["scale_modifier"]),
                     slab[0], slab[1], _p(V), _p(PV), _p(cam), _p(means), _p(sc), _p(ro), _p(cp), _p(op), _p(shs), _p(col),
                     _p(radii), _p(tiles), _p(cl), _p(rec))
    return dict(radii=radii, tiles=tiles, clamped=cl, rec=rec, args=(P, D, M, W, H, V, PV, cam, means, sc, ro, cp, shs, col))


@pytest.mark.parametrize("mode", ["sh+scale", "color+cov"])
@pytest.mark.parametrize("D", [0, 3])
def test_preprocess_matches_oracle(hh, mode, D):
    scene, cam = S.make_scene(3000, 160, 112, D, 41 + D, scale_lo=0.005, scale_hi=0.08), S.make_camera(160, 112)
    a = scene.activated()
    extra = {}
    if mode == "color+cov":
        extra = dict(colors_precomp=torch.rand(scene.P, 3, generator=torch.Generator().manual_seed(1)),
                     cov3D_precomp=cov3d_from(a["scales"], a["rotations"]))
    kw = raster_kwargs(scene, cam, **extra)
    fr = oracle.rasterize(dtype=np.float32, **kw)
    got = _run(hh, kw)
    np.testing.assert_array_equal(got["radii"], fr.radii)
    # the binning rectangle is the reference's A.5 rectangle clipped to the alpha >= 1/255 bounding box
    assert np.all(got["tiles"] <= fr.tiles_touched) and got["tiles"].sum() < fr.tiles_touched.sum()
    rx = got["rec"][:, 10].copy().view(np.uint32); ry = got["rec"][:, 11].copy().view(np.uint32)
    x0, x1, y0, y1 = rx & 0xFFFF, rx >> 16, ry & 0xFFFF, ry >> 16
    v = fr.radii > 0
    assert np.all(x0[v] >= fr.rect[v, 0]) and np.all(x1[v] <= np.maximum(fr.rect[v, 2], x0[v]))
    assert np.all(y0[v] >= fr.rect[v, 1]) and np.all(y1[v] <= np.maximum(fr.rect[v, 3], y0[v]))
    np.testing.assert_array_equal(((x1 - x0) * (y1 - y0))[v], got["tiles"][v])
    vis = fr.radii > 0
    assert vis.sum() > 1000
    rec = unscale_records(got["rec"])[vis]
    np.testing.assert_allclose(rec[:, 0:2], fr.xy[vis], rtol=1e-6, atol=1e-4)
    np.testing.assert_allclose(rec[:, 2:5], fr.conic_opacity[vis, :3], rtol=2e-5, atol=1e-7)
    np.testing.assert_allclose(rec[:, 5], fr.conic_opacity[vis, 3], rtol=2e-6, atol=0)
    np.testing.assert_allclose(rec[:, 6:9], fr.rgb[vis], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(rec[:, 9], fr.depth[vis], rtol=1e-6)
    bits = (fr.clamped[:, 0] | (fr.clamped[:, 1] << 1) | (fr.clamped[:, 2] << 2)).astype(np.uint8)
    np.testing.assert_array_equal(got["clamped"][vis], bits[vis])
    # slab clipping only changes the tile count
    slab = _run(hh, kw, slab=(2, 5))
    frs = oracle.rasterize(dtype=np.float32, tile_rows=(2, 5), **kw)
    assert np.all(slab["tiles"] <= frs.tiles_touched) and np.all(slab["tiles"] <= got["tiles"])
    np.testing.assert_array_equal(slab["radii"], fr.radii)


@pytest.mark.parametrize("mode", ["sh+scale", "color+cov"])
def test_geom_backward_matches_oracle(hh, mode):
    D = 3
    scene, cam = S.make_scene(2000, 128, 96, D, 77, scale_lo=0.005, scale_hi=0.08), S.make_camera(128, 96)
    a = scene.activated()
    extra = {}
    if mode == "color+cov":
        extra = dict(colors_precomp=torch.rand(scene.P, 3, generator=torch.Generator().manual_seed(1)),
                     cov3D_precomp=cov3d_from(a["scales"], a["rotations"]))
    kw = raster_kwargs(scene, cam, **extra)
    fr = oracle.rasterize(dtype=np.float64, **kw)
    gimg = S.make_grad_image(128, 96, 5).numpy()
    want = fr.backward(gimg)
    got_f = _run(hh, kw)
    P, D, M, W, H, V, PV, camc, means, sc, ro, cp, shs, col = got_f["args"]
    screen = np.ascontiguousarray(want["screen"], np.float32)
    out = dict(means3D=np.zeros((P, 3), np.float32), means2D=np.zeros((P, 3), np.float32),
               shs=np.zeros((P, max(M, 1), 3), np.float32), colors_precomp=np.zeros((P, 3), np.float32),
               opacities=np.zeros(P, np.float32), scales=np.zeros((P, 3), np.float32),
               rotations=np.zeros((P, 4), np.float32), cov3D_precomp=np.zeros((P, 6), np.float32))
    hh.hh_geom_backward(P, D, M, W, H, C.c_float(kw["tanfovx"]), C.c_float(kw["tanfovy"]), C.c_float(1.0), _p(V), _p(PV),
                        _p(camc), _p(means), _p(sc), _p(ro), _p(cp), _p(shs), int(col is not None), _p(got_f["radii"]),
                        _p(got_f["clamped"]), _p(screen), _p(out["means3D"]), _p(out["means2D"]), _p(out["shs"]),
                        _p(out["colors_precomp"]), _p(out["opacities"]), _p(out["scales"]), _p(out["rotations"]),
                        _p(out["cov3D_precomp"]))
    names = ["means3D", "means2D", "opacities"] + (["colors_precomp", "cov3D_precomp"] if mode == "color+cov" else ["shs", "scales", "rotations"])
    for n in names:
        w = want[n].reshape(out[n].shape) if n != "shs" else want[n]
        g = out[n] if n != "shs" else out[n][:, :M]
        scale = np.abs(w).max()
        assert scale > 0, n
        # fp32 evaluation of a cancellation-prone chain: 1e-4 of the tensor's scale, 2e-3 relative per element
        err = np.abs(g - w)
        assert (err <= 1e-4 * scale + 2e-3 * np.abs(w)).all(), f"{n}: max err {err.max():.3e} scale {scale:.3e}"


def test_tile_culling_is_conservative(hh):
    """csrc/gsr_math.h tile_may_contribute(): whenever it says "no", the oracle (fp32 AND fp64) finds no pixel
    of that tile that accepts the splat; and it does cull a useful share of the rectangle instances."""
    scene, cam = S.make_scene(6000, 256, 192, 0, 91, scale_lo=0.004, scale_hi=0.12), S.make_camera(256, 192)
    kw = raster_kwargs(scene, cam)
    fr = oracle.rasterize(dtype=np.float32, **kw)
    fr64 = oracle.rasterize(dtype=np.float64, **kw)
    vis = np.nonzero(fr.radii > 0)[0]
    gs, txs, tys = [], [], []
    for g in vis:
        x0, y0, x1, y1 = fr.rect[g]
        yy, xx = np.mgrid[y0:y1, x0:x1]
        gs.append(np.full(xx.size, g)); txs.append(xx.ravel()); tys.append(yy.ravel())
    g = np.concatenate(gs); tx = np.concatenate(txs).astype(np.int32); ty = np.concatenate(tys).astype(np.int32)
    co = fr.conic_opacity[g].astype(np.float32)
    sx, sy = (np.ascontiguousarray(fr.xy[g, i], np.float32) for i in (0, 1))
    A, B, Cc, op = (np.ascontiguousarray(co[:, i]) for i in range(4))
    out = np.zeros(g.size, np.uint8)
    hh.hh_tile_may_contribute(g.size, _p(sx), _p(sy), _p(A), _p(B), _p(Cc), _p(op), _p(tx), _p(ty), _p(out))
    # the preprocess's tight rectangle removes instances too: those count as culled as well
    got = _run(hh, kw)
    rx = got["rec"][:, 10].copy().view(np.uint32); ry = got["rec"][:, 11].copy().view(np.uint32)
    inside = (tx >= (rx & 0xFFFF)[g]) & (tx < (rx >> 16)[g]) & (ty >= (ry & 0xFFFF)[g]) & (ty < (ry >> 16)[g])
    assert 0.05 < (~inside).mean() < 0.9
    out[~inside] = 0
    culled = np.nonzero(out == 0)[0]
    assert 0.2 < culled.size / g.size < 0.9, culled.size / g.size
    # brute force on the culled instances: max over the tile's pixels of alpha, both precisions
    px = (tx[culled, None] * 16 + np.arange(16)[None, :])[:, None, :]          # [n,1,16]
    py = (ty[culled, None] * 16 + np.arange(16)[None, :])[:, :, None]          # [n,16,1]
    for frx in (fr, fr64):
        co_ = frx.conic_opacity[g[culled]].astype(np.float64)
        dx = frx.xy[g[culled], 0].astype(np.float64)[:, None, None] - px
        dy = frx.xy[g[culled], 1].astype(np.float64)[:, None, None] - py
        power = -0.5 * (co_[:, 0, None, None] * dx * dx + co_[:, 2, None, None] * dy * dy) - co_[:, 1, None, None] * dx * dy
        alpha = np.minimum(0.99, co_[:, 3, None, None] * np.exp(np.minimum(power, 0)))
        accept = (power <= 0) & (alpha >= (1 / 255) * (1 - 1e-3))
        assert not accept.any(), f"{accept.any((1, 2)).sum()} culled instances have an accepting pixel"


def test_quadrant_culling_is_conservative(hh):
    """csrc/gsr_math.h quadrant_mask_q() (sub-tile culling of the blend kernels, on the pre-scaled splat record): a clear bit
    means that NO pixel of that 8x8 quadrant accepts the splat in the oracle (fp32 and fp64); and for small splats it does
    clear a useful share of the quadrants of the (splat, tile) instances that survive the tile-level test."""
    scene, cam = S.make_scene(8000, 256, 192, 0, 93, scale_lo=0.002, scale_hi=0.05), S.make_camera(256, 192)
    kw = raster_kwargs(scene, cam)
    fr = oracle.rasterize(dtype=np.float32, **kw)
    fr64 = oracle.rasterize(dtype=np.float64, **kw)
    got = _run(hh, kw)
    vis = np.nonzero(fr.radii > 0)[0]
    gs, txs, tys = [], [], []
    for g in vis:
        x0, y0, x1, y1 = fr.rect[g]
        yy, xx = np.mgrid[y0:y1, x0:x1]
        gs.append(np.full(xx.size, g)); txs.append(xx.ravel()); tys.append(yy.ravel())
    g = np.concatenate(gs); tx = np.concatenate(txs).astype(np.int32); ty = np.concatenate(tys).astype(np.int32)
    rec = np.ascontiguousarray(got["rec"][g], np.float32)
    mask = np.zeros(g.size, np.uint8)
    hh.hh_quadrant_mask(g.size, _p(rec), _p(tx), _p(ty), _p(mask))
    # brute force: per instance and quadrant, does any pixel accept?
    px = (tx[:, None] * 16 + np.arange(16)[None, :])[:, None, :]
    py = (ty[:, None] * 16 + np.arange(16)[None, :])[:, :, None]
    for frx in (fr, fr64):
        co_ = frx.conic_opacity[g].astype(np.float64)
        dx = frx.xy[g, 0].astype(np.float64)[:, None, None] - px
        dy = frx.xy[g, 1].astype(np.float64)[:, None, None] - py
        power = -0.5 * (co_[:, 0, None, None] * dx * dx + co_[:, 2, None, None] * dy * dy) - co_[:, 1, None, None] * dx * dy
        alpha = np.minimum(0.99, co_[:, 3, None, None] * np.exp(np.minimum(power, 0)))
        accept = (power <= 0) & (alpha >= (1 / 255) * (1 - 1e-3))                         # [n, 16 (y), 16 (x)]
        for k in range(4):
            qa = accept[:, (k >> 1) * 8:(k >> 1) * 8 + 8, (k & 1) * 8:(k & 1) * 8 + 8].any((1, 2))
            clear = (mask >> k) & 1 == 0
            assert not (qa & clear).any(), f"quadrant {k}: {(qa & clear).sum()} culled quadrants hold an accepting pixel"
    n_set = sum(((mask >> k) & 1).sum() for k in range(4))
    reach = mask != 0
    assert 0.3 < n_set / (4.0 * reach.sum()) < 0.85, n_set / (4.0 * reach.sum())
    # the cheap bounding-box form (k_emit_team) never clears a bit the exact form sets, and is not much looser
    bbox = np.zeros(g.size, np.uint8)
    hh.hh_quadrant_mask_bbox(g.size, _p(rec), _p(tx), _p(ty), _p(bbox))
    assert ((mask & ~bbox) == 0).all()
    n_bbox = sum(((bbox[reach] >> k) & 1).sum() for k in range(4))
    assert n_bbox <= 1.3 * n_set, (n_bbox, n_set)


def test_raw_activations_match_torch_autograd(hh):
    """csrc/gsr_math.h activate_raw / activate_raw_backward (raw-parameter mode) against torch
    autograd through the reference's activations (scene/gaussian_model.py:47-60: exp, sigmoid,
    torch.nn.functional.normalize), including a zero quaternion (normalize's eps branch)."""
    n = 500
    g = torch.Generator().manual_seed(3)
    ls = (torch.rand(n, 3, generator=g) * 8 - 7).requires_grad_(True)
    rq = torch.randn(n, 4, generator=g)
    rq[0] = 0.0
    rq[1] *= 1e-3
    rq.requires_grad_(True)
    lo = (torch.randn(n, generator=g) * 3).requires_grad_(True)
    d_s, d_q, d_o = torch.randn(n, 3, generator=g), torch.randn(n, 4, generator=g), torch.randn(n, generator=g)
    s, q, o = torch.exp(ls), torch.nn.functional.normalize(rq), torch.sigmoid(lo)
    ((s * d_s).sum() + (q * d_q).sum() + (o * d_o).sum()).backward()
    f = lambda t: np.ascontiguousarray(t.detach().numpy(), np.float32)
    out = {k: np.zeros(shape, np.float32) for k, shape in dict(s=(n, 3), q=(n, 4), o=(n,), gls=(n, 3), grq=(n, 4), glo=(n,)).items()}
    hh.hh_activate_raw(n, _p(f(ls)), _p(f(rq)), _p(f(lo)), _p(f(d_s)), _p(f(d_q)), _p(f(d_o)), _p(out["s"]), _p(out["q"]),
                       _p(out["o"]), _p(out["gls"]), _p(out["grq"]), _p(out["glo"]))
    np.testing.assert_allclose(out["s"], f(s), rtol=2e-6)
    np.testing.assert_allclose(out["q"], f(q), rtol=2e-6, atol=1e-7)
    np.testing.assert_allclose(out["o"], f(o), rtol=2e-6, atol=1e-8)
    np.testing.assert_allclose(out["gls"], f(ls.grad), rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(out["glo"], f(lo.grad), rtol=1e-5, atol=1e-7)
    # row 0 (zero quaternion) divides by eps = 1e-12 on both sides; compare relative to the row's scale
    err = np.abs(out["grq"] - f(rq.grad))
    assert (err <= 1e-5 * np.abs(f(rq.grad)).max(axis=1, keepdims=True) + 1e-6).all()
