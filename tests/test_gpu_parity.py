"""GPU parity tests proper: the HIP path, called through the C ABI (libgsrast.so via the drop-in Python
package), against the CPU oracle on the same seeded inputs.

Tolerances (BASELINE.json north_star): pixels 1e-5, gradients 1e-4 (relative to the tensor's scale,
with a per-element relative term).  Integer / index work is bit-exact: radii, tiles touched, prefix
sums, the sorted per-tile splat lists, tile ranges, n_contrib.

"Fragile" pixels: the blend takes discrete decisions (alpha < 1/255, T < 1e-4, power > 0).  Where the
fp64 oracle finds a decision within `fragile_eps` of its threshold, an fp32 implementation may
legitimately decide the other way; those pixels (a fraction of a percent, asserted below) are checked
against a bound of one skipped/added splat instead of 1e-5.

Gradients: dL/dcolor is set to ZERO on the oracle's fragile pixels for BOTH the HIP run and the oracle
(`_forward_backward_strict`), so no gradient term depends on a decision that may legitimately flip, and
EVERY Gaussian is held to |err| <= 1e-4 * max|want| + 1e-4 * |want| (GRAD_ATOL_REL, GRAD_RTOL: the
north_star's 1e-4).  The tests print how many Gaussians with a non-zero gradient were checked that way.
The un-masked gradient image is kept for the small fixtures only, where Gaussians that reach a fragile
pixel (oracle `fragile_g`) get a bound of one flipped splat.
"""
import math
import os

import numpy as np
import pytest
import torch

import oracle
import scene_synth as S
from util import cov3d_from, raster_kwargs, unscale_records

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


def _settings(kw, debug=False):
    from diff_gaussian_rasterization import GaussianRasterizationSettings
    t = lambda a: torch.as_tensor(np.asarray(a), dtype=torch.float32).to(DEV)
    return GaussianRasterizationSettings(
        image_height=kw["image_height"], image_width=kw["image_width"], tanfovx=kw["tanfovx"], tanfovy=kw["tanfovy"],
        bg=t(kw["bg"]), scale_modifier=kw["scale_modifier"], viewmatrix=t(kw["viewmatrix"]),
        projmatrix=t(kw["projmatrix"]), sh_degree=kw["sh_degree"], campos=t(kw["campos"]), prefiltered=False,
        debug=debug)


def _inputs(kw, requires_grad=True):
    out = {}
    for k in ("means3D", "opacities", "shs", "colors_precomp", "scales", "rotations", "cov3D_precomp"):
        if kw.get(k) is not None:
            out[k] = torch.as_tensor(np.asarray(kw[k]), dtype=torch.float32).to(DEV).requires_grad_(requires_grad)
    return out


def _run_gpu(kw, grad_img=None, debug=False):
    """Forward (+ backward when grad_img is given) through the drop-in package.  grad_img may be a callable
    (color, radii) -> dL/dcolor: it sees the forward's outputs before the backward runs (pixel masks)."""
    from diff_gaussian_rasterization import GaussianRasterizer
    rs = _settings(kw, debug)
    inp = _inputs(kw, grad_img is not None)
    P = inp["means3D"].shape[0]
    means2D = torch.zeros(P, 3, device=DEV, requires_grad=grad_img is not None)
    color, radii = GaussianRasterizer(rs)(means2D=means2D, **inp)
    grads = None
    if grad_img is not None:
        if callable(grad_img):
            grad_img = grad_img(color.detach().cpu().numpy(), radii.cpu().numpy())
        color.backward(torch.as_tensor(np.ascontiguousarray(grad_img), dtype=torch.float32).to(DEV))
        grads = {k: v.grad.detach().cpu().numpy() for k, v in inp.items()}
        grads["means2D"] = means2D.grad.detach().cpu().numpy()
    torch.cuda.synchronize()
    return color.detach().cpu().numpy(), radii.cpu().numpy(), grads


def _strict_pixels(fr64, radii, exact_radii=True):
    """Pixels on which the HIP path must take the oracle's decisions: not fragile, and (exact_radii=False) not in a
    tile that only one of the two binning rectangles of a Gaussian with a differently rounded radius covers."""
    strict = fr64.fragile_px == 0
    if exact_radii:
        np.testing.assert_array_equal(radii, fr64.radii)
    else:
        # at 1e6 Gaussians a handful sit on an integer boundary of ceil(3 sqrt(lambda)): binary32 vs binary64 may
        # round the radius differently by one; the tiles such a Gaussian touches are excluded from the strict check
        bad = np.nonzero(radii != fr64.radii)[0]
        assert bad.size <= 1e-5 * radii.size and np.all(np.abs(radii[bad] - fr64.radii[bad]) <= 1), bad.size
        strict = strict.copy()

        def rect(g, r):                      # A.5 for Gaussian g with radius r
            x, y = fr64.xy[g]
            lo = lambda v, G: int(min(G, max(0, int((v - r) / 16))))
            hi = lambda v, G: int(min(G, max(0, int((v + r + 15) / 16))))
            return lo(x, fr64.Gx), lo(y, fr64.Gy), hi(x, fr64.Gx), hi(y, fr64.Gy)
        for g in bad:                        # only tiles inside one rectangle and not the other can differ
            ra, rb = rect(g, radii[g]), rect(g, fr64.radii[g])
            if ra != rb:
                x0, y0, x1, y1 = min(ra[0], rb[0]), min(ra[1], rb[1]), max(ra[2], rb[2]), max(ra[3], rb[3])
                inner = (max(ra[0], rb[0]), max(ra[1], rb[1]), min(ra[2], rb[2]), min(ra[3], rb[3]))
                m = np.zeros_like(strict)
                m[y0 * 16:y1 * 16, x0 * 16:x1 * 16] = True
                m[inner[1] * 16:inner[3] * 16, inner[0] * 16:inner[2] * 16] = False
                strict &= ~m
    return strict


def _check_forward(kw, fr64, color, radii, exact_radii=True, fr32=None):
    """Pixels at 1e-5 on the strict (non-fragile) pixels.  `fr32` (the oracle's binary32 instantiation of the same frame) switches
    to the bound for DEEP lists: with hundreds of composited splats per pixel (cfg3n: 400) a binary32 blend — the reference's own
    arithmetic restated in binary32 included — sits more than 1e-5 from the exact result on a fraction of a percent of the pixels
    (every alpha inherits ~1e-4 of relative error from its splat's binary32 screen position at 1080p; measured: 0.41 % of cfg3n's
    pixels for the binary32 oracle, 5e-5 at most).  There the HIP image must be at least as close to the binary64 result as the
    binary32 oracle is: no strict pixel beyond 1e-4, and no more pixels beyond 1e-5 than the binary32 oracle has (x1.25 + 1e-4 N)."""
    strict = _strict_pixels(fr64, radii, exact_radii)
    err = np.abs(color.astype(np.float64) - fr64.color).max(0)
    assert strict.mean() > 0.97, f"too many fragile pixels: {1 - strict.mean():.4f}"
    if fr32 is None:
        assert err[strict].max() <= 1e-5, f"pixel error {err[strict].max():.3e} on non-fragile pixels"
    else:
        ref = np.abs(fr32.color.astype(np.float64) - fr64.color).max(0)
        n_hip, n_ref = int(((err > 1e-5) & strict).sum()), int(((ref > 1e-5) & strict).sum())
        print(f"deep lists: {n_hip} strict pixels beyond 1e-5 (binary32 oracle: {n_ref}) of {int(strict.sum())}; max {err[strict].max():.3e} "
              f"(binary32 oracle: {ref[strict].max():.3e}); mean n_contrib {fr64.n_contrib.mean():.0f}")
        assert err[strict].max() <= 1e-4, f"pixel error {err[strict].max():.3e} on non-fragile pixels"
        assert n_hip <= 1.25 * n_ref + 1e-4 * strict.sum(), (n_hip, n_ref)
    if (~strict).any():        # one splat more or less: <= alpha_min * |colour| (+ downstream T change)
        assert err[~strict].max() <= 2e-2, f"fragile-pixel error {err[~strict].max():.3e}"
    return err


GRAD_ATOL_REL = 1e-4        # x max|want| of the tensor
GRAD_RTOL = 1e-4            # x |want| of the element        (north_star: "1e-4 on grads")


def _check_grads(fr64, want, got, names, masked=False):
    """masked=True: dL/dcolor was zero on every fragile pixel for both sides -> EVERY Gaussian is strict.
    masked=False: Gaussians that reach a fragile pixel (fr64.fragile_g) may differ by one flipped splat.
    Returns (Gaussians with a non-zero gradient, how many of them were held to the strict bound)."""
    frag = np.zeros_like(fr64.fragile_g, dtype=bool) if masked else fr64.fragile_g.astype(bool)
    live = np.zeros(frag.shape[0], bool)
    for n in names:
        w = want[n].reshape(got[n].shape).astype(np.float64)
        g = got[n].astype(np.float64)
        live |= (np.abs(w).reshape(w.shape[0], -1).max(1) > 0) if w.size else False
        scale = max(np.abs(w).max(initial=0.0), 1e-30)
        err = np.abs(g - w)
        bound = GRAD_ATOL_REL * scale + GRAD_RTOL * np.abs(w)
        mask = ~frag.reshape((-1,) + (1,) * (w.ndim - 1)) & np.ones_like(w, bool)
        bad = (err > bound) & mask
        assert not bad.any(), (f"{n}: {bad.sum()} elements of {np.unique(np.nonzero(bad)[0]).size} Gaussians off; "
                               f"max err {err[mask].max():.3e} scale {scale:.3e} "
                               f"worst ratio {(err / bound)[mask].max():.2f}")
        if frag.any():
            loose = 2e-2 * scale + 5e-2 * np.abs(w)
            assert not ((err > loose) & ~mask).any(), f"{n}: fragile-Gaussian error {err[~mask].max():.3e}"
    return int(live.sum()), int((live & ~frag).sum())


GRAD_NAMES = ("means3D", "means2D", "opacities", "shs", "colors_precomp", "scales", "rotations", "cov3D_precomp")


def _forward_backward_strict(kw, fr64, gimg, exact_radii=True, label="", parallel=False, fr32=None):
    """The parity check proper: forward at 1e-5 on strict pixels, then the backward of a dL/dcolor that is ZERO on every
    non-strict pixel — identically for the HIP run and the oracle — with EVERY Gaussian held to the strict bound."""
    state = {}

    def masked(color, radii):
        state["strict"] = _strict_pixels(fr64, radii, exact_radii)
        return np.where(state["strict"][None], gimg, 0.0).astype(np.float32)
    color, radii, grads = _run_gpu(kw, masked)
    _check_forward(kw, fr64, color, radii, exact_radii, fr32=fr32)
    gm = np.where(state["strict"][None], gimg, 0.0).astype(np.float64)
    want = fr64.backward(gm, parallel=parallel)
    names = [n for n in GRAD_NAMES if n in grads]
    live, strict_live = _check_grads(fr64, want, grads, names, masked=True)
    assert strict_live == live
    print(f"{label}: {live} Gaussians with a non-zero gradient, {strict_live} held to {GRAD_ATOL_REL:g}*scale + "
          f"{GRAD_RTOL:g}*|w|; {int((~state['strict']).sum())} of {state['strict'].size} pixels masked")
    return color, radii, grads, want, live


FIXTURES = [
    dict(P=1, W=32, H=32, D=0, seed=101),
    dict(P=2, W=32, H=32, D=1, seed=102),
    dict(P=64, W=48, H=80, D=2, seed=103),
    dict(P=64, W=80, H=48, D=3, seed=104, bg=(1.0, 1.0, 1.0)),
    dict(P=2048, W=128, H=128, D=3, seed=105),
    dict(P=2048, W=128, H=128, D=0, seed=106, mode="color"),
    dict(P=2048, W=100, H=60, D=3, seed=107, mode="cov", bg=(0.2, 0.4, 0.6)),
    dict(P=2048, W=128, H=128, D=1, seed=108, mode="color+cov"),
    dict(P=5000, W=256, H=192, D=3, seed=109, scale_modifier=1.7),
]


def _fixture_kwargs(c):
    W, H = c["W"], c["H"]
    lo, hi = (0.01, 0.2) if c["P"] <= 64 else (0.005, 0.06)
    scene, cam = S.make_scene(c["P"], W, H, c["D"], c["seed"], scale_lo=lo, scale_hi=hi), S.make_camera(W, H)
    if c["P"] <= 2:     # make sure the tiny cases are actually on screen
        scene.means3D[:] = torch.tensor([[0.05, -0.03, 2.0], [-0.2, 0.1, 3.0]])[:c["P"]]
    a = scene.activated()
    extra = {}
    mode = c.get("mode", "")
    if "color" in mode:
        extra["colors_precomp"] = torch.rand(scene.P, 3, generator=torch.Generator().manual_seed(c["seed"]))
    if "cov" in mode:
        extra["cov3D_precomp"] = cov3d_from(a["scales"], a["rotations"], c.get("scale_modifier", 1.0))
    return raster_kwargs(scene, cam, bg=c.get("bg", (0, 0, 0)), scale_modifier=c.get("scale_modifier", 1.0), **extra)


@pytest.mark.parametrize("c", FIXTURES, ids=lambda c: f"P{c['P']}_{c['W']}x{c['H']}_D{c['D']}_{c.get('mode', 'sh')}")
def test_forward_backward_match_oracle(c):
    kw = _fixture_kwargs(c)
    fr64 = oracle.rasterize(dtype=np.float64, **kw)
    gimg = S.make_grad_image(c["W"], c["H"], c["seed"]).numpy()
    _forward_backward_strict(kw, fr64, gimg, label=f"fixture P={c['P']} {c['W']}x{c['H']}")
    # and the un-masked gradient image: Gaussians that reach a fragile pixel may differ by one flipped splat
    color, radii, grads = _run_gpu(kw, gimg)
    want = fr64.backward(gimg.astype(np.float64))
    _check_grads(fr64, want, grads, [n for n in GRAD_NAMES if n in grads])


def _per_tile_lists(v, plan, Tn):
    """Concatenate each tile's per-chunk ranges -> (list of Gaussian ids per tile, per-chunk lengths)."""
    sg = v["sorted_gaussian"].cpu().numpy().astype(np.uint32) if v["sorted_gaussian"] is not None else np.zeros(0, np.uint32)
    rng = v["ranges"].cpu().numpy().astype(np.int64)[:plan.chunks_run]
    lists, lens = [], np.zeros((plan.chunks_run, Tn), np.int64)
    for t in range(Tn):
        parts = []
        for c in range(plan.chunks_run):
            a, b = rng[c, t]
            parts.append(sg[a:b])
            lens[c, t] = b - a
        lists.append(np.concatenate(parts) if parts else np.zeros(0, np.uint32))
    return lists, lens


def _decode_n_contrib(v, lens, W, H):
    """last_enc -> 1-based position in the tile's concatenated list (the reference's n_contrib)."""
    enc = v["n_contrib"].cpu().numpy().astype(np.int64)
    c = (enc >> 26) - 1
    pos = enc & ((1 << 26) - 1)
    Gx = (W + 15) // 16
    ys, xs = np.mgrid[0:H, 0:W]
    tile = (ys // 16) * Gx + xs // 16
    before = np.concatenate([np.zeros((1, lens.shape[1]), np.int64), np.cumsum(lens, 0)], 0)   # [chunks+1, Tn]
    return np.where(c >= 0, before[np.maximum(c, 0), tile] + pos, 0)


def test_intermediates_bit_exact_vs_oracle_f32():
    """Integer work is bit-exact against the binary32 oracle: tiles touched, depth order, prefix sum, and the
    progressive per-tile lists = a prefix of the oracle's fully sorted list that covers every contributor."""
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization import _native as N
    c = dict(P=20000, W=320, H=200, D=3, seed=201)
    kw = _fixture_kwargs(c)
    fr = oracle.rasterize(dtype=np.float32, **kw)
    rs = _settings(kw)
    inp = _inputs(kw, False)
    color, radii, frame = dgr.rasterize_forward(inp["means3D"], inp["shs"], None, inp["opacities"], inp["scales"],
                                                inp["rotations"], None, rs)
    torch.cuda.synchronize()
    v = N.debug_views(frame.desc, frame.geom_ws, frame.binning_ws, frame.image_ws, frame.plan)
    np.testing.assert_array_equal(radii.cpu().numpy(), fr.radii)
    tiles_dev = v["tiles_touched"].cpu().numpy().astype(np.uint32)
    assert np.all(tiles_dev <= fr.tiles_touched)          # A.5 rectangle clipped to the alpha >= 1/255 box
    assert frame.R == int(tiles_dev.sum()) <= fr.num_rendered and frame.plan.num_visible == int((fr.radii > 0).sum())
    # depth order by SELECTION (csrc/gsr_select.hip): order[:V] holds every visible Gaussian once, chunk after chunk; chunks
    # partition the visible set by depth key (chunk c's keys lie in (key_end[c-1], key_end[c]]); the chunks that were BINNED are
    # in (binary32 depth, index) order — together exactly the prefix of the stable full sort — the others in index order.
    # Device depth is FMA-contracted, so swaps against the oracle are allowed only between depths a few ulp apart.
    rec = unscale_records(v["splat_records"].cpu().numpy())
    order = v["depth_order"].cpu().numpy().astype(np.int64)
    V = frame.plan.num_visible
    plan = frame.plan
    vis = np.nonzero(fr.radii > 0)[0]
    assert sorted(order[:V].tolist()) == vis.tolist()
    bnd = [int(plan.chunk_rank_begin[c]) for c in range(plan.num_chunks + 1)]
    assert bnd[0] == 0 and bnd[-1] == V and all(b > a for a, b in zip(bnd, bnd[1:]))
    keys_dev = rec[:, 9].view(np.uint32).astype(np.int64)
    for c in range(plan.num_chunks):
        k = keys_dev[order[bnd[c]:bnd[c + 1]]]
        lo = int(plan.chunk_key_end[c - 1]) if c else -1
        assert k.min() > lo and k.max() <= int(plan.chunk_key_end[c]), c
        if c >= plan.chunks_run:
            assert np.all(np.diff(order[bnd[c]:bnd[c + 1]]) > 0), c             # not needed: still in index order
    n_sorted = bnd[plan.chunks_run]
    want_order = vis[np.lexsort((vis, fr.depth[vis].view(np.uint32)))][:n_sorted]
    d_dev = rec[order[:n_sorted], 9]
    assert np.all(np.diff(d_dev.view(np.uint32).astype(np.int64)) >= 0)
    mism = np.nonzero(order[:n_sorted] != want_order)[0]
    if mism.size:
        assert mism.size < 1e-3 * V + 2
        dw = fr.depth[want_order[mism]]
        assert np.all(np.abs(rec[order[mism], 9] - dw) <= 4 * np.spacing(np.abs(dw)))
    offs = v["point_offsets"].cpu().numpy().astype(np.int64)
    for c in range(plan.chunks_run):                                               # the tile-count scan restarts at every chunk
        np.testing.assert_array_equal(offs[bnd[c]:bnd[c + 1]], np.cumsum(tiles_dev[order[bnd[c]:bnd[c + 1]]].astype(np.int64)))
        assert int(offs[bnd[c + 1] - 1]) == int(plan.chunk_instances_max[c])
    # per-tile lists
    Tn = fr.Gx * fr.Gy
    lists, lens = _per_tile_lists(v, frame.plan, Tn)
    # With exact tile culling the device list is a SUBSEQUENCE of the oracle's list (instances whose
    # alpha >= 1/255 ellipse cannot reach the tile are never emitted), in the same order.
    def is_subsequence(sub, full):
        j = 0
        for x in full:
            if j < sub.size and sub[j] == x:
                j += 1
        return j == sub.size
    n_bad, n_emitted, n_full = 0, 0, 0
    for t in range(Tn):
        want = fr.point_list[fr.ranges[t, 0]:fr.ranges[t, 1]]
        got = lists[t]
        n_emitted += got.size; n_full += want.size
        if not is_subsequence(got, want):
            n_bad += 1                      # only possible through depth-ulp reorderings
    assert n_bad <= max(2, Tn // 100), n_bad
    assert n_emitted < n_full
    vis = fr.radii > 0
    np.testing.assert_allclose(rec[vis, 0:2], fr.xy[vis], rtol=1e-6, atol=2e-4)
    np.testing.assert_allclose(rec[vis, 2:5], fr.conic_opacity[vis, :3], rtol=3e-5, atol=2e-6)   # B cancels to ~0
    # SH colours are evaluated lazily: only Gaussians that some tile took (of the depth chunks that were binned) have one
    n_binned = int(frame.plan.chunk_rank_begin[frame.plan.chunks_run])
    in_prefix = v["depth_order"].cpu().numpy().astype(np.int64)[:n_binned]
    binned = np.unique(np.concatenate([l for l in lists if l.size] + [np.empty(0, np.int64)]))
    assert binned.size > 0 and bool(vis[binned].all()) and np.isin(binned, in_prefix).all()
    np.testing.assert_allclose(rec[binned, 6:9], fr.rgb[binned], rtol=1e-5, atol=2e-6)
    bits = (fr.clamped[:, 0] | (fr.clamped[:, 1] << 1) | (fr.clamped[:, 2] << 2)).astype(np.uint8)
    np.testing.assert_array_equal(v["clamped"].cpu().numpy()[binned], bits[binned])
    # last contributor per pixel: same GAUSSIAN as the oracle's (positions differ because of the culling)
    nc = _decode_n_contrib(v, lens, fr.W, fr.H)
    strict = fr.fragile_px == 0
    ys, xs = np.mgrid[0:fr.H, 0:fr.W]
    tile = (ys // 16) * fr.Gx + xs // 16
    same = 0
    idx = np.nonzero(strict)
    for y, x in zip(*idx):
        t = tile[y, x]
        g_dev = lists[t][nc[y, x] - 1] if nc[y, x] > 0 else -1
        g_ora = fr.point_list[fr.ranges[t, 0] + fr.n_contrib[y, x] - 1] if fr.n_contrib[y, x] > 0 else -1
        same += int(g_dev == g_ora)
    assert same / idx[0].size > 0.9999
    np.testing.assert_allclose(np.abs(v["final_T"].cpu().numpy())[strict], fr.final_T[strict], rtol=1e-4, atol=1e-7)


def test_empty_and_degenerate_inputs():
    from diff_gaussian_rasterization import GaussianRasterizer
    kw = _fixture_kwargs(dict(P=64, W=48, H=80, D=2, seed=103, bg=(0.1, 0.2, 0.3)))
    rs = _settings(kw)
    # P = 0: image is the background, radii is empty
    z = lambda *s: torch.zeros(*s, device=DEV)
    color, radii = GaussianRasterizer(rs)(means3D=z(0, 3), means2D=z(0, 3), opacities=z(0, 1), shs=z(0, 9, 3),
                                          scales=z(0, 3), rotations=z(0, 4))
    assert radii.numel() == 0
    assert torch.allclose(color, torch.tensor([0.1, 0.2, 0.3], device=DEV)[:, None, None].expand(3, 80, 48))
    # everything behind the camera: R = 0
    inp = _inputs(kw, True)
    with torch.no_grad():
        inp["means3D"][:, 2] = -1.0
    means2D = z(64, 3).requires_grad_(True)
    color, radii = GaussianRasterizer(rs)(means2D=means2D, **inp)
    assert int((radii > 0).sum()) == 0
    color.sum().backward()
    for k, v in inp.items():
        assert torch.all(v.grad == 0), k
    assert torch.all(means2D.grad == 0)


def test_image_with_more_than_65536_tiles():
    """4352 x 4112 pixels = 272 x 257 = 69 904 tiles: tile ids need 17 bits, so the per-chunk tile sort takes
    3 radix passes instead of 2.  Forward + backward against the fp64 oracle, tolerances as everywhere (the scene's
    nearest splats are ~3000 px wide: the oracle's fragile-decision bands grow with the image size and with the
    conditioning of the 2D covariance, oracle/gsr_oracle_impl.h render_tile)."""
    W, H = 4352, 4112
    kw = _fixture_kwargs(dict(P=3000, W=W, H=H, D=1, seed=211))
    fr64 = oracle.rasterize(dtype=np.float64, parallel=True, **kw)
    assert fr64.Gx * fr64.Gy > 65536
    gimg = S.make_grad_image(W, H, 9).numpy()
    _forward_backward_strict(kw, fr64, gimg, label="69 904 tiles", parallel=True)


def test_api_contract():
    from diff_gaussian_rasterization import GaussianRasterizer
    kw = _fixture_kwargs(dict(P=2048, W=128, H=128, D=3, seed=105))
    rs = _settings(kw)
    inp = _inputs(kw, True)
    means2D = torch.zeros(2048, 3, device=DEV, requires_grad=True)
    r = GaussianRasterizer(rs)
    with pytest.raises(Exception, match="SHs or precomputed colors"):
        r(means3D=inp["means3D"], means2D=means2D, opacities=inp["opacities"], scales=inp["scales"], rotations=inp["rotations"])
    with pytest.raises(Exception, match="scale/rotation pair or precomputed 3D covariance"):
        r(means3D=inp["means3D"], means2D=means2D, opacities=inp["opacities"], shs=inp["shs"], scales=inp["scales"])
    # non-contiguous / detached inputs, 0-dim tensor sh_degree, no_grad
    rs2 = rs._replace(sh_degree=torch.tensor(3))
    shs_nc = inp["shs"].detach().transpose(1, 2).contiguous().transpose(1, 2)
    assert not shs_nc.is_contiguous()
    with torch.no_grad():
        c1, rad1 = GaussianRasterizer(rs2)(means3D=inp["means3D"], means2D=means2D, opacities=inp["opacities"],
                                           shs=shs_nc, scales=inp["scales"], rotations=inp["rotations"])
    c2, rad2 = r(means2D=means2D, **inp)
    assert torch.equal(c1, c2) and torch.equal(rad1, rad2)
    assert c2.dtype == torch.float32 and c2.shape == (3, 128, 128) and rad2.dtype == torch.int32
    # frozen inputs (scene/gaussian_model.py:104-125 detach()): only requested grads are produced
    frozen = dict(inp)
    frozen["means3D"] = inp["means3D"].detach()
    c3, _ = r(means2D=means2D, **frozen)
    c3.mean().backward()
    assert inp["scales"].grad is not None and means2D.grad is not None
    # markVisible == near-plane test
    vis = r.markVisible(inp["means3D"].detach())
    fr = oracle.rasterize(dtype=np.float32, **kw)
    pv_z = kw["means3D"] @ kw["viewmatrix"][:3, 2] + kw["viewmatrix"][3, 2]
    np.testing.assert_array_equal(vis.cpu().numpy(), pv_z > 0.2)
    assert np.all(vis.cpu().numpy()[fr.radii > 0])


def test_second_backward_follows_autograd_retain_graph_rules():
    """train.py:106 calls loss.backward() once; like the upstream extension's saved buffers the frame's workspaces live in
    autograd's saved-tensor slots: backward(retain_graph=True) may be repeated (same gradients, bit for bit), a second
    backward() without it raises autograd's own error, and a no_grad() render retains nothing and prepares no backward."""
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization import GaussianRasterizer
    kw = _fixture_kwargs(dict(P=5000, W=256, H=192, D=3, seed=109))
    rs, inp = _settings(kw), _inputs(kw, True)
    gimg = S.make_grad_image(256, 192, 9).to(DEV)
    means2D = torch.zeros(5000, 3, device=DEV, requires_grad=True)
    color, _ = GaussianRasterizer(rs)(means2D=means2D, **inp)
    color.backward(gimg, retain_graph=True)
    g1 = {k: v.grad.clone() for k, v in inp.items()}
    for v in inp.values():
        v.grad = None
    color.backward(gimg)                                  # second pass over the retained graph
    for k, v in inp.items():
        assert torch.equal(v.grad, g1[k]), k
    with pytest.raises(RuntimeError, match="backward through the graph a second time|already been freed"):
        color.backward(gimg)
    calls = []
    orig = dgr.N.backward_prepare
    dgr.N.backward_prepare = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    try:
        with torch.no_grad():
            c2, _ = GaussianRasterizer(rs)(means2D=means2D, **inp)
        assert calls == [] and c2.grad_fn is None and torch.equal(c2, color.detach())
        # ... and with grad mode on, a depth-complex frame (its first chunk holds < P/4 Gaussians) gets its backward outputs
        # zero-filled early, once
        scene, cam = S.make_scene(60_000, 320, 200, 1, 5, scale_lo=0.02, scale_hi=0.2), S.make_camera(320, 200)
        kw2 = raster_kwargs(scene, cam)
        rs2, inp2 = _settings(kw2), _inputs(kw2, True)
        m2 = torch.zeros(60_000, 3, device=DEV, requires_grad=True)
        c3, _ = GaussianRasterizer(rs2)(means2D=m2, **inp2)
        assert calls == [1], calls
        c3.sum().backward()
        assert calls == [1] and all(torch.isfinite(v.grad).all() for v in inp2.values())
    finally:
        dgr.N.backward_prepare = orig


def test_frozen_parameters_get_no_gradient_with_fused_activations():
    """The reference's freeze flags (train.py:58-59 --freeze_xyz etc., scene/gaussian_model.py:104-125: the getters detach())
    with pipe.fused_activations on: the raw-parameter path bypasses the getters, so a model with a freeze flag must take
    the plain path — the frozen parameter's .grad stays None, single-GPU and through the ShardedRenderer's render()."""
    from gaussian_params import Pipe
    from gaussian_renderer import render
    from scene import GaussianModel
    W, H = 160, 112
    cam = S.make_camera(W, H).to(DEV)
    bg = torch.zeros(3, device=DEV)
    pipe = Pipe()
    pipe.fused_activations = True
    for flag, name in (("freeze_means", "xyz"), ("freeze_scales", "scaling"), ("freeze_opacities", "opacity"), (None, None)):
        gm = GaussianModel(1)
        gm.adopt_scene(S.make_scene(3000, W, H, 1, 5, scale_lo=0.01, scale_hi=0.08), device=DEV)
        if flag:
            setattr(gm, flag, True)
        out = render(cam, gm, pipe, bg)
        out["render"].sum().backward()
        for k, p in gm._t.items():
            if k == name:
                assert p.grad is None, (flag, k)
            else:
                assert p.grad is not None and torch.isfinite(p.grad).all(), (flag, k)


def test_binning_workspace_is_capped_and_grows_on_demand():
    """The binning workspace (24 B per instance) is sized for the first depth chunk, not for the upper bound R of all of them.
    A frame whose lower half stays open (no splats there: those tiles never saturate) needs its later chunks: the first
    attempt stops with GSR_ERR_WORKSPACE before writing past the workspace, the frame is re-run with room for R, and image
    and gradients equal a render that had the full workspace from the start, bit for bit.  A saturating frame of the same
    size stays on the small workspace."""
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization import _native as N
    W, H, P = 640, 368, 150_000
    scene = S.make_scene(P, W, H, 1, 77, scale_lo=0.01, scale_hi=0.08)
    cam = S.make_camera(W, H)
    uneven = S.make_scene(P, W, H, 1, 77, scale_lo=0.01, scale_hi=0.08)
    uneven.means3D[:, 1] = -uneven.means3D[:, 1].abs() - 0.02 * uneven.means3D[:, 2]          # everything in the upper half
    gimg = S.make_grad_image(W, H, 3).to(DEV)
    for name, sc, expect_growth in (("saturating", scene, False), ("upper half only", uneven, True)):
        kw = raster_kwargs(sc, cam)
        rs, inp = _settings(kw), _inputs(kw, False)
        args = (inp["means3D"], inp["shs"], None, inp["opacities"], inp["scales"], inp["rotations"], None, rs)
        dgr._binning_guess.clear()
        errors = []
        orig = N.forward_render

        def spy(*a, **k):
            try:
                return orig(*a, **k)
            except N.GsrError as e:
                errors.append(e.status)
                raise
        N.forward_render = spy
        try:
            c1, r1, f1 = dgr.rasterize_forward(*args)
        finally:
            N.forward_render = orig
        s1 = dgr.rasterize_backward_screen(f1, gimg).clone()
        cap1 = int(f1.plan.binning_capacity)
        assert f1.plan.num_chunks > 1, (name, f1.plan.num_chunks)
        if expect_growth:
            assert errors == [N.ERR_WORKSPACE] and f1.plan.chunks_run > 1 and cap1 == f1.R, (name, errors, f1.plan.chunks_run, cap1, f1.R)
        else:
            assert errors == [] and f1.plan.chunks_run == 1 and cap1 < f1.R // 2, (name, errors, cap1, f1.R)
        # the same frame with the workspace sized for R from the start
        key = next(iter(dgr._binning_guess))
        dgr._binning_guess[key] = f1.R
        c2, r2, f2 = dgr.rasterize_forward(*args)
        s2 = dgr.rasterize_backward_screen(f2, gimg)
        assert int(f2.plan.binning_capacity) == f1.R
        assert torch.equal(c1, c2) and torch.equal(r1, r2) and torch.equal(s1, s2), name
    dgr._binning_guess.clear()


def test_backward_is_bitwise_deterministic():
    kw = _fixture_kwargs(dict(P=5000, W=256, H=192, D=3, seed=109))
    gimg = S.make_grad_image(256, 192, 9).numpy()
    _, _, g1 = _run_gpu(kw, gimg)
    _, _, g2 = _run_gpu(kw, gimg)
    for k in g1:
        assert np.array_equal(g1[k], g2[k]), k


def test_chunk_sort_paths_agree_and_equal_depths_keep_index_order():
    """The depth order is built per chunk (csrc/gsr_select.hip).  (a) The one-block LDS sort's two paths — bucket + rank sort,
    and the 4-bit radix passes it falls back to when keys crowd into one bucket — give bitwise the same frame and gradients.
    (b) Gaussians with bit-identical depth (4000 of them on one plane facing the camera: every one of them lands in the same
    bucket, so the fallback runs on its own) blend in index order, as the reference's stable sort has it: parity with the oracle
    at the strict bound."""
    c = dict(P=5000, W=256, H=192, D=3, seed=109)
    kw = _fixture_kwargs(c)
    gimg = S.make_grad_image(256, 192, 9).numpy()
    c1, r1, g1 = _run_gpu(kw, gimg)
    os.environ["GSR_SORT_FORCE_RADIX"] = "1"
    try:
        c2, r2, g2 = _run_gpu(kw, gimg)
    finally:
        del os.environ["GSR_SORT_FORCE_RADIX"]
    assert np.array_equal(c1, c2) and np.array_equal(r1, r2)
    for k in g1:
        assert np.array_equal(g1[k], g2[k]), k
    kw = _fixture_kwargs(dict(P=6000, W=256, H=192, D=1, seed=131))
    kw["means3D"] = kw["means3D"].copy()
    kw["means3D"][:4000, 2] = 1.5                              # S.make_camera looks down +z from the origin: view depth = z
    fr64 = oracle.rasterize(dtype=np.float64, **kw)
    assert np.unique(fr64.depth[:4000][fr64.radii[:4000] > 0].astype(np.float32)).size == 1
    *_, live = _forward_backward_strict(kw, fr64, gimg, label="equal depths")
    assert live > 100


def test_both_blend_forward_kernels_render_the_same_bits(tmp_path):
    """csrc/gsr_render.hip has two blend forward kernels (lock step; one 16-lane group per quadrant for chunks of small splats) and
    the backward recomputes every alpha: all three share one definition of lp with every rounding spelled out, so the kernels must
    agree bit for bit (slabs of one frame may pick different kernels).  GSR_FWD_GROUPS forces the choice; it is read when the library
    is loaded, hence the two child processes."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for forced in ("0", "1"):
        out = str(tmp_path / f"fwd{forced}.npy")
        env = dict(os.environ, GSR_FWD_GROUPS=forced)
        subprocess.run([sys.executable, os.path.join(root, "tools", "fwd_kernels_cmp.py"), "fix", out], check=True, env=env, timeout=300)
        outs.append(np.load(out))
    assert np.isfinite(outs[0]).all() and outs[0].std() > 0.01
    assert np.array_equal(outs[0], outs[1])


def test_slab_renders_tile_the_image_and_gradients_sum():
    """Tile-row slabs (multi-GPU sharding) on one device: slabs reproduce the full render
    bit-for-bit, and slab screen-space gradients sum to the full ones."""
    import diff_gaussian_rasterization as dgr
    kw = _fixture_kwargs(dict(P=5000, W=256, H=192, D=3, seed=109))
    rs = _settings(kw)
    inp = _inputs(kw, False)
    args = (inp["means3D"], inp["shs"], None, inp["opacities"], inp["scales"], inp["rotations"], None, rs)
    gimg = S.make_grad_image(256, 192, 9).to(DEV)
    full, radii, fr = dgr.rasterize_forward(*args)
    sfull = dgr.rasterize_backward_screen(fr, gimg)
    acc = torch.zeros_like(full)
    ssum = torch.zeros_like(sfull)
    from diff_gaussian_rasterization import _native as N
    from diff_gaussian_rasterization.sharded import NativeBackend
    keys_full = N.frame_arrays(fr.desc, fr.geom_ws)[0].clone()
    for rows in ((0, 5), (5, 6), (6, 12)):
        dgr.rasterize_forward(*args, tile_rows=rows, out_color=acc)
        _, r2, f2 = dgr.rasterize_forward(*args, tile_rows=rows)
        assert torch.equal(r2, radii)
        part = dgr.rasterize_backward_screen(f2, gimg)
        ssum += part
        # what the multi-GPU exchange relies on: the depth keys do not depend on the slab, and a slab's non-zero gradient
        # rows all belong to Gaussians with a key up to the end of the last chunk it binned (n of them)
        keys, key_end, n_prefix = NativeBackend().binned_prefix(f2)
        assert torch.equal(keys, keys_full)
        live = (part.abs().sum(1) > 0).nonzero().flatten()
        in_prefix = (keys >= 0) & (keys <= key_end)
        assert int(in_prefix.sum()) == n_prefix
        assert bool(in_prefix[live].all())
    assert torch.equal(acc, full)
    scale = sfull.abs().max()
    assert (ssum - sfull).abs().max() <= 2e-6 * scale


def test_cfg1_vs_oracle():
    """BASELINE.json configs[0]: 10k Gaussians, SH degree 0, 256x256.  BASELINE asks for the CPU rasterize forward only
    (bench.py times that as cpu_baseline.cfg1); here the HIP path runs the same frame, forward AND backward, against it."""
    scene, cam = S.make_config("cfg1")
    kw = raster_kwargs(scene, cam)
    fr64 = oracle.rasterize(dtype=np.float64, **kw)
    assert fr64.M == 1 and fr64.D == 0 and fr64.W == fr64.H == 256 and fr64.P == 10_000
    gimg = S.make_grad_image(256, 256, 1).numpy()
    *_, live = _forward_backward_strict(kw, fr64, gimg, label="cfg1")
    assert live > 100


def test_cfg2_full_size_vs_oracle():
    """BASELINE.json configs[1]: 100k Gaussians, SH degree 3, 800x800, forward + backward."""
    scene, cam = S.make_config("cfg2")
    kw = raster_kwargs(scene, cam)
    fr64 = oracle.rasterize(dtype=np.float64, parallel=True, **kw)
    gimg = S.make_grad_image(800, 800, 2).numpy()
    *_, live = _forward_backward_strict(kw, fr64, gimg, label="cfg2", parallel=True)
    assert live > 1000


@pytest.mark.parametrize("workload", ["cfg3", "cfg5", "cfg5n"])
def test_full_size_properties(workload):
    """BASELINE.json configs[2] (1M Gaussians, 1920x1080) and configs[4] (5M Gaussians, 3840x2160 — the maximum size,
    on ONE GPU) through size-independent properties: determinism, linearity of the backward in dL/dcolor, bg linearity
    of the forward, sortedness of the per-tile lists, and consistency of R with the per-Gaussian tile counts."""
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization import _native as N
    scene, cam = S.make_config(workload)
    Wd, Ht = S.CONFIGS[workload]["W"], S.CONFIGS[workload]["H"]
    kw = raster_kwargs(scene, cam, as_numpy=False)
    kwn = {k: (v.numpy() if isinstance(v, torch.Tensor) else v) for k, v in kw.items()}
    rs = _settings(kwn)
    inp = _inputs(kwn, False)
    args = (inp["means3D"], inp["shs"], None, inp["opacities"], inp["scales"], inp["rotations"], None, rs)
    c1, radii, fr = dgr.rasterize_forward(*args)
    c2, _, fr2 = dgr.rasterize_forward(*args)
    assert torch.equal(c1, c2)
    v = N.debug_views(fr.desc, fr.geom_ws, fr.binning_ws, fr.image_ws, fr.plan)
    assert int(v["tiles_touched"].long().sum()) == fr.R
    V = fr.plan.num_visible
    assert sum(int(fr.plan.chunk_instances_max[c]) for c in range(fr.plan.num_chunks)) == fr.R
    # the binned part of the depth order is sorted, and nothing behind it is nearer; every chunk's per-tile ranges are
    # disjoint, ordered, and hold depth-sorted splats
    rec = v["splat_records"]
    n_sorted = int(fr.plan.chunk_rank_begin[fr.plan.chunks_run])
    d = rec[:, 9][v["depth_order"].long()[:V]]
    assert bool((d[1:n_sorted] >= d[:n_sorted - 1]).all()) and (n_sorted == V or float(d[n_sorted:].min()) >= float(d[n_sorted - 1]))
    assert int(torch.unique(v["depth_order"][:V]).numel()) == V
    rng = v["ranges"].long()[:fr.plan.chunks_run]
    lens = rng[..., 1] - rng[..., 0]
    assert bool((lens >= 0).all())
    emitted = int(lens.sum())
    assert 0 < emitted <= fr.R
    sg = v["sorted_gaussian"].long()[:emitted]
    depth = rec[:, 9][sg]
    starts = torch.zeros(emitted, dtype=torch.bool, device=DEV)
    starts[rng[..., 0][lens > 0]] = True
    assert bool(((depth[1:] >= depth[:-1]) | starts[1:]).all())
    print(f"{workload}: R={fr.R} emitted={emitted} chunks_run={fr.plan.chunks_run}/{fr.plan.num_chunks}")
    # forward linear in bg: C(bg) = C(0) + T * bg
    rs_w = rs._replace(bg=torch.ones(3, device=DEV))
    cw, _, _ = dgr.rasterize_forward(*args[:-1], rs_w)
    assert (cw - (c1 + v["final_T"].abs()[None])).abs().max() <= 1e-6
    # backward linear in dL/dcolor and deterministic
    g1 = S.make_grad_image(Wd, Ht, 3).to(DEV)
    g2 = S.make_grad_image(Wd, Ht, 4).to(DEV)
    s1 = dgr.rasterize_backward_screen(fr, g1).clone()
    s1b = dgr.rasterize_backward_screen(fr, g1).clone()
    assert torch.equal(s1, s1b)
    s2 = dgr.rasterize_backward_screen(fr, g2).clone()
    s12 = dgr.rasterize_backward_screen(fr, g1 + 2 * g2).clone()
    scale = s12.abs().max()
    assert (s12 - (s1 + 2 * s2)).abs().max() <= 1e-4 * scale
    assert int((radii > 0).sum()) > scene.P // 2


@pytest.mark.parametrize("which", ["small", "cfg4"])
def test_sharded_renderer_native_backend_world1(which):
    """ShardedRenderer with the native HIP backend over RCCL ("nccl"), world_size 1 on the one GPU of this
    box: exercises slab render -> all-gather -> reduce-scatter -> sharded geometry backward -> all-gather
    end to end on device tensors; must equal the plain render() bit for bit (forward) / to rounding.
    which="cfg4": BASELINE.json configs[3] at full size (the cfg3 scene: 1e6 Gaussians, 1920x1080, SH 3)."""
    import os
    import socket

    import torch.distributed as dist
    from diff_gaussian_rasterization.sharded import ShardedRenderer
    from gaussian_params import GaussianParams, Pipe
    from gaussian_renderer import render
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        if which == "cfg4":
            scene, cam = S.make_config("cfg3")
        else:
            scene, cam = S.make_scene(20000, 320, 200, 3, 301, scale_lo=0.005, scale_hi=0.06), S.make_camera(320, 200)
        Wd, Ht = cam.image_width, cam.image_height
        cam = cam.to(DEV)
        bg = torch.tensor([0.1, 0.2, 0.3], device=DEV)
        gimg = S.make_grad_image(Wd, Ht, 1).to(DEV)
        res = []
        for mode in ("plain", "sharded"):
            model = GaussianParams(scene.to(DEV)).to(DEV)
            out = render(cam, model, Pipe(), bg) if mode == "plain" else ShardedRenderer(dist, 1, 0).render(cam, model, Pipe(), bg)
            out["render"].backward(gimg)
            res.append((out["render"].detach().clone(), out["radii"].clone(), out["viewspace_points"].grad.clone(),
                        [p.grad.clone() for p in model.parameters()]))
        assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
        assert torch.equal(res[0][2], res[1][2])
        for a, b in zip(res[0][3], res[1][3]):
            assert torch.equal(a, b)
        # the bench's N > 1 step: render + slab-local loss + backward over RCCL (device-tensor collectives incl. the
        # deferred MAX and the 8-byte sum of the loss terms) against the single-GPU step
        import loss_utils
        gt = torch.rand(3, Ht, Wd, generator=torch.Generator().manual_seed(2)).to(DEV)
        steps = []
        for mode in ("plain", "sharded"):
            model = GaussianParams(scene.to(DEV)).to(DEV)
            if mode == "plain":
                loss = loss_utils.training_loss(render(cam, model, Pipe(), bg)["render"], gt)
            else:
                sr = ShardedRenderer(dist, 1, 0)
                loss = sr.training_loss(sr.render(cam, model, Pipe(), bg)["render"], gt)
            loss.backward()
            steps.append((float(loss.detach()), [p.grad.clone() for p in model.parameters()]))
        assert abs(steps[0][0] - steps[1][0]) < 2e-6
        for a, b in zip(steps[0][1], steps[1][1]):
            assert torch.allclose(a, b, rtol=1e-5, atol=1e-7 * float(a.abs().max()))
    finally:
        dist.destroy_process_group()


def test_fused_loss_matches_reference_golden():
    """csrc/gsr_loss.hip against the reference's own l1_loss/ssim values and autograd gradient
    (tests/golden/loss.npz, generated from utils/loss_utils.py), and against the torch mirror at 1080p."""
    import os
    import loss_utils
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "loss.npz"))
    for i in range(2):
        a = torch.tensor(gold[f"a{i}"], device=DEV, requires_grad=True)
        b = torch.tensor(gold[f"b{i}"], device=DEV)
        loss = loss_utils.training_loss(a, b)
        loss.backward()
        assert abs(loss.item() - float(gold[f"loss_{i}"])) < 2e-6
        g = a.grad.cpu().numpy()
        np.testing.assert_allclose(g, gold[f"grad_a{i}"], atol=2e-9, rtol=2e-4)
    gen = torch.Generator().manual_seed(3)
    a = torch.rand(3, 1080, 1920, generator=gen).to(DEV).requires_grad_(True)
    b = torch.rand(3, 1080, 1920, generator=gen).to(DEV)
    (loss_utils.training_loss(a, b) * 3.0).backward()
    a2 = a.detach().clone().requires_grad_(True)
    (loss_utils.training_loss_torch(a2, b) * 3.0).backward()
    assert (a.grad - a2.grad).abs().max() <= 2e-4 * a2.grad.abs().max()
    # widths that are / are not multiples of 4 (float4 and scalar staging), sizes that are not multiples of the tile
    for (Hh, Ww) in ((70, 100), (67, 101), (33, 34), (5, 7)):
        a = torch.rand(3, Hh, Ww, generator=gen).to(DEV).requires_grad_(True)
        b = torch.rand(3, Hh, Ww, generator=gen).to(DEV)
        la = loss_utils.training_loss(a, b)
        la.backward()
        a2 = a.detach().clone().requires_grad_(True)
        lb = loss_utils.training_loss_torch(a2, b)
        lb.backward()
        assert abs(float(la.detach()) - float(lb.detach())) < 5e-6, (Hh, Ww)
        assert (a.grad - a2.grad).abs().max() <= 2e-4 * a2.grad.abs().max(), (Hh, Ww)


def test_drop_in_l1_loss_and_ssim_match_reference_golden():
    """loss_utils.l1_loss / loss_utils.ssim with the reference's signatures (utils/loss_utils.py:17-18, 33-63) as autograd ops
    over the fused kernels, each on its own against the reference's values and autograd gradients (tests/golden/loss.npz:
    l1_*, ssim_*, grad_l1_a*, grad_ssim_a*), then composed exactly as train.py:104-105 writes the loss."""
    import loss_utils
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "loss.npz"))
    for i in range(2):
        b = torch.tensor(gold[f"b{i}"], device=DEV)
        a = torch.tensor(gold[f"a{i}"], device=DEV, requires_grad=True)
        l1 = loss_utils.l1_loss(a, b)
        assert isinstance(l1.grad_fn, loss_utils._L1._backward_cls) or "L1" in type(l1.grad_fn).__name__
        assert abs(float(l1) - float(gold[f"l1_{i}"])) < 1e-6
        l1.backward()
        np.testing.assert_allclose(a.grad.cpu().numpy(), gold[f"grad_l1_a{i}"], atol=1e-9, rtol=1e-5)
        a = torch.tensor(gold[f"a{i}"], device=DEV, requires_grad=True)
        s = loss_utils.ssim(a, b)
        assert abs(float(s) - float(gold[f"ssim_{i}"])) < 2e-6
        (s * 1.0).backward()
        np.testing.assert_allclose(a.grad.cpu().numpy(), gold[f"grad_ssim_a{i}"], atol=3e-9, rtol=3e-4)
        # train.py:104-105 verbatim; the two calls share one fused forward
        a = torch.tensor(gold[f"a{i}"], device=DEV, requires_grad=True)
        Ll1 = loss_utils.l1_loss(a, b)
        assert loss_utils._shared is not None and loss_utils._shared[0] is a
        loss = (1.0 - 0.2) * Ll1 + 0.2 * (1.0 - loss_utils.ssim(a, b))
        assert loss_utils._shared is None
        loss.backward()
        assert abs(float(loss) - float(gold[f"loss_{i}"])) < 2e-6
        np.testing.assert_allclose(a.grad.cpu().numpy(), gold[f"grad_a{i}"], atol=2e-9, rtol=2e-4)
        # an in-place change of the image between the two calls must not be served from the shared forward
        a2 = torch.tensor(gold[f"a{i}"], device=DEV)
        v1 = float(loss_utils.l1_loss(a2, b))
        a2.mul_(0.5)
        s2 = float(loss_utils.ssim(a2, b))
        assert abs(s2 - float(loss_utils.ssim_torch(a2, b))) < 2e-6 and abs(v1 - float(gold[f"l1_{i}"])) < 1e-6
    # other signatures fall back to the torch form: window sizes, per-image means, batches, CPU tensors
    a, b = torch.rand(2, 3, 20, 24, device=DEV), torch.rand(2, 3, 20, 24, device=DEV)
    assert loss_utils.ssim(a, b, size_average=False).shape == (2,)
    assert abs(float(loss_utils.ssim(a[0], b[0], window_size=7)) - float(loss_utils.ssim_torch(a[0], b[0], 7))) < 1e-7
    assert abs(float(loss_utils.l1_loss(a.cpu(), b.cpu())) - float((a - b).abs().mean())) < 1e-6


def test_cfg3_full_size_vs_oracle():
    """BASELINE.json configs[2] scene at full size (1e6 Gaussians, 1920x1080, SH 3): forward and backward
    against the fp64 oracle.  This frame exercises what the progressive pipeline adds: 1 of the planned depth
    chunks runs, 95 % of the reference's 43.8 M instances are never binned (closed tiles + tile culling)."""
    scene, cam = S.make_config("cfg3")
    kw = raster_kwargs(scene, cam)
    fr64 = oracle.rasterize(dtype=np.float64, parallel=True, **kw)
    gimg = S.make_grad_image(1920, 1080, 3).numpy()
    *_, live = _forward_backward_strict(kw, fr64, gimg, exact_radii=False, label="cfg3", parallel=True)
    assert live > 1000


def test_cfg3n_full_size_vs_oracle():
    """The NON-saturating workload of bench.py's `secondary` line (scene_synth.CONFIGS["cfg3n"]: R/P = 3.55, ~400 composited
    splats per pixel, 98 % of the visible Gaussians carry a gradient) at full size through the standard API: the variant-B
    emit + radix path at 2.4 M instances, k_chunk_colors_all and the dense geometry backward against the fp64 oracle."""
    scene, cam = S.make_config("cfg3n")
    kw = raster_kwargs(scene, cam)
    fr64 = oracle.rasterize(dtype=np.float64, parallel=True, **kw)
    fr32 = oracle.rasterize(dtype=np.float32, parallel=True, **kw)
    gimg = S.make_grad_image(1920, 1080, 3).numpy()
    *_, live = _forward_backward_strict(kw, fr64, gimg, exact_radii=False, label="cfg3n", parallel=True, fr32=fr32)
    assert live > 800_000


def test_debug_flag_synchronises_and_matches():
    """`debug=True` (README.md:147-150 semantics: check after every kernel) gives the same result."""
    kw = _fixture_kwargs(dict(P=2048, W=128, H=128, D=3, seed=105))
    gimg = S.make_grad_image(128, 128, 1).numpy()
    c0, r0, g0 = _run_gpu(kw, gimg, debug=False)
    c1, r1, g1 = _run_gpu(kw, gimg, debug=True)
    assert np.array_equal(c0, c1) and np.array_equal(r0, r1)
    for k in g0:
        assert np.array_equal(g0[k], g1[k]), k


@pytest.mark.parametrize("n,end_bit,dev_count", [(0, 8, False), (1, 32, False), (1000, 13, True), (70_000, 32, False),
                                                 (1_300_000, 15, True), (5_000_000, 13, True), (9_000_001, 32, False)])
def test_radix_sort_is_stable_and_exact(n, end_bit, dev_count):
    """csrc/gsr_sort.hip against torch.sort(stable=True): every size class (small scatter, LDS-reordering
    scatter >= 4 M), host- and device-side counts, narrow (tile id) and full-width (depth) keys."""
    from diff_gaussian_rasterization import _native as N
    g = torch.Generator(device="cpu").manual_seed(n + end_bit)
    hi = (1 << end_bit) - 1 if end_bit < 31 else (1 << 31) - 1
    keys = torch.randint(0, hi + 1, (n,), generator=g, dtype=torch.int64).to(torch.int32).to(DEV)
    if n > 10:                        # long runs of equal keys: stability matters
        keys[: n // 3] = keys[0]
    vals = torch.arange(n, dtype=torch.int32, device=DEV)
    ks, vs = N.debug_sort_pairs(keys, vals, end_bit, dev_count)
    torch.cuda.synchronize()
    want_k, order = torch.sort(keys.long(), stable=True)
    assert torch.equal(ks.long(), want_k)
    assert torch.equal(vs.long(), order)


@pytest.mark.parametrize("P", [1, 2, 3, 5, 777, 20_000, 300_000])
def test_dist2_knn3_matches_brute_force(P):
    """csrc/gsr_knn.hip (the reference's simple_knn.distCUDA2) against the oracle's brute force
    (up to 20k points) and against a chunked torch.cdist top-k beyond that."""
    from simple_knn._C import distCUDA2
    g = torch.Generator().manual_seed(P)
    pts = torch.randn(P, 3, generator=g) * torch.tensor([1.0, 0.3, 2.0])
    if P >= 777:
        pts[:50] = pts[50:100]                  # exact duplicates -> zero distances
    got = distCUDA2(pts.to(DEV)).cpu().numpy()
    if P <= 20_000:
        want = oracle.dist2_knn3(pts.numpy())
    else:
        want = np.zeros(P, np.float32)
        pd = pts.double().to(DEV)
        for i in range(0, P, 4096):
            d = torch.cdist(pd[i:i + 4096], pd).pow(2)
            d[torch.arange(d.shape[0]), torch.arange(i, i + d.shape[0])] = float("inf")
            want[i:i + 4096] = d.topk(3, largest=False).values.sum(1).div(3).float().cpu().numpy()
    np.testing.assert_allclose(got, want, rtol=2e-5, atol=1e-9)


def test_training_loop_densifies_and_learns():
    """f1: the reference's training iteration (train.py:63-146) around the HIP rasterizer on a synthetic scene:
    the loss goes down, densification / pruning / opacity reset run, PLY export round-trips on device."""
    import tempfile
    from dataclasses import replace

    from gaussian_params import Pipe
    from gaussian_renderer import render
    from scene import GaussianModel, OptimizationDefaults
    from train_loop import train
    torch.manual_seed(0)
    W, H = 160, 112
    cams = [c.to(DEV) for c in S.arc_cameras(W, H, 4)]
    truth = GaussianModel(2); truth.adopt_scene(S.make_scene(4000, W, H, 2, 30, scale_lo=0.01, scale_hi=0.08), device=DEV)
    bg = torch.zeros(3, device=DEV)
    with torch.no_grad():
        targets = [render(c, truth, Pipe(), bg)["render"].clone() for c in cams]

    class Pcd:                                  # BasicPointCloud-shaped init: noisy subset of the true cloud
        points = (truth._xyz.detach()[::4] + 0.01 * torch.randn(1000, 3, device=DEV)).cpu().numpy()
        colors = np.full((1000, 3), 0.5, np.float32)
    gm = GaussianModel(2)
    gm.create_from_pcd(Pcd, spatial_lr_scale=1.0, device=DEV)
    assert gm._scaling.shape == (1000, 3) and torch.isfinite(gm._scaling).all()
    opt = replace(OptimizationDefaults(), densify_from_iter=20, densification_interval=20, densify_until_iter=150)
    gm.training_setup(opt)
    losses, counts = [], []
    train(gm, cams, targets, opt, Pipe(), bg, iterations=160, scene_extent=3.0,
          on_iteration=lambda it, loss, g: (losses.append(float(loss.detach())), counts.append(g._xyz.shape[0])))
    assert np.mean(losses[-10:]) < 0.8 * np.mean(losses[:10]), (losses[:3], losses[-3:])
    assert max(counts) != 1000 and len(set(counts)) > 2          # the point count changed: densify + prune ran
    assert gm.active_sh_degree == 0 and all(np.isfinite(losses))
    gm.reset_opacity()
    assert float(gm.get_opacity.detach().max()) <= 0.01 + 1e-6
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "point_cloud.ply")
        gm.save_ply(path)
        gm2 = GaussianModel(2); gm2.load_ply(path, device=DEV)
        assert torch.equal(gm2._xyz.detach(), gm._xyz.detach()) and torch.equal(gm2._features_rest.detach(), gm._features_rest.detach())


def test_fused_adam_matches_torch_adam():
    """csrc/gsr_optim.hip against torch.optim.Adam (the reference's optimizer, eps = 1e-15) over several steps,
    odd sizes included; state_dict layout is interchangeable."""
    from fused_adam import FusedAdam
    torch.manual_seed(3)
    shapes = [(1000, 3), (1001, 15, 3), (7,), (1, 1)]
    ref_p = [torch.randn(*s, device=DEV).requires_grad_(True) for s in shapes]
    fus_p = [p.detach().clone().requires_grad_(True) for p in ref_p]
    mk = lambda ps: [{"params": [p], "lr": 0.01 * (i + 1), "name": str(i)} for i, p in enumerate(ps)]
    ref, fus = torch.optim.Adam(mk(ref_p), lr=0.0, eps=1e-15), FusedAdam(mk(fus_p), lr=0.0, eps=1e-15)
    for it in range(6):
        for a, b in zip(ref_p, fus_p):
            g = torch.randn_like(a) * (0.0 if it == 3 else 1.0)       # a zero-gradient step too
            a.grad, b.grad = g.clone(), g.clone()
        ref.step(); fus.step()
    for a, b in zip(ref_p, fus_p):
        assert (a - b).detach().abs().max() <= 2e-6 * max(1.0, float(a.detach().abs().max()))
        assert (ref.state[a]["exp_avg_sq"] - fus.state[b]["exp_avg_sq"]).abs().max() <= 1e-6
    ref.load_state_dict(fus.state_dict())                              # same layout


def test_integration_md_binding_example_runs():
    """INTEGRATION.md section 2 shows the ctypes binding a maintainer of the upstream extension would write against
    libgsrast.so.  The code block is executed as written (only the library path is substituted) and must give
    bitwise the package's own result: the document cannot drift from the ABI."""
    import re
    from diff_gaussian_rasterization import GaussianRasterizer, _native
    text = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "INTEGRATION.md")).read()
    code = re.search(r"```python\n(.*?)```", text, re.S).group(1).replace('"libgsrast.so"', repr(_native.lib_path()))
    ns = {}
    exec(compile(code, "INTEGRATION.md", "exec"), ns)
    kw = _fixture_kwargs(dict(P=5000, W=200, H=136, D=2, seed=77, bg=(0.2, 0.1, 0.3)))
    rs, inp = _settings(kw), _inputs(kw, True)
    P, M = inp["means3D"].shape[0], inp["shs"].shape[1]
    gimg = S.make_grad_image(200, 136, 4).to(DEV)
    means2D = torch.zeros(P, 3, device=DEV, requires_grad=True)
    color, radii = GaussianRasterizer(rs)(means2D=means2D, **inp)
    color.backward(gimg)
    e = torch.empty(0, device=DEV)
    d = {k: v.detach() for k, v in inp.items()}
    R, color2, radii2, saved = ns["rasterize_gaussians"](
        rs.bg, d["means3D"], e, d["opacities"], d["scales"], d["rotations"], rs.scale_modifier, e, rs.viewmatrix,
        rs.projmatrix, rs.tanfovx, rs.tanfovy, rs.image_height, rs.image_width, d["shs"], rs.sh_degree, rs.campos, False, False)
    assert R > 0 and torch.equal(color2, color.detach()) and torch.equal(radii2, radii)
    out = ns["rasterize_gaussians_backward"](saved, gimg, P, M)
    torch.cuda.synchronize()
    for k in ("means3D", "shs", "opacities", "scales", "rotations"):
        assert torch.equal(out[k], inp[k].grad), k
    assert torch.equal(out["means2D"], means2D.grad)


def test_more_than_2_pow_32_tile_instances_is_an_error_not_a_wrap():
    """70 000 screen-filling Gaussians over 65 536 tiles: the sum of tiles touched (4.6e9) does not fit the 32-bit
    instance index.  The frame is refused with a message (the sum is also formed in 64 bits on the device);
    rendering it as two tile-row slabs, as the message says, works."""
    from diff_gaussian_rasterization import GaussianRasterizer, _native
    from diff_gaussian_rasterization import rasterize_forward
    P, W, H = 70_000, 4096, 4096
    cam = S.make_camera(W, H)
    g = torch.Generator().manual_seed(5)
    means = torch.cat([(torch.rand(P, 2, generator=g) - 0.5) * 0.1, torch.full((P, 1), 1.0)], 1)
    scene = S.make_scene(P, W, H, 0, 5)
    kw = raster_kwargs(scene, cam)
    kw["means3D"] = means.numpy()
    kw["scales"] = np.full((P, 3), 50.0, np.float32)
    kw["opacities"] = np.full((P, 1), 0.9, np.float32)
    rs, inp = _settings(kw), _inputs(kw, False)
    with pytest.raises(_native.GsrError, match="exceed 2\\^32"):
        GaussianRasterizer(rs)(means2D=torch.zeros(P, 3, device=DEV), **inp)
    # a slab of 16 of the 256 tile rows fits (2.9e8 instances of upper bound).  All 70 000 depths are bit-identical here, and
    # Gaussians with the same depth key always share a chunk (chunks are selected by key): this frame is ONE chunk, binned whole
    color, radii, fr = rasterize_forward(inp["means3D"], inp["shs"], None, inp["opacities"], inp["scales"], inp["rotations"],
                                         None, rs, tile_rows=(0, 16))
    torch.cuda.synchronize()
    assert fr.R == P * 16 * 256 and int((radii > 0).sum()) == P
    assert fr.plan.num_chunks == 1 and fr.plan.chunks_run == 1
    assert torch.isfinite(color).all() and float(color[:, :256].abs().sum()) > 0 and float(color[:, 256:].abs().sum()) == 0


@pytest.mark.parametrize("D,max_D", [(3, 3), (1, 3), (0, 0)])
def test_raw_parameter_mode_equals_getters_plus_standard_api(D, max_D):
    """SURVEY 8a row a14 fused (`pipe.fused_activations`, GaussianRasterizer.forward_raw): rendering from the raw
    parameters with the activations inside the kernels gives the image of render() through the getters and the
    same gradients on the raw parameters as torch autograd through exp / sigmoid / normalize / cat."""
    from gaussian_params import GaussianParams, Pipe
    from gaussian_renderer import render
    W, H = 320, 208
    scene = S.make_scene(20_000, W, H, max_D, 17 + D, scale_lo=0.005, scale_hi=0.06).to(DEV)
    cam = S.make_camera(W, H).to(DEV)
    bg = torch.tensor([0.1, 0.0, 0.2], device=DEV)
    gimg = S.make_grad_image(W, H, 6).to(DEV)
    results = []
    for fused in (False, True):
        model = GaussianParams(scene, max_sh_degree=max_D).to(DEV)
        model.active_sh_degree = D
        pipe = Pipe()
        pipe.fused_activations = fused
        out = render(cam, model, pipe, bg)
        out["render"].backward(gimg)
        torch.cuda.synchronize()
        grads = {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}
        results.append((out["render"].detach(), out["radii"], out["viewspace_points"].grad.detach().clone(), grads))
    (img_a, rad_a, vp_a, g_a), (img_b, rad_b, vp_b, g_b) = results
    assert int((rad_a != rad_b).sum()) <= 2                     # exp() may differ in the last bit: ceil() boundary
    # exp / sigmoid evaluated in the kernel may differ from torch's in the last bit, which can flip a discrete blend
    # decision (alpha < 1/255, T < 1e-4) on a pixel: all but a handful of pixels agree to 2e-5, those to one splat
    derr = (img_a - img_b).abs().amax(0)
    assert float((derr > 2e-5).float().mean()) <= 1e-4 and float(derr.max()) <= 8e-3
    names = ["_xyz", "_features_dc", "_opacity", "_scaling", "_rotation"] + (["_features_rest"] if max_D > 0 else [])
    assert set(names) <= set(g_a) and set(names) <= set(g_b)
    for n in names:
        a, b = g_a[n].reshape(g_a[n].shape[0], -1), g_b[n].reshape(g_b[n].shape[0], -1)
        assert a.shape == b.shape, n
        scale = float(a.abs().max())
        assert scale > 0, n
        err = (a - b).abs()
        bad_rows = (err > 2e-5 * scale + 2e-3 * a.abs()).any(1)
        assert float(bad_rows.float().mean()) <= 2e-4, f"{n}: {int(bad_rows.sum())} rows differ"
        assert float((a - b).norm() / a.norm()) <= 1e-4, n
    assert float((vp_a - vp_b).norm() / vp_a.norm()) <= 1e-4
    if D < max_D:                                               # coefficients above the active degree get exact zeros
        K = (D + 1) ** 2
        assert float(g_b["_features_rest"][:, K - 1:].abs().max()) == 0.0


def test_native_densification_stats_equal_masked_torch_ops():
    """GaussianModel.update_densification_stats (gsr_densify_stats) against the reference's masked updates
    (train.py:127-130, scene/gaussian_model.py:415-417) applied twice with different visible sets."""
    from scene import GaussianModel
    P = 50_000
    gm = GaussianModel(0)
    gm.adopt_scene(S.make_scene(P, 64, 64, 0, 3), device=DEV)
    gm.training_setup(__import__("scene").OptimizationDefaults())
    ref = dict(mr=torch.zeros(P, device=DEV), acc=torch.zeros(P, 1, device=DEV), den=torch.zeros(P, 1, device=DEV))
    g = torch.Generator().manual_seed(1)
    for _ in range(2):
        radii = (torch.randint(-3, 40, (P,), generator=g).clamp_min(0)).to(torch.int32).to(DEV)
        vsp = torch.zeros(P, 3, device=DEV, requires_grad=True)
        vsp.grad = torch.randn(P, 3, generator=g).to(DEV)
        gm.update_densification_stats(vsp, radii)
        vis = radii > 0
        ref["mr"][vis] = torch.max(ref["mr"][vis], radii[vis])
        ref["acc"][vis] += torch.norm(vsp.grad[vis, :2], dim=-1, keepdim=True)
        ref["den"][vis] += 1
    torch.cuda.synchronize()
    assert torch.equal(gm.max_radii2D, ref["mr"]) and torch.equal(gm.denom, ref["den"])
    assert torch.allclose(gm.xyz_gradient_accum, ref["acc"], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("scale_lo,scale_hi,label", [(0.5, 2.0, "teams of 16 waves"), (0.08, 0.3, "teams of 4 waves")])
def test_chunks_of_few_large_splats(scale_lo, scale_hi, label):
    """Count/emit variant A with wide teams (mean rectangle >= 1024 tiles -> 16 waves per Gaussian, >= 96 -> 4): a few
    hundred screen-filling splats at 1280x720 (3600 tiles).  Forward, intermediates and backward against the oracle."""
    W, H = 1280, 720
    scene = S.make_scene(300, W, H, 1, 301, scale_lo=scale_lo, scale_hi=scale_hi)
    kw = raster_kwargs(scene, S.make_camera(W, H))
    fr64 = oracle.rasterize(dtype=np.float64, fragile_eps=2e-5, parallel=True, **kw)
    vis = fr64.radii > 0
    mean_rect = fr64.tiles_touched[vis].mean()
    assert mean_rect >= (1024 if "16" in label else 96), mean_rect
    gimg = S.make_grad_image(W, H, 12).numpy()
    _forward_backward_strict(kw, fr64, gimg, label=label, parallel=True)


def test_prefiltered_flag():
    """`prefiltered=True` promises that every Gaussian passes the frustum test (A.1).  With the promise kept the frame
    is the same as without the flag; with it broken the upstream kernel traps, here the call fails with
    GSR_ERR_PREFILTERED and a message."""
    from diff_gaussian_rasterization import GaussianRasterizer, _native
    kw = _fixture_kwargs(dict(P=500, W=96, H=64, D=1, seed=9))
    kw["means3D"] = kw["means3D"].copy()
    kw["means3D"][:, 2] = np.abs(kw["means3D"][:, 2]) + 0.5           # all in front of the near cut
    rs, inp = _settings(kw), _inputs(kw, False)
    z = torch.zeros(500, 3, device=DEV)
    ref, _ = GaussianRasterizer(rs)(means2D=z, **inp)
    got, _ = GaussianRasterizer(rs._replace(prefiltered=True))(means2D=z, **inp)
    assert torch.equal(ref, got)
    inp["means3D"][7, 2] = 0.1                                           # behind the 0.2 cut
    with pytest.raises(_native.GsrError, match="prefiltered"):
        GaussianRasterizer(rs._replace(prefiltered=True))(means2D=z, **inp)
    GaussianRasterizer(rs)(means2D=z, **inp)                             # without the promise: culled silently


def test_randomised_configurations_against_oracle():
    """Seeded sweep over shapes the fixed fixtures do not hit: image sizes off the 16-px tile grid (down to 1x1 and 17x1),
    Gaussian counts around the 64-rank wave boundaries of the binning kernels, splat sizes from sub-pixel to
    screen-filling (all three count/emit variants), every SH degree, non-square pixels, tile-row slabs."""
    from diff_gaussian_rasterization import rasterize_forward, rasterize_backward_screen, rasterize_backward_geom
    rng = np.random.default_rng(20241004)
    sizes = [(1, 1), (17, 1), (16, 16), (33, 47), (250, 130), (640, 360), (96, 300)]
    counts = [1, 63, 64, 65, 129, 1000, 4097]
    scales = [(0.0005, 0.003), (0.005, 0.05), (0.05, 0.6), (0.3, 3.0)]
    n_live = 0
    for it in range(42):
        W, H = sizes[it % len(sizes)]
        P = counts[int(rng.integers(len(counts)))]
        lo, hi = scales[int(rng.integers(len(scales)))]
        D = int(rng.integers(0, 4))
        cam = S.make_camera(W, H, tanfovy=0.5, tanfovx=float(0.5 * W / H * rng.uniform(0.7, 1.4)))
        scene = S.make_scene(P, W, H, D, 1000 + it, scale_lo=lo, scale_hi=hi)
        kw = raster_kwargs(scene, cam, bg=tuple(rng.uniform(0, 1, 3).round(2)), scale_modifier=float(rng.choice([1.0, 0.6, 1.5])))
        Gy = (H + 15) // 16
        rows = None
        if Gy > 2 and it % 3 == 2:
            a = int(rng.integers(0, Gy - 1))
            rows = (a, int(rng.integers(a + 1, Gy + 1)))
        fr64 = oracle.rasterize(dtype=np.float64, fragile_eps=2e-5, tile_rows=rows, **kw)
        gimg = S.make_grad_image(W, H, it).numpy()
        rs, inp = _settings(kw), _inputs(kw, False)
        color, radii, fr = rasterize_forward(inp["means3D"], inp["shs"], None, inp["opacities"], inp["scales"], inp["rotations"], None,
                                             rs, tile_rows=rows)
        strict = fr64.fragile_px == 0
        gimg = np.where(strict[None], gimg, 0.0).astype(np.float32)   # no gradient through decisions that may flip
        screen = rasterize_backward_screen(fr, torch.as_tensor(gimg).to(DEV))
        g = rasterize_backward_geom(fr, screen, (True,) * 8)
        torch.cuda.synchronize()
        tag = f"case {it}: {W}x{H} P={P} scales=({lo},{hi}) D={D} rows={rows}"
        np.testing.assert_array_equal(radii.cpu().numpy(), fr64.radii, err_msg=tag)
        got = color.cpu().numpy()
        if rows is not None:                       # rows outside the slab are left at 0 by the library and by the oracle
            y0, y1 = rows[0] * 16, min(rows[1] * 16, H)
            assert np.all(got[:, :y0] == 0) and np.all(got[:, y1:] == 0), tag
        err = np.abs(got - fr64.color).max(0)
        tol = 1e-5 * max(1.0, float(np.abs(fr64.color).max()))      # synthetic SH colours are not confined to [0, 1]
        assert err[strict].max(initial=0.0) <= tol, f"{tag}: pixel error {err[strict].max():.3e} (tolerance {tol:.1e})"
        assert err.max(initial=0.0) <= 2e-2 * max(1.0, float(np.abs(fr64.color).max())), tag
        want = fr64.backward(gimg)
        grads = dict(means3D=g[0], means2D=g[1], shs=g[2], opacities=g[4], scales=g[5], rotations=g[6])
        grads = {k: v.cpu().numpy() for k, v in grads.items()}
        try:
            live, strict_live = _check_grads(fr64, want, grads, ["means3D", "means2D", "opacities", "shs", "scales", "rotations"],
                                             masked=True)
            n_live += live
        except AssertionError as e:
            raise AssertionError(f"{tag}: {e}") from None
    print(f"sweep: {n_live} Gaussians with a non-zero gradient, all held to the strict bound")


def test_bench_line_contract():
    """bench.py on the small workload, in a child process: exactly one JSON line on stdout with the fields the driver and
    the judge read (metric / value / unit / n_gpus / steps / warmup / ms_per_step / higher_is_better / scaling /
    vs_baseline / dtype / data / config.workload + roofline{bound, achieved, peak, unit, frac, traffic} +
    cpu_baseline), and self-consistent numbers."""
    import json
    import subprocess
    import sys
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", "cfg2", "--steps", "4", "--warmup", "1",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "images/s" and d["dtype"] == "f32" and d["data"] == "synthetic" and d["vs_baseline"] is None
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 1000.0 / d["ms_per_step"]) <= 1e-2 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] == "GB/s" and r["peak"] == 8000.0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-4 and 0 < r["frac"] < 1
    assert r["traffic"] is None                      # PMC traffic is only attached to the run it was measured on (cfg3)
    assert r["kernels_over_hbm_peak"] == []          # the per-kernel algorithmic-byte table charges nothing twice
    assert d["cpu_baseline"] is None                 # --no-cpu-baseline


def test_getter_fusion_is_transparent():
    """FUSE_GETTERS (opt-in): render() exactly as the reference writes it — getters, standard forward() — with the flag on
    reaches the raw-parameter kernels through the autograd-history match: same image, same gradients on the leaves;
    arguments that are not the getters' outputs (a scaled opacity here) take the plain path."""
    import diff_gaussian_rasterization as dgr
    from gaussian_params import GaussianParams, Pipe
    from gaussian_renderer import render
    W, H = 320, 208
    scene = S.make_scene(20_000, W, H, 3, 23, scale_lo=0.005, scale_hi=0.06).to(DEV)
    cam = S.make_camera(W, H).to(DEV)
    bg = torch.zeros(3, device=DEV)
    gimg = S.make_grad_image(W, H, 6).to(DEV)
    seen = []
    orig = dgr._RasterizeGaussiansRaw.apply
    results = []
    try:
        for on in (False, True):
            dgr.FUSE_GETTERS = on
            dgr._RasterizeGaussiansRaw.apply = staticmethod(lambda *a, **k: (seen.append(on), orig(*a, **k))[1])
            model = GaussianParams(scene).to(DEV)
            out = render(cam, model, Pipe(), bg)
            out["render"].backward(gimg)
            torch.cuda.synchronize()
            results.append((out["render"].detach(), {n: p.grad.clone() for n, p in model.named_parameters()},
                            out["viewspace_points"].grad.clone()))
        assert seen == [True]                                   # the raw path ran exactly once: with the flag on
        (ia, ga, va), (ib, gb, vb) = results
        derr = (ia - ib).abs().amax(0)
        assert float((derr > 2e-5).float().mean()) <= 1e-4 and float(derr.max()) <= 8e-3
        for n in ga:
            assert float((ga[n] - gb[n]).norm() / ga[n].norm()) <= 1e-4, n
        assert float((va - vb).norm() / va.norm()) <= 1e-4
        # a look-alike: opacity scaled after the sigmoid -> no match, plain path, still correct
        seen.clear()
        model = GaussianParams(scene).to(DEV)
        rs = _settings(raster_kwargs(scene.to("cpu"), S.make_camera(W, H)))
        color, _ = dgr.GaussianRasterizer(rs)(means3D=model.get_xyz, means2D=torch.zeros(scene.P, 3, device=DEV), shs=model.get_features,
                                              opacities=model.get_opacity * 0.5, scales=model.get_scaling, rotations=model.get_rotation)
        assert seen == [] and torch.isfinite(color).all()
    finally:
        dgr.FUSE_GETTERS = False
        dgr._RasterizeGaussiansRaw.apply = orig


def test_frame_with_an_uncovered_region_runs_the_live_filter_and_matches_the_oracle():
    """A frame whose lower half no splat covers: those tiles never close, every planned chunk runs, and the late chunks are most of
    the scene.  They go through the live filter (csrc/gsr_binning.hip k_live_*: only the Gaussians whose rectangle still holds an
    open tile are sorted and binned, one wave each; plan.chunks_filtered).  Pixels and every gradient against the fp64 oracle, and
    the depth order stays a permutation of the visible set with the filtered chunks' live part sorted in front."""
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization import _native as N
    W, H, P = 480, 320, 260_000
    scene = S.make_scene(P, W, H, 1, 91, scale_lo=0.01, scale_hi=0.07)
    scene.means3D[:, 1] = -scene.means3D[:, 1].abs() - 0.02 * scene.means3D[:, 2]            # everything in the upper half
    cam = S.make_camera(W, H)
    kw = raster_kwargs(scene, cam)
    rs, inp = _settings(kw), _inputs(kw, False)
    color, radii, fr = dgr.rasterize_forward(inp["means3D"], inp["shs"], None, inp["opacities"], inp["scales"], inp["rotations"], None, rs)
    torch.cuda.synchronize()
    plan = fr.plan
    assert plan.num_chunks >= 2 and plan.chunks_run == plan.num_chunks and plan.chunks_filtered != 0, (plan.num_chunks, plan.chunks_run, plan.chunks_filtered)
    v = N.debug_views(fr.desc, fr.geom_ws, fr.binning_ws, fr.image_ws, plan)
    V = plan.num_visible
    order = v["depth_order"][:V].long()
    assert int(torch.unique(order).numel()) == V
    depth = v["splat_records"][:, 9]
    for c in range(plan.num_chunks):
        if (plan.chunks_filtered >> c) & 1:
            b0, b1 = int(plan.chunk_rank_begin[c]), int(plan.chunk_rank_begin[c + 1])
            lists = v["sorted_gaussian"].long()
            rng = v["ranges"][c].long()
            used = torch.unique(torch.cat([lists[a:b] for a, b in rng.tolist() if b > a] or [lists[:0]]))
            pos = torch.full((fr.desc.P,), -1, dtype=torch.long, device=DEV)
            pos[order[b0:b1]] = torch.arange(b1 - b0, device=DEV)
            n_live_max = int(pos[used].max()) + 1 if used.numel() else 0
            assert 0 < n_live_max < (b1 - b0) // 2, (c, n_live_max, b1 - b0)           # the binned ones sit in front ...
            d = depth[order[b0:b0 + n_live_max]]
            assert bool((d[1:] >= d[:-1]).all())                                          # ... in depth order
    fr64 = oracle.rasterize(dtype=np.float64, parallel=True, **kw)
    gimg = S.make_grad_image(W, H, 4).numpy()
    *_, live = _forward_backward_strict(kw, fr64, gimg, label="uncovered half", parallel=True)
    assert live > 1000


def test_one_call_forward_falls_back_when_the_guessed_workspace_is_too_small():
    """rasterize_forward brings a binning workspace guessed from the last frame of the same shape and runs both stages in one
    native call (gsr_forward).  A guess that does not cover the first chunk (the scene grew) makes that call answer
    GSR_ERR_WORKSPACE after stage 1, nothing of stage 2 enqueued: the two-call path takes over and the frame equals a frame
    rendered without any guess, bit for bit — image, radii and screen-space gradients."""
    import diff_gaussian_rasterization as dgr
    kw = _fixture_kwargs(dict(P=5000, W=256, H=192, D=3, seed=109))
    rs, inp = _settings(kw), _inputs(kw, False)
    args = (inp["means3D"], inp["shs"], None, inp["opacities"], inp["scales"], inp["rotations"], None, rs)
    gimg = S.make_grad_image(256, 192, 9).to(DEV)
    dgr._binning_guess.clear()
    c0, r0, f0 = dgr.rasterize_forward(*args)                     # no guess: preprocess, then render
    s0 = dgr.rasterize_backward_screen(f0, gimg).clone()
    key = next(iter(dgr._binning_guess))
    assert dgr._binning_guess[key] >= int(f0.plan.chunk_instances_max[0])
    c1, r1, f1 = dgr.rasterize_forward(*args)                     # good guess: one call
    s1 = dgr.rasterize_backward_screen(f1, gimg).clone()
    dgr._binning_guess[key] = 7                                   # hopeless guess: one call gives up, two-call path finishes
    c2, r2, f2 = dgr.rasterize_forward(*args)
    s2 = dgr.rasterize_backward_screen(f2, gimg)
    assert dgr._binning_guess[key] >= int(f2.plan.chunk_instances_max[0])
    for c, r, s_ in ((c1, r1, s1), (c2, r2, s2)):
        assert torch.equal(c, c0) and torch.equal(r, r0) and torch.equal(s_, s0)
    dgr._binning_guess.clear()


def _check_bwd_units(fr, v, W, H, rows):
    """The blend backward's unit lists against the frame: every (tile, chunk) pair with walked entries is covered by the
    segments 0 .. ceil(walk / SEG) - 1 exactly once, in the shard of its tile (slab-relative index % 8)."""
    from diff_gaussian_rasterization import _native as N
    SEG = N.bwd_segment_entries()
    Gx, Gy = (W + 15) // 16, (H + 15) // 16
    ty0, ty1 = (0, Gy) if rows is None else rows
    counts = v["bwd_unit_count"].cpu().numpy().astype(np.int64)
    lists = v["bwd_units"].cpu().numpy().astype(np.int64) & 0xFFFFFFFF
    rng = v["ranges"].long().cpu().numpy()[:fr.plan.chunks_run]
    lens = rng[..., 1] - rng[..., 0]
    walk = np.minimum(v["tile_walk"].long().cpu().numpy()[:fr.plan.chunks_run], lens)
    # what the forward recorded = the tile's deepest contributor in that chunk (exact for the chunk a pixel's final record
    # names; with several chunks the earlier ones' depths are a lower bound)
    enc = v["n_contrib"].long().cpu().numpy()
    c_last, n_last = (enc >> 26) - 1, enc & ((1 << 26) - 1)
    inside = np.zeros(Gx * Gy, bool); inside[ty0 * Gx:ty1 * Gx] = True
    for c in range(fr.plan.chunks_run):
        d = np.zeros((Gy * 16, Gx * 16), np.int64)
        d[:H, :W] = np.where(c_last == c, n_last, 0)
        want = d.reshape(Gy, 16, Gx, 16).max(axis=(1, 3)).reshape(-1)
        got = np.where(lens[c] > 0, walk[c], 0)
        if fr.plan.chunks_run == 1:
            np.testing.assert_array_equal(got[inside], want[inside])
        else:
            assert np.all(got[inside] >= want[inside])
    nseg = np.where(lens > 0, (walk + SEG - 1) // SEG, 0)
    nseg[:, ~inside] = 0
    n_units = int(counts.sum())
    assert n_units == int(nseg.sum()) and n_units > 0
    seen = set()
    cap_full, cap_part = v["bwd_unit_caps"]
    for sh in range(8):
        for cls in range(17):
            begin = 0 if cls == 0 else cap_full + (cls - 1) * cap_part
            u = lists[sh, begin:begin + counts[sh, cls]]
            tile, chunk, seg = u[:, 0] & ((1 << 24) - 1), u[:, 0] >> 24, u[:, 1]
            assert np.all((tile - ty0 * Gx) % 8 == sh) and np.all(seg < nseg[chunk, tile])
            length = np.minimum(walk[chunk, tile] - seg * SEG, SEG)           # full segments, then the partial ones by length class
            step = SEG // 16
            lo, hi = (SEG, SEG) if cls == 0 else (max((16 - cls) * step, 1), (17 - cls) * step - 1)
            assert np.all((length >= lo) & (length <= hi)), (sh, cls)
            seen |= set(zip(chunk.tolist(), tile.tolist(), seg.tolist()))
    assert len(seen) == n_units                                               # none twice, none outside its pair: every one exactly once
    return n_units


@pytest.mark.gpu
@pytest.mark.parametrize("W,H,P,rows", [(320, 208, 6000, None), (320, 208, 6000, (3, 9)), (2560, 1616, 40_000, None),
                                        (480, 320, 260_000, None)])
def test_backward_work_units_cover_every_walked_entry_once(W, H, P, rows):
    """K7 runs one wave per (tile, chunk, segment) unit, appended by the forward's waves (csrc/gsr_render.hip).  The last case is the frame with an uncovered
    half of test_frame_with_an_uncovered_region_...: several chunks, chunk-start checkpoints."""
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization import _native as N
    if P == 260_000:
        scene = S.make_scene(P, W, H, 1, 91, scale_lo=0.01, scale_hi=0.07)
        scene.means3D[:, 1] = -scene.means3D[:, 1].abs() - 0.02 * scene.means3D[:, 2]
    else:
        scene = S.make_scene(P, W, H, 1, 77, scale_lo=0.004, scale_hi=0.05)
    kw = raster_kwargs(scene, S.make_camera(W, H))
    rs, inp = _settings(kw), _inputs(kw, False)
    args = (inp["means3D"], inp["shs"], None, inp["opacities"], inp["scales"], inp["rotations"], None, rs)
    _, _, fr = dgr.rasterize_forward(*args, **({} if rows is None else {"tile_rows": rows}))
    g = S.make_grad_image(W, H, 5).to(DEV)
    dgr.rasterize_backward_screen(fr, g)
    torch.cuda.synchronize()
    v = N.debug_views(fr.desc, fr.geom_ws, fr.binning_ws, fr.image_ws, fr.plan)
    n = _check_bwd_units(fr, v, W, H, rows)
    if P == 260_000:
        assert fr.plan.chunks_run >= 2
    print(f"{W}x{H} P={P} rows={rows}: {n} units, {fr.plan.chunks_run} chunk(s)")


@pytest.mark.gpu
def test_early_fill_clears_the_row_flags_and_gradients_are_bitwise_those_of_the_plain_path():
    """A training frame (gradients wanted, sparse geometry backward, binning workspace guessed from the previous frame) runs as
    ONE gsr_forward whose zero fill also clears the blend backward's row-valid flags (plan.tile_order_ready): the unit lists are
    valid and the screen-space gradients equal, bit for bit, those of the same frame run without the early fill (memset in
    gsr_backward_render)."""
    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization import _native as N
    W, H, P = 640, 368, 200_000
    scene = S.make_scene(P, W, H, 1, 123)
    kw = raster_kwargs(scene, S.make_camera(W, H))
    rs, inp = _settings(kw), _inputs(kw, False)
    args = (inp["means3D"], inp["shs"], None, inp["opacities"], inp["scales"], inp["rotations"], None, rs)
    needs = (True, True, True, False, True, True, True, False)
    g = S.make_grad_image(W, H, 6).to(DEV)
    _, _, plain = dgr.rasterize_forward(*args)                                    # also leaves the workspace guess behind
    want = dgr.rasterize_backward_screen(plain, g).clone()
    assert plain.plan.tile_order_ready == 0
    _, _, fr = dgr.rasterize_forward(*args, prepare_needs=needs)
    assert N.effective_binned_ranks(fr.plan) * 4 < P, "the scene must take the sparse path"
    assert fr.pre is not None and fr.pre["grads"].prezeroed == 1 and fr.plan.tile_order_ready == 1
    dgr.prepare_backward(fr, needs, screen_prefix_only=True)
    got = dgr.rasterize_backward_screen(fr, g)
    torch.cuda.synchronize()
    v = N.debug_views(fr.desc, fr.geom_ws, fr.binning_ws, fr.image_ws, fr.plan)
    _check_bwd_units(fr, v, W, H, None)
    n_pref = int(fr.plan.chunk_rank_begin[fr.plan.chunks_run])
    pref = N.frame_arrays(fr.desc, fr.geom_ws)[1][:n_pref].long()
    assert torch.equal(got[pref], want[pref])                                     # (rows outside the binned prefix: never cleared here)
    for t in fr.pre["tensors"] if fr.pre else ():                                # the early fill cleared every parameter gradient
        assert t is None or float(t.abs().max()) == 0.0
