"""world_size-2 (and 3) CPU tests of the multi-GPU path under gloo: the slab partition, the
all-gather of slabs, the reduce-scatter of screen-space gradients, the sharded geometry backward and the
all-gather of parameter gradients in diff_gaussian_rasterization/sharded.py.  The HIP kernels cannot run
here, so the compute provider is the CPU oracle (injected as `backend`; oracle use is confined to tests);
the same file runs over RCCL with the native backend on GPUs.  Result must equal the unsharded oracle.
"""
import math
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import scene_synth as S
from util import raster_kwargs


class OracleBackend:
    """CPU stand-in for NativeBackend with the same three methods (fp64 oracle underneath)."""

    def forward(self, means3D, sh, colors_precomp, opacities, scales, rotations, cov3D_precomp, rs, tile_rows, out_color):
        import oracle
        n = lambda t: None if (t is None or t.numel() == 0) else t.detach().double().numpy()
        fr = oracle.rasterize(dtype=np.float64, image_height=rs.image_height, image_width=rs.image_width, tanfovx=rs.tanfovx,
                              tanfovy=rs.tanfovy, bg=n(rs.bg), scale_modifier=rs.scale_modifier, viewmatrix=n(rs.viewmatrix),
                              projmatrix=n(rs.projmatrix), sh_degree=int(rs.sh_degree), campos=n(rs.campos), means3D=n(means3D),
                              opacities=n(opacities), shs=n(sh), colors_precomp=n(colors_precomp), scales=n(scales),
                              rotations=n(rotations), cov3D_precomp=n(cov3D_precomp), tile_rows=tile_rows)
        H = rs.image_height
        y0, y1 = min(tile_rows[0] * 16, H), min(tile_rows[1] * 16, H)
        if y1 > y0:
            out_color[:, y0:y1] = torch.from_numpy(fr.color[:, y0:y1]).to(out_color.dtype)
        return out_color, torch.from_numpy(fr.radii.copy()), fr

    def backward_screen(self, fr, grad_color):
        scr = fr.backward_screen(grad_color.detach().double().numpy())
        out = torch.zeros(fr.P, 12, dtype=torch.float64)
        out[:, :9] = torch.from_numpy(scr)
        return out.to(grad_color.dtype)

    def binned_prefix(self, fr):
        # the oracle bins everything: no usable prefix -> dense exchange (exercised), or the keyed form: keys = bits of the
        # fp32 view depth (negative = invisible), this rank "binned" everything up to the largest key
        if not getattr(self, "use_prefix", False):
            return None, -1, fr.P
        vis = fr.radii > 0
        keys = np.where(vis, fr.depth.astype(np.float32).view(np.int32), -1).astype(np.int64)
        cut = int(keys[vis].max()) if vis.any() else -1
        return torch.from_numpy(keys), cut, int(((keys >= 0) & (keys <= cut)).sum())

    def backward_geom(self, fr, screen, needs, g0, g1, rows=None):
        g = fr.backward_geom(screen[:, :9].double().numpy(), g0, g1)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(screen.dtype)
        return (t(g["means3D"]), t(g["means2D"]), t(g["shs"]) if fr.M else None,
                t(g["colors_precomp"]) if fr.has_colors else None, t(g["opacities"]),
                None if fr.has_cov else t(g["scales"]), None if fr.has_cov else t(g["rotations"]),
                t(g["cov3D_precomp"]) if fr.has_cov else None)


def _case():
    scene, cam = S.make_scene(300, 96, 112, 2, 61, scale_lo=0.01, scale_hi=0.12), S.make_camera(96, 112)
    return scene, cam


def _worker(rank, world, port, q, mode, use_prefix):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from diff_gaussian_rasterization import GaussianRasterizationSettings
        from diff_gaussian_rasterization.sharded import ShardedRenderer
        scene, cam = _case()
        a = scene.activated()
        leaves = {k: a[k].double().clone().requires_grad_(True) for k in ("means3D", "opacities", "shs", "scales", "rotations")}
        means2D = torch.zeros(scene.P, 3, dtype=torch.float64, requires_grad=True)
        rs = GaussianRasterizationSettings(cam.image_height, cam.image_width, math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2),
                                           torch.tensor([0.1, 0.2, 0.3], dtype=torch.float32).double(), 1.0,
                                           cam.world_view_transform.double(), cam.full_proj_transform.double(), scene.sh_degree,
                                           cam.camera_center.double(), False, False)
        be = OracleBackend()
        be.use_prefix = use_prefix
        sr = ShardedRenderer(dist, world, rank, backend=be, backward_mode=mode)
        image, radii = sr.rasterize(rs, leaves["means3D"], means2D, leaves["opacities"], shs=leaves["shs"],
                                    scales=leaves["scales"], rotations=leaves["rotations"])
        gimg = S.make_grad_image(cam.image_width, cam.image_height, 5).double()
        image.backward(gimg)
        out = dict(image=image.detach().numpy(), radii=radii.numpy(), means2D=means2D.grad.numpy())
        out.update({k: v.grad.numpy() for k, v in leaves.items()})
        q.put((rank, out))
    finally:
        dist.destroy_process_group()


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,mode,use_prefix", [(2, "allreduce_screen", False), (2, "allreduce_screen", True),
                                                   (3, "allreduce_screen", True), (2, "reduce_scatter", False),
                                                   (3, "reduce_scatter", False)])
def test_sharded_equals_single(world, mode, use_prefix):
    import oracle
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, mode, use_prefix)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    scene, cam = _case()
    kw = raster_kwargs(scene, cam, bg=(0.1, 0.2, 0.3))
    fr = oracle.rasterize(dtype=np.float64, **kw)
    want = fr.backward(S.make_grad_image(cam.image_width, cam.image_height, 5).double().numpy())
    for r in range(world):
        got = results[r]
        assert np.abs(got["image"] - fr.color).max() < 1e-14
        np.testing.assert_array_equal(got["radii"], fr.radii)
        for k in ("means3D", "means2D", "opacities", "shs", "scales", "rotations"):
            w = want[k].reshape(got[k].shape)
            assert np.abs(got[k] - w).max() <= 1e-11 * max(1.0, np.abs(w).max()), (r, k)


def test_slab_and_shard_partitions():
    from diff_gaussian_rasterization.sharded import gaussian_shard, slab_bounds
    for rows, world in ((68, 8), (68, 3), (5, 8), (1, 2), (135, 8)):
        s = slab_bounds(rows, world)
        assert s[0][0] == 0 and s[-1][1] == rows and all(a[1] == b[0] for a, b in zip(s, s[1:]))
        assert max(b - a for a, b in s) - min(b - a for a, b in s) <= 1
    w = [1.0] * 10 + [10.0] * 10
    s = slab_bounds(20, 2, w)
    assert s == [(0, 15), (15, 20)] or (s[0][1] >= 14 and s[0][1] <= 16)
    s = slab_bounds(20, 4, [0.0] * 20)
    assert s[0][0] == 0 and s[-1][1] == 20
    for P, world in ((10, 4), (1_000_000, 8), (3, 8), (0, 2)):
        cover = []
        for r in range(world):
            g0, g1, slen = gaussian_shard(P, world, r)
            assert g1 - g0 <= slen
            cover += list(range(g0, g1)) if P < 100 else []
        if P < 100:
            assert cover == list(range(P))
