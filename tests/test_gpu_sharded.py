"""Two (and three) ranks sharing the ONE GPU of the test box: the native HIP backend renders real tile-row
slabs in separate processes and the slab/gradient exchange of diff_gaussian_rasterization/sharded.py runs over
gloo (device tensors staged through the host; RCCL refuses two ranks on one device).  The result must equal the
single-process render: the image bit for bit, gradients to fp32 summation order."""
import math
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import scene_synth as S

pytestmark = pytest.mark.gpu


def _scene(which="small"):
    if which == "cfg4":          # BASELINE.json configs[3]: the cfg3 scene (1e6 Gaussians, 1920x1080, SH 3), tile-row sharded
        return S.make_config("cfg3")
    return S.make_scene(30000, 400, 304, 3, 401, scale_lo=0.004, scale_hi=0.07), S.make_camera(400, 304)


def _render(rank, world, mode, which="small"):
    """mode: a backward_mode of ShardedRenderer, or "loss": default exchange + the train.py loss (slab-local on the ranks)."""
    from gaussian_params import GaussianParams, Pipe
    from gaussian_renderer import render
    from loss_utils import training_loss
    dev = "cuda:0"
    scene, cam = _scene(which)
    Wd, Ht = cam.image_width, cam.image_height
    cam = cam.to(dev)
    if mode == "packed":         # this package's GaussianModel: rendered from its raw leaves (raw mode 2), on one GPU and on the ranks
        from scene import GaussianModel
        model = GaussianModel(scene.sh_degree)
        model.adopt_scene(scene, device=dev)
    else:
        model = GaussianParams(scene.to(dev)).to(dev)
    bg = torch.tensor([0.2, 0.1, 0.3], device=dev)
    gt = torch.rand(3, Ht, Wd, generator=torch.Generator().manual_seed(77)).to(dev)
    loss_value = None
    pipe = Pipe()
    if mode != "packed":
        pipe.fused_activations = mode == "fused"      # raw parameters into the kernels, on one GPU and on the ranks alike
    if world == 1:
        out = render(cam, model, pipe, bg)
        if mode == "loss":
            loss = training_loss(out["render"], gt)
    else:
        from diff_gaussian_rasterization.sharded import ShardedRenderer
        sr = ShardedRenderer(dist, world, rank, backward_mode="allreduce_screen" if mode in ("loss", "fused", "packed") else mode)
        out = sr.render(cam, model, pipe, bg)
        if mode == "loss":
            loss = sr.training_loss(out["render"], gt)
    if mode == "loss":
        loss.backward()
        loss_value = float(loss.detach())
    else:
        out["render"].backward(S.make_grad_image(Wd, Ht, 8).to(dev))
    torch.cuda.synchronize()
    res = dict(image=out["render"].detach().cpu().numpy(), radii=out["radii"].cpu().numpy(),
               means2D=out["viewspace_points"].grad.cpu().numpy())
    if loss_value is not None:
        res["loss"] = np.float64(loss_value)
    res.update({n: p.grad.cpu().numpy() for n, p in (model._t.items() if mode == "packed" else model.named_parameters())})
    return res


def _worker(rank, world, port, mode, q, which="small"):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        q.put((rank, _render(rank, world, mode, which)))
    finally:
        dist.destroy_process_group()


def _run_ranks(world, mode, which="small", timeout=300):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, mode, q, which)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=timeout) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return results


def _assert_equal_single(results, want, world, tol=5e-6):
    for r in range(world):
        got = results[r]
        assert np.array_equal(got["image"], want["image"]) and np.array_equal(got["radii"], want["radii"])
        for k in want:
            if k in ("image", "radii"):
                continue
            scale = max(np.abs(want[k]).max(), 1e-30)
            assert np.abs(got[k] - want[k]).max() <= tol * scale, (r, k, np.abs(got[k] - want[k]).max(), scale)


@pytest.mark.parametrize("world,mode", [(2, "allreduce_screen"), (3, "allreduce_screen"), (2, "reduce_scatter"), (3, "loss"),
                                        (2, "fused"), (2, "packed")])
def test_native_slabs_in_separate_processes_equal_single_render(world, mode):
    results = _run_ranks(world, mode)
    _assert_equal_single(results, _render(0, 1, mode), world)


@pytest.mark.parametrize("mode", ["allreduce_screen", "loss"])
def test_cfg4_full_size_two_ranks_equal_single_render(mode):
    """BASELINE.json configs[3] at FULL size (1e6 Gaussians, 1920x1080, SH 3) through ShardedRenderer: two processes on the
    one GPU of the box, each rendering its tile-row slab of the cfg3 frame, slabs and the screen-space gradient prefix
    exchanged (gloo-staged here; RCCL on a multi-GPU node: test_two_rank_rccl...).  The frame equals the plain render()
    bit for bit, the parameter gradients to fp32 summation order; with mode "loss" the step is the bench's N > 1 step
    (slab-local L1/D-SSIM)."""
    results = _run_ranks(2, mode, "cfg4", timeout=600)
    _assert_equal_single(results, _render(0, 1, mode, "cfg4"), 2)


def _rccl_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    try:
        from gaussian_params import GaussianParams, Pipe
        from diff_gaussian_rasterization.sharded import ShardedRenderer
        scene, cam = _scene()
        cam = cam.to(dev)
        bg = torch.tensor([0.2, 0.1, 0.3], device=dev)
        gt = torch.rand(3, 304, 400, generator=torch.Generator().manual_seed(77)).to(dev)
        res = {}
        for mode in ("allreduce_screen", "reduce_scatter", "loss", "allreduce_screen/sync_gather"):
            model = GaussianParams(scene.to(dev)).to(dev)
            sr = ShardedRenderer(dist, world, rank, backward_mode="reduce_scatter" if mode == "reduce_scatter" else "allreduce_screen",
                                 async_gather=not mode.endswith("sync_gather"))          # A/B of the asynchronous all-gather (unverified at world > 1)
            out = sr.render(cam, model, Pipe(), bg)
            if mode == "loss":
                sr.training_loss(out["render"], gt).backward()
            else:
                out["render"].backward(S.make_grad_image(400, 304, 8).to(dev))
            torch.cuda.synchronize(dev)
            r = dict(image=out["render"].detach().cpu().numpy(), radii=out["radii"].cpu().numpy(),
                     means2D=out["viewspace_points"].grad.cpu().numpy())
            r.update({n: p.grad.cpu().numpy() for n, p in model.named_parameters()})
            res[mode] = r
        q.put((rank, res))
    finally:
        dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL refuses two ranks on one device)")
def test_two_rank_rccl_equals_single_render():
    """world_size 2 over RCCL ("nccl"), one rank per GPU: the NCCL branches of sharded._Comm (all_gather_into_tensor with the
    piggy-backed prefix words, reduce_scatter_tensor, the deferred MAX through pinned memory, the 8-byte loss all-reduce)
    against the single-GPU render.  Skipped on the one-GPU box; runs wherever two devices are visible."""
    world = 2
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_rccl_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for mode in ("allreduce_screen", "reduce_scatter", "loss", "allreduce_screen/sync_gather"):
        want = _render(0, 1, mode.split("/")[0])
        want.pop("loss", None)
        _assert_equal_single({r: results[r][mode] for r in range(world)}, want, world)


def test_slab_local_loss_pieces_add_up_to_the_full_loss():
    """gsr_loss_l1_ssim_forward_rows / backward_rows in one process: the partial sums of three slabs (16-row aligned, as the
    renderer's are, and one ragged) add up to the full-frame sums, and the slabs' gradient rows tile the full gradient."""
    from diff_gaussian_rasterization import _native as N
    import loss_utils
    dev = "cuda:0"
    g = torch.Generator().manual_seed(5)
    for (H, W, cuts) in ((304, 400, (0, 96, 208, 304)), (67, 101, (0, 16, 48, 67)), (40, 36, (0, 0, 32, 40))):
        a = torch.rand(3, H, W, generator=g).to(dev).requires_grad_(True)
        b = torch.rand(3, H, W, generator=g).to(dev)
        full = loss_utils.training_loss(a, b, 0.2)
        full.backward()
        ws = torch.empty(N.loss_workspace_size(3, H, W), dtype=torch.uint8, device=dev)
        sums = torch.zeros(2, device=dev)
        grad = torch.full_like(a, float("nan")).detach()
        up = torch.ones(1, device=dev)
        ad = a.detach()
        for y0, y1 in zip(cuts[:-1], cuts[1:]):
            out2 = torch.empty(2, device=dev)
            N.loss_forward_rows(ad, b, ws, out2, y0, y1)
            sums += out2
            piece = torch.zeros_like(ad)
            N.loss_backward_rows(ad, b, 0.2, up, ws, piece, y0, y1)
            assert float(piece[:, :y0].abs().sum()) == 0 and float(piece[:, y1:].abs().sum()) == 0
            grad[:, y0:y1] = piece[:, y0:y1]
        n = a.numel()
        loss = 0.8 * sums[0] / n + 0.2 * (1 - sums[1] / n)
        assert abs(float(loss) - float(full.detach())) < 2e-6, (H, W)
        assert torch.allclose(grad, a.grad, rtol=1e-5, atol=1e-9), (H, W, float((grad - a.grad).abs().max()))


def _balance_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from gaussian_params import GaussianParams, Pipe
        from diff_gaussian_rasterization.sharded import ShardedRenderer
        from loss_utils import training_loss
        dev = "cuda:0"
        scene, cam = _uneven_scene()
        cam = cam.to(dev)
        model = GaussianParams(scene.to(dev)).to(dev)
        bg = torch.zeros(3, device=dev)
        gt = torch.rand(3, 304, 400, generator=torch.Generator().manual_seed(7)).to(dev)
        sr = ShardedRenderer(dist, world, rank, balance_every=1)
        hist = []
        for it in range(3):
            for p in model.parameters():
                p.grad = None
            out = sr.render(cam, model, Pipe(), bg)
            hist.append(sr.pixel_rows(304))
            sr.training_loss(out["render"], gt).backward()
        torch.cuda.synchronize()
        q.put((rank, dict(image=out["render"].detach().cpu().numpy(), rows=hist,
                          grads={n: p.grad.cpu().numpy() for n, p in model.named_parameters()})))
    finally:
        dist.destroy_process_group()


def _uneven_scene():
    """Every Gaussian in the top third of the frame: equal-row slabs would leave the lower ranks idle."""
    scene, cam = S.make_scene(30000, 400, 304, 1, 402, scale_lo=0.004, scale_hi=0.05), S.make_camera(400, 304)
    scene.means3D[:, 1] = -scene.means3D[:, 1].abs() * 0.6 - 0.15 * scene.means3D[:, 2]
    return scene, cam


def test_slab_balancer_moves_the_boundaries_and_keeps_the_result():
    """balance_every=1 with 3 ranks on a frame whose splats sit in the top third: after the first (equal-rows) frame the
    measured per-tile-row work moves the slab boundaries up, on every rank alike; image and gradients still equal the
    single-GPU step."""
    from gaussian_params import GaussianParams, Pipe
    from gaussian_renderer import render
    from loss_utils import training_loss
    world = 3
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_balance_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    dev = "cuda:0"
    scene, cam = _uneven_scene()
    model = GaussianParams(scene.to(dev)).to(dev)
    gt = torch.rand(3, 304, 400, generator=torch.Generator().manual_seed(7)).to(dev)
    out = render(cam.to(dev), model, Pipe(), torch.zeros(3, device=dev))
    training_loss(out["render"], gt).backward()
    want_img = out["render"].detach().cpu().numpy()
    rows0 = [results[r]["rows"][0] for r in range(world)]
    rows2 = [results[r]["rows"][2] for r in range(world)]
    assert rows0 == [(0, 112), (112, 208), (208, 304)]                      # 19 tile rows split 7 / 6 / 6
    assert rows2[0][1] < rows0[0][1] and rows2[1][1] < rows0[1][1]          # boundaries moved towards the busy top
    assert rows2[0][0] == 0 and rows2[2][1] == 304 and rows2[0][1] == rows2[1][0] and rows2[1][1] == rows2[2][0]
    for r in range(world):
        assert np.array_equal(results[r]["image"], want_img)
        for n, p in model.named_parameters():
            w = p.grad.cpu().numpy()
            scale = max(np.abs(w).max(), 1e-30)
            assert np.abs(results[r]["grads"][n] - w).max() <= 5e-6 * scale, (r, n)


@pytest.mark.gpu
@pytest.mark.parametrize("P,frac", [(1, 1.0), (5000, 0.3), (300_000, 0.02), (4_500_000, 0.5)])
def test_exchange_rows_gather_and_scatter_equal_the_torch_expression(P, frac):
    """gsr_exchange_rows_gather / _scatter (the sharded backward's gradient exchange) against what they replace:
    idx = nonzero((keys >= 0) & (keys <= k_max)) in index order, packed = partial[idx], screen[idx] = summed rows.  Keys are
    written into a frame's geometry workspace by hand; 4.5 M Gaussians: more blocks than the 1024 per-block counters."""
    from diff_gaussian_rasterization import _native as N
    dev = "cuda:0"
    g = torch.Generator().manual_seed(P)
    desc = N.make_desc(P, 0, 1, 64, 64, 0.5, 0.5, 1.0, False, False)
    geom_bytes, _ = N.workspace_sizes(desc)
    geom_ws = torch.zeros(geom_bytes, dtype=torch.uint8, device=dev)
    keys = N.frame_arrays(desc, geom_ws)[0]                       # int32 view [P] into the workspace
    vals = torch.randint(0x3E4CCCCD, 0x40C00000, (P,), generator=g, dtype=torch.int64)
    vals[torch.rand(P, generator=g) < 0.25] = 0xFFFFFFFF          # invisible
    keys.copy_(vals.to(torch.int64).where(vals < 2 ** 31, vals - 2 ** 32).to(torch.int32).to(dev))
    visible = vals[vals != 0xFFFFFFFF]
    k_max = int(visible.sort().values[max(int(frac * visible.numel()) - 1, 0)]) if visible.numel() else 0x3E4CCCCD
    partial = torch.randn(P, 12, generator=g).to(dev)
    want_idx = torch.nonzero((keys >= 0) & (keys <= k_max)).view(-1)
    n = int(want_idx.numel())
    rows, packed = N.exchange_rows_gather(desc, geom_ws, k_max, partial, n)
    torch.cuda.synchronize()
    assert torch.equal(rows.long(), want_idx) and torch.equal(packed, partial[want_idx])
    screen = torch.zeros_like(partial)
    N.exchange_rows_scatter(desc, rows, packed * 2.0, screen)
    want = torch.zeros_like(partial)
    want[want_idx] = partial[want_idx] * 2.0
    assert torch.equal(screen, want)
