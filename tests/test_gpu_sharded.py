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


def _scene():
    return S.make_scene(30000, 400, 304, 3, 401, scale_lo=0.004, scale_hi=0.07), S.make_camera(400, 304)


def _render(rank, world, mode):
    from gaussian_params import GaussianParams, Pipe
    from gaussian_renderer import render
    dev = "cuda:0"
    scene, cam = _scene()
    cam = cam.to(dev)
    model = GaussianParams(scene.to(dev)).to(dev)
    bg = torch.tensor([0.2, 0.1, 0.3], device=dev)
    if world == 1:
        out = render(cam, model, Pipe(), bg)
    else:
        from diff_gaussian_rasterization.sharded import ShardedRenderer
        out = ShardedRenderer(dist, world, rank, backward_mode=mode).render(cam, model, Pipe(), bg)
    out["render"].backward(S.make_grad_image(400, 304, 8).to(dev))
    torch.cuda.synchronize()
    res = dict(image=out["render"].detach().cpu().numpy(), radii=out["radii"].cpu().numpy(),
               means2D=out["viewspace_points"].grad.cpu().numpy())
    res.update({n: p.grad.cpu().numpy() for n, p in model.named_parameters()})
    return res


def _worker(rank, world, port, mode, q):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        q.put((rank, _render(rank, world, mode)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,mode", [(2, "allreduce_screen"), (3, "allreduce_screen"), (2, "reduce_scatter")])
def test_native_slabs_in_separate_processes_equal_single_render(world, mode):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = dict(q.get(timeout=300) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    want = _render(0, 1, mode)
    for r in range(world):
        got = results[r]
        assert np.array_equal(got["image"], want["image"]) and np.array_equal(got["radii"], want["radii"])
        for k in want:
            if k in ("image", "radii"):
                continue
            scale = max(np.abs(want[k]).max(), 1e-30)
            assert np.abs(got[k] - want[k]).max() <= 5e-6 * scale, (r, k, np.abs(got[k] - want[k]).max(), scale)
