"""Per-frame plan statistics inside the training loop (which chunks run, how much they bin)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "structured-gaussian-splatting_amd")]
import torch
import diff_gaussian_rasterization as dgr
from dataclasses import replace
import scene_synth as S
from gaussian_params import Pipe
from gaussian_renderer import render
from scene import GaussianModel, OptimizationDefaults
from loss_utils import training_loss
dev = "cuda:0"
cfg = S.CONFIGS["cfg3"]
cams = [c.to(dev) for c in S.arc_cameras(cfg["W"], cfg["H"], 8)]
bg = torch.zeros(3, device=dev)
gm = GaussianModel(cfg["D"]); gm.adopt_scene(S.make_config("cfg3")[0], device=dev)
orig = dgr.N.forward_render
stats = []
def spy(desc, cam, g, geom, binning, image, plan, color, device, **kw):
    r = orig(desc, cam, g, geom, binning, image, plan, color, device, **kw)
    stats.append((plan.num_chunks, plan.chunks_run, [int(plan.chunk_rank_begin[c]) for c in range(plan.num_chunks + 1)], int(plan.num_rendered), int(plan.instances_emitted)))
    return r
dgr.N.forward_render = spy
for i, c in enumerate(cams):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    with torch.no_grad():
        render(c, gm, Pipe(), bg)
    torch.cuda.synchronize()
    print("cam", i, "%.3f ms" % (1e3 * (time.perf_counter() - t0)), stats[-1])
N = dgr.N
for i in (1, 2):
    with torch.no_grad():
        render(cams[i], gm, Pipe(), bg)
    torch.cuda.synchronize()
    N.profile_enable(True)
    with torch.no_grad():
        for _ in range(5): render(cams[i], gm, Pipe(), bg)
    torch.cuda.synchronize()
    p = N.profile_read(); N.profile_enable(False)
    print("cam", i, {k: (round(1e3 * ms / 5, 1), n // 5) for k, (ms, n) in p.items()}, "sum %.1f us" % (1e3 * sum(ms for ms, _ in p.values()) / 5))
