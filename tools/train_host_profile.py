"""cProfile of the training loop's host side (which Python / ctypes calls the iteration spends its time in) + GPU-busy share."""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "structured-gaussian-splatting_amd")]
import torch
from dataclasses import replace
import scene_synth as S
from gaussian_params import Pipe
from gaussian_renderer import render
from scene import GaussianModel, OptimizationDefaults
from train_loop import train
dev = "cuda:0"
cfg = S.CONFIGS["cfg3"]
cams = [c.to(dev) for c in S.arc_cameras(cfg["W"], cfg["H"], 8)]
bg = torch.zeros(3, device=dev)
truth = GaussianModel(cfg["D"]); truth.adopt_scene(S.make_scene(cfg["P"], cfg["W"], cfg["H"], cfg["D"], 30), device=dev)
with torch.no_grad():
    targets = [render(c, truth, Pipe(), bg)["render"].clone() for c in cams]
del truth
gm = GaussianModel(cfg["D"]); gm.adopt_scene(S.make_config("cfg3")[0], device=dev)
opt = replace(OptimizationDefaults(), densify_from_iter=10 ** 9)
gm.training_setup(opt)
train(gm, cams, targets, opt, Pipe(), bg, iterations=30, scene_extent=6.0)
torch.cuda.synchronize()
n = 200
t0 = time.perf_counter()
train(gm, cams, targets, opt, Pipe(), bg, iterations=30 + n, first_iter=31, scene_extent=6.0)
torch.cuda.synchronize()
print("plain: %.3f ms/it" % (1e3 * (time.perf_counter() - t0) / n))
pr = cProfile.Profile()
pr.enable()
train(gm, cams, targets, opt, Pipe(), bg, iterations=30 + 2 * n, first_iter=31 + n, scene_extent=6.0)
torch.cuda.synchronize()
pr.disable()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22)
print(s.getvalue()[:5000])
