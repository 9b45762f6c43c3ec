"""Per-kernel table of full training iterations (render + loss + backward + densification stats + Adam) on the arc cameras."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "structured-gaussian-splatting_amd")]
import torch
from dataclasses import replace
import scene_synth as S
from gaussian_params import Pipe
from gaussian_renderer import render
from scene import GaussianModel, OptimizationDefaults
from train_loop import train
from diff_gaussian_rasterization import _native as N
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
train(gm, cams, targets, opt, Pipe(), bg, iterations=20, scene_extent=6.0)
torch.cuda.synchronize()
n = 40
N.profile_enable(True)
train(gm, cams, targets, opt, Pipe(), bg, iterations=20 + n, first_iter=21, scene_extent=6.0)
torch.cuda.synchronize()
p = N.profile_read(); N.profile_enable(False)
tot = 0.0
for k, (ms, cnt) in sorted(p.items(), key=lambda kv: -kv[1][0]):
    print("%-18s %8.1f us/it  %5.1f launches/it" % (k, 1e3 * ms / n, cnt / n)); tot += ms
print("sum %.1f us/it" % (1e3 * tot / n))
