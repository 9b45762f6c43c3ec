"""Where the training iteration's time goes: the loop of train_loop.train on cfg3 with host-side timers around its phases
(each phase synchronised, so the sum is an upper bound of the pipelined loop)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "structured-gaussian-splatting_amd")]
import torch
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
truth = GaussianModel(cfg["D"]); truth.adopt_scene(S.make_scene(cfg["P"], cfg["W"], cfg["H"], cfg["D"], 30), device=dev)
with torch.no_grad():
    targets = [render(c, truth, Pipe(), bg)["render"].clone() for c in cams]
del truth
gm = GaussianModel(cfg["D"]); gm.adopt_scene(S.make_config("cfg3")[0], device=dev)
opt = replace(OptimizationDefaults(), densify_from_iter=0)
gm.training_setup(opt)
T = {}
def tick(name, t0):
    torch.cuda.synchronize(); T[name] = T.get(name, 0.0) + time.perf_counter() - t0
def run(n, sync_phases):
    for it in range(1, n + 1):
        t0 = time.perf_counter(); gm.update_learning_rate(it)
        pkg = render(cams[it % 8], gm, Pipe(), bg)
        loss = training_loss(pkg["render"], targets[it % 8], opt.lambda_dssim)
        if sync_phases: tick("forward+loss", t0); t0 = time.perf_counter()
        loss.backward()
        if sync_phases: tick("backward", t0); t0 = time.perf_counter()
        with torch.no_grad():
            gm.update_densification_stats(pkg["viewspace_points"], pkg["radii"])
            if sync_phases: tick("densify stats", t0); t0 = time.perf_counter()
            if it % 100 == 0:
                gm.densify_and_prune(opt.densify_grad_threshold, 0.005, 6.0, None)
                if sync_phases: tick("densify+prune (per 100 it)", t0); t0 = time.perf_counter()
            gm.optimizer.step()
            if sync_phases: tick("adam", t0); t0 = time.perf_counter()
            gm.optimizer.zero_grad(set_to_none=True)
            if sync_phases: tick("zero_grad", t0)
run(20, False)
torch.cuda.synchronize(); t0 = time.perf_counter(); run(200, False); torch.cuda.synchronize()
print("pipelined: %.3f ms/it" % (1e3 * (time.perf_counter() - t0) / 200))
run(200, True)
for k, v in T.items():
    print("%-28s %.3f ms/it" % (k, 1e3 * v / 200))
import cProfile, pstats
for p in gm._t.values(): p.grad = None
gm.xyz_gradient_accum += 1e-3; gm.denom += 1.0
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
with torch.no_grad():
    gm.densify_and_prune(opt.densify_grad_threshold, 0.005, 6.0, None)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
