"""Where the training loop stalls: host time of every iteration (no per-iteration sync), the cyclic collector's runs and the
caching allocator's device allocations / frees during 300 iterations of bench.py's --train-loop."""
import gc, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "structured-gaussian-splatting_amd")]
import torch
from dataclasses import replace
import scene_synth as S
from gaussian_params import Pipe
from gaussian_renderer import render
from scene import GaussianModel, OptimizationDefaults
from train_loop import train
dev = torch.device("cuda", 0)
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
pipe = Pipe()
train(gm, cams, targets, opt, pipe, bg, iterations=20, scene_extent=6.0)
with torch.no_grad():
    gm.densify_and_prune(opt.densify_grad_threshold, 0.005, 6.0, None)
torch.cuda.synchronize()
if os.environ.get("FREEZE"):
    gc.collect(); gc.freeze()
gcs, g0 = [], [0.0]
def cb(phase, info):
    if phase == "start": g0[0] = time.perf_counter()
    else: gcs.append((info["generation"], 1e3 * (time.perf_counter() - g0[0])))
gc.callbacks.append(cb)
stamps = []
def on_it(it, loss, g): stamps.append(time.perf_counter())
def stat():
    s = torch.cuda.memory_stats(dev); return s["num_device_alloc"], s["num_device_free"], s["reserved_bytes.all.current"]
a = stat()
t0 = time.perf_counter(); stamps.append(t0)
train(gm, cams, targets, opt, pipe, bg, iterations=320, first_iter=21, scene_extent=6.0, on_iteration=on_it)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
b = stat()
d = [1e3 * (y - x) for x, y in zip(stamps[:-1], stamps[1:])]
print("ms/it %.3f   host per iteration: median %.3f  p90 %.3f  max %.1f" % (1e3 * dt / 300, sorted(d)[len(d) // 2], sorted(d)[int(.9 * len(d))], max(d)))
print("iterations over 5 ms (index: ms):", {i: round(x, 1) for i, x in enumerate(d) if x > 5})
print("gc runs:", [(g, round(ms, 1)) for g, ms in gcs if ms > 0.5], "total %.1f ms" % sum(ms for _, ms in gcs))
print("device allocs %d frees %d, reserved %.2f -> %.2f GB" % (b[0] - a[0], b[1] - a[1], a[2] / 1e9, b[2] / 1e9))
