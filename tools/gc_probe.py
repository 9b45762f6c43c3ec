"""What the cyclic garbage collector costs the training step: every collection during 600 steps (generation, duration) and
the step time with the collector as it is, frozen after set-up (gc.freeze()), and disabled."""
import gc, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "structured-gaussian-splatting_amd")]
import torch
import scene_synth as S
from gaussian_params import Pipe
import gaussian_renderer
from scene import GaussianModel
import loss_utils
dev = "cuda:0"
cfg = S.CONFIGS["cfg3"]
scene, cam = S.make_config("cfg3"); scene, cam = scene.to(dev), cam.to(dev)
gm = GaussianModel(scene.sh_degree); gm.adopt_scene(scene, device=dev)
bg = torch.zeros(3, device=dev); gt = torch.rand(3, cfg["H"], cfg["W"], device=dev); pipe = Pipe()
ps = list(gm._t.values())
def step():
    for p in ps: p.grad = None
    out = gaussian_renderer.render(cam, gm, pipe, bg)
    loss_utils.training_loss(out["render"], gt).backward()
log, t_start = [], [0]
def cb(phase, info):
    if phase == "start": t_start[0] = time.perf_counter()
    else: log.append((info["generation"], time.perf_counter() - t_start[0], info["collected"]))
gc.callbacks.append(cb)
def run(label, n=600):
    for _ in range(30): step()
    torch.cuda.synchronize(); log.clear()
    t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    by = {}
    for g, d, c in log: by.setdefault(g, []).append(d)
    print("   objects collected:", sum(c for _, _, c in log), " tracked now:", len(gc.get_objects()))
    print("%-34s %.4f ms/step   collections: %s" % (label, 1e3 * dt / n, {g: "%d x %.2f ms (max %.2f)" % (len(v), 1e3 * sum(v) / len(v), 1e3 * max(v)) for g, v in sorted(by.items())}),
          " total GC %.1f ms = %.3f ms/step" % (1e3 * sum(d for _, d, _ in log), 1e3 * sum(d for _, d, _ in log) / n))
print("tracked objects:", len(gc.get_objects()), "thresholds", gc.get_threshold())
if os.environ.get("WHAT"):
    import collections
    for _ in range(30): step()
    gc.collect(); gc.disable()
    before = collections.Counter(type(o).__name__ for o in gc.get_objects())
    for _ in range(100): step()
    after = collections.Counter(type(o).__name__ for o in gc.get_objects())
    print("growth over 100 steps (collector off):", {k: v - before.get(k, 0) for k, v in after.items() if v - before.get(k, 0) > 20})
    gc.set_debug(gc.DEBUG_SAVEALL); gc.enable(); gc.collect(); gc.set_debug(0)
    kinds = collections.Counter(type(o).__name__ for o in gc.garbage)
    print("cyclic garbage of those 100 steps:", kinds.most_common(12))
    gc.garbage.clear()
run("collector as it is")
gc.collect(); gc.freeze()
run("after gc.freeze()")
gc.disable()
run("collector disabled")
