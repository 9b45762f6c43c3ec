"""Does the training step reach hipMalloc / hipFree in steady state?  Counts the caching allocator's device allocations per step
and times torch.empty of the gradient-tensor sizes."""
import os, sys, time
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
for _ in range(30): step()
torch.cuda.synchronize()
def stat():
    s = torch.cuda.memory_stats(dev)
    return {k: s[k] for k in ("num_device_alloc", "num_device_free", "num_alloc_retries", "allocation.all.allocated", "segment.all.allocated",
                              "reserved_bytes.all.current", "allocated_bytes.all.current", "allocated_bytes.all.peak") if k in s}
a = stat()
t0 = time.perf_counter()
for _ in range(100): step()
torch.cuda.synchronize()
print("ms/step %.4f" % (1e3 * (time.perf_counter() - t0) / 100))
b = stat()
print({k: (b[k] - a[k]) / 100 if "bytes" not in k else b[k] for k in a})
P = cfg["P"]
for shape in ((P, 3), (P, 16, 3), (P, 1), (P, 4), (P, 12)):
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        t = time.perf_counter(); x = torch.empty(*shape, device=dev); ts.append(time.perf_counter() - t); del x
    print(shape, "torch.empty us: median %.1f max %.1f" % (1e6 * sorted(ts)[10], 1e6 * max(ts)))
# the same while the stream is busy
x = torch.empty(P, 16, 3, device=dev)
for _ in range(3):
    for _ in range(50): x.zero_()
    t = time.perf_counter(); y = torch.empty(P, 16, 3, device=dev); dt = time.perf_counter() - t; del y
    torch.cuda.synchronize()
    print("torch.empty behind 50 queued fills: %.1f us" % (1e6 * dt))
