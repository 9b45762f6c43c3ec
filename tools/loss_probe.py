import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "structured-gaussian-splatting_amd")]
import torch
import scene_synth as S
from gaussian_params import GaussianParams, Pipe
from gaussian_renderer import render
from scene import GaussianModel
import loss_utils
from diff_gaussian_rasterization import _native as N
dev = "cuda:0"
cfg = S.CONFIGS["cfg3"]
scene, cam = S.make_config("cfg3"); scene, cam = scene.to(dev), cam.to(dev)
gm = GaussianModel(scene.sh_degree); gm.adopt_scene(scene, device=dev)
gp = GaussianParams(scene).to(dev)
bg = torch.zeros(3, device=dev); gt = torch.rand(3, cfg["H"], cfg["W"], device=dev); pipe = Pipe()
fused = lambda im, g: loss_utils.training_loss(im, g)
two = lambda im, g: (1.0 - 0.2) * loss_utils.l1_loss(im, g) + 0.2 * (1.0 - loss_utils.ssim(im, g))
def run(m, loss, n=20):
    ps = list(m._t.values()) if hasattr(m, "_t") else list(m.parameters())
    def step():
        for p in ps: p.grad = None
        out = render(cam, m, pipe, bg); loss(out["render"], gt).backward()
    for _ in range(5): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): step()
    torch.cuda.synchronize(); return 1e3 * (time.perf_counter() - t0) / n
for name, m in (("model", gm), ("store", gp), ("model", gm)):
    print(name, "fused %.3f ms" % run(m, fused), "two-call %.3f ms" % run(m, two), flush=True)
N.profile_enable(True)
run(gm, two, 10); torch.cuda.synchronize()
p = N.profile_read(); N.profile_enable(False)
print({k: (round(1e3 * ms / n, 1), n) for k, (ms, n) in p.items()})
import diff_gaussian_rasterization as dgr
print("--- bench sequence")
print("model fused %.3f" % run(gm, fused, 30))
print("store fused %.3f" % run(gp, fused, 30))
pipe.fused_activations = True
print("store fused_act %.3f" % run(gp, fused, 30))
pipe.fused_activations = False
dgr.FUSE_GETTERS = True
print("store getter_fusion %.3f" % run(gp, fused, 30))
dgr.FUSE_GETTERS = False
print("model two-call %.3f" % run(gm, two, 30))
print("model fused %.3f" % run(gm, fused, 30))
print("store two-call %.3f" % run(gp, two, 30))
