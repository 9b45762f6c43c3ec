"""Host-side profile of the training step (cProfile, cumulative), to find CPU time on the critical path."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "structured-gaussian-splatting_amd")]
import torch
import scene_synth as S
from gaussian_params import Pipe
from gaussian_renderer import render
from scene import GaussianModel
import loss_utils
dev = "cuda:0"
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
cfg = S.CONFIGS[wl]
scene, cam = S.make_config(wl); scene, cam = scene.to(dev), cam.to(dev)
gm = GaussianModel(scene.sh_degree); gm.adopt_scene(scene, device=dev)
bg = torch.zeros(3, device=dev); gt = torch.rand(3, cfg["H"], cfg["W"], device=dev); pipe = Pipe()
ps = list(gm._t.values())
def step():
    for p in ps: p.grad = None
    out = render(cam, gm, pipe, bg); loss_utils.training_loss(out["render"], gt).backward()
    return out
for _ in range(10): step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(50): step()
torch.cuda.synchronize(); print("ms/step %.3f" % (1e3 * (time.perf_counter() - t0) / 50))
pr = cProfile.Profile(); pr.enable()
for _ in range(50): step()
torch.cuda.synchronize(); pr.disable()
st = pstats.Stats(pr); st.sort_stats("cumulative").print_stats(45)
