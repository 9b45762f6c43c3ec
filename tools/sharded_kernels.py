"""Per-kernel table of the SHARDED training step at world size 1 (RCCL) + the step time: python tools/sharded_kernels.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "structured-gaussian-splatting_amd")]
import torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29534")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import scene_synth as S
from gaussian_params import Pipe
from scene import GaussianModel
from diff_gaussian_rasterization.sharded import ShardedRenderer
from diff_gaussian_rasterization import _native as N
dev = "cuda:0"
W = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
cfg = S.CONFIGS[W]
scene, cam = S.make_config(W); scene, cam = scene.to(dev), cam.to(dev)
gm = GaussianModel(scene.sh_degree); gm.adopt_scene(scene, device=dev)
bg = torch.zeros(3, device=dev); gt = torch.rand(3, cfg["H"], cfg["W"], device=dev); pipe = Pipe()
sr = ShardedRenderer(dist, 1, 0)
ps = list(gm._t.values())
def step():
    for p in ps: p.grad = None
    out = sr.render(cam, gm, pipe, bg); sr.training_loss(out["render"], gt).backward()
for _ in range(20): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 100
for _ in range(n): step()
torch.cuda.synchronize(); print("ms/step %.3f" % (1e3 * (time.perf_counter() - t0) / n))
N.profile_enable(True)
for _ in range(n): step()
torch.cuda.synchronize()
p = N.profile_read(); N.profile_enable(False)
tot = 0.0
for k, (ms, cnt) in sorted(p.items(), key=lambda kv: -kv[1][0]):
    print("%-18s %8.1f us/step  %5.1f launches" % (k, 1e3 * ms / n, cnt / n)); tot += ms
print("sum %.1f us/step" % (1e3 * tot / n))
dist.destroy_process_group()
