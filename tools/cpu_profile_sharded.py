"""Host-side profile of the SHARDED training step at world size 1 (RCCL), cumulative."""
import cProfile, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "structured-gaussian-splatting_amd")]
import torch
import torch.distributed as dist
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import scene_synth as S
from gaussian_params import Pipe
from scene import GaussianModel
from diff_gaussian_rasterization.sharded import ShardedRenderer
dev = "cuda:0"
cfg = S.CONFIGS["cfg3"]
scene, cam = S.make_config("cfg3"); scene, cam = scene.to(dev), cam.to(dev)
gm = GaussianModel(scene.sh_degree); gm.adopt_scene(scene, device=dev)
bg = torch.zeros(3, device=dev); gt = torch.rand(3, cfg["H"], cfg["W"], device=dev); pipe = Pipe()
sr = ShardedRenderer(dist, 1, 0)
ps = list(gm._t.values())
def step():
    for p in ps: p.grad = None
    out = sr.render(cam, gm, pipe, bg); sr.training_loss(out["render"], gt).backward()
for _ in range(10): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): step()
torch.cuda.synchronize(); print("ms/step %.3f" % (1e3 * (time.perf_counter() - t0) / 50))
pr = cProfile.Profile(); pr.enable()
for _ in range(50): step()
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
dist.destroy_process_group()
