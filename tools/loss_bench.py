"""Loss kernels alone: (1 - l) L1 + l (1 - SSIM) forward + backward on random images, per-kernel hipEvent times from the library.
python tools/loss_bench.py [H W]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "structured-gaussian-splatting_amd")]
import torch
import loss_utils
from diff_gaussian_rasterization import _native as N
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (1080, 1920)
dev = "cuda:0"
g = torch.Generator(device="cpu").manual_seed(1)
im = torch.rand(3, H, W, generator=g).to(dev).requires_grad_(True)
gt = torch.rand(3, H, W, generator=g).to(dev)
def step():
    im.grad = None
    l = loss_utils.training_loss(im, gt); l.backward(); return l
for _ in range(10): step()
torch.cuda.synchronize()
N.profile_enable(True)
n = 50
for _ in range(n): l = step()
torch.cuda.synchronize()
p = N.profile_read(); N.profile_enable(False)
print("%dx%d loss %.6f  " % (W, H, float(l)), {k: round(1e3 * ms / n, 1) for k, (ms, c) in p.items()}, " grad checksum %.6e" % float(im.grad.double().abs().sum()))
