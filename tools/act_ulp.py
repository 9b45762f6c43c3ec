import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "structured-gaussian-splatting_amd")]
import torch
from diff_gaussian_rasterization import _native as N
g = torch.Generator().manual_seed(0)
P = 1_000_000
sc = (torch.randn(P, 3, generator=g) * 2 - 3).cuda(); ro = torch.randn(P, 4, generator=g).cuda(); op = (torch.randn(P, 1, generator=g) * 4).cuda()
mine = N.activations_forward(sc, ro, op)
ref = (torch.exp(sc), torch.nn.functional.normalize(ro), torch.sigmoid(op))
truth = (torch.exp(sc.double()), torch.nn.functional.normalize(ro.double()), torch.sigmoid(op.double()))
eps = torch.finfo(torch.float32).eps
for n, m, r, t in zip(("exp", "normalize", "sigmoid"), mine, ref, truth):
    ulp = t.abs() * eps
    print(n, "mine vs f64: %.2f ulp   torch vs f64: %.2f ulp   mine vs torch: %.2f ulp   bit-equal: %.4f" % (
        float(((m.double() - t).abs() / ulp).max()), float(((r.double() - t).abs() / ulp).max()),
        float(((m - r).double().abs() / ulp).max()), float((m == r).float().mean())))
