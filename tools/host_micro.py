"""Cost of the host-side helpers every native call goes through (microseconds per call, on the GPU box)."""
import os, sys, timeit
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "structured-gaussian-splatting_amd")]
import ctypes as C
import torch
from diff_gaussian_rasterization import _native as N
dev = torch.device("cuda:0")
x = torch.empty(1000, 3, device=dev)
lib = N.load()
def t(name, f, n=20000):
    f(); print("%-58s %.2f us" % (name, timeit.timeit(f, number=n) / n * 1e6))
t("_native._stream(device)", lambda: N._stream(dev))
t("torch.cuda.current_stream(dev).cuda_stream", lambda: torch.cuda.current_stream(dev).cuda_stream)
t("torch._C._cuda_getCurrentRawStream(0)", lambda: torch._C._cuda_getCurrentRawStream(0))
def ctx():
    with torch.cuda.device(dev): pass
t("with torch.cuda.device(dev): pass", ctx)
t("torch.cuda.current_device()", lambda: torch.cuda.current_device())
t("_native._ptr(tensor)", lambda: N._ptr(x))
t("lib.gsr_version()  (ctypes call, no arguments)", lambda: lib.gsr_version())
t("torch.empty(1000, 3, device=dev)", lambda: torch.empty(1000, 3, device=dev))
t("torch.empty_like(x)", lambda: torch.empty_like(x))
t("x.detach()", lambda: x.detach())
t("x.contiguous()", lambda: x.contiguous())
t("x.is_contiguous()", lambda: x.is_contiguous())
class F(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a): return a.detach()
    @staticmethod
    def backward(ctx, g): return g
y = x.clone().requires_grad_(True)
t("autograd.Function.apply (trivial forward)", lambda: F.apply(y))
desc = N.make_desc(1000, 3, 16, 640, 480, 0.5, 0.5, 1.0, False, False)
t("N.workspace_sizes(desc)  (ctypes, 3 arguments)", lambda: N.workspace_sizes(desc))
t("N.make_desc(...)", lambda: N.make_desc(1000, 3, 16, 640, 480, 0.5, 0.5, 1.0, False, False))
