"""Where the blend backward's waves ran: needs a library built with -DGSR_BWD_TRACE (tools/bwd_trace.sh).  Renders the
workload forward + backward a few times, reads {start, end, HW_ID, XCC_ID} of every block of the last k_render_bwd launch and
prints: wave lifetimes, waves per SIMD over time, tiles per SIMD, the tail."""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "structured-gaussian-splatting_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import torch
import scene_synth as S
import diff_gaussian_rasterization as dgr
from diff_gaussian_rasterization import _native as N
from util import raster_kwargs

W = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
scene, cam = S.make_config(W)
kw = raster_kwargs(scene, cam, as_numpy=False)
dev = "cuda:0"
t = lambda x: x.to(dev).contiguous()
rs = dgr.GaussianRasterizationSettings(kw["image_height"], kw["image_width"], kw["tanfovx"], kw["tanfovy"], t(kw["bg"]), 1.0,
                                       t(kw["viewmatrix"]), t(kw["projmatrix"]), scene.sh_degree, t(kw["campos"]), False, False)
rast = dgr.GaussianRasterizer(rs)
p = {k: t(kw[k]).requires_grad_(True) for k in ("means3D", "shs", "opacities", "scales", "rotations")}
m2d = torch.zeros_like(p["means3D"], requires_grad=True)
g = t(S.make_grad_image(kw["image_width"], kw["image_height"], 3))
for _ in range(4):
    color, radii = rast(means3D=p["means3D"], means2D=m2d, opacities=p["opacities"], shs=p["shs"], scales=p["scales"], rotations=p["rotations"])
    color.backward(g)
    for q in list(p.values()) + [m2d]:
        q.grad = None
torch.cuda.synchronize()
n = 65536                      # one record per BLOCK of the last launch (a block = one work unit while the units fit the grid)
buf = (C.c_ulonglong * (4 * n))()
lib = N.load()
assert lib.gsr_debug_bwd_trace(buf, n) == 0
a = np.frombuffer(buf, dtype=np.uint64).reshape(n, 4).astype(np.int64)
a = a[a[:, 0] > a[:, 0].max() - 500_000]          # blocks of the last launch only (5 ms of 100 MHz ticks)
print("blocks of the launch", len(a), " of them with work (> 1 us):", int(((a[:, 1] - a[:, 0]) > 100).sum()))
a = a[(a[:, 1] - a[:, 0]) > 100]
n = len(a)
t0, t1, hw, xcc = a[:, 0], a[:, 1], a[:, 2], a[:, 3] & 0xF
base = t0.min()
t0, t1 = (t0 - base) / 100.0, (t1 - base) / 100.0              # 100 MHz -> microseconds
simd, cu, sh, se = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
cu_key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
simd_key = cu_key * 4 + simd
print("blocks", n, "kernel span %.1f us" % t1.max(), " wave life mean %.1f us  p10 %.1f  p90 %.1f" % ((t1 - t0).mean(), *np.percentile(t1 - t0, [10, 90])))
print("distinct XCC", len(np.unique(xcc)), "CUs", len(np.unique(cu_key)), "SIMDs", len(np.unique(simd_key)))
cnt = np.bincount(np.unique(simd_key, return_inverse=True)[1])
print("tiles per SIMD: min %d  mean %.2f  max %d   histogram %s" % (cnt.min(), cnt.mean(), cnt.max(), np.bincount(cnt).tolist()))
ccnt = np.bincount(np.unique(cu_key, return_inverse=True)[1])
print("tiles per CU  : min %d  mean %.2f  max %d" % (ccnt.min(), ccnt.mean(), ccnt.max()))
xc = np.bincount(xcc)
print("tiles per XCC :", xc.tolist())
# residency over time
edges = np.linspace(0, t1.max(), 25)
nsimd = len(np.unique(simd_key))
print("time us : resident waves per SIMD (mean over SIMDs that exist) / starts in the bin")
for lo, hi in zip(edges[:-1], edges[1:]):
    mid = 0.5 * (lo + hi)
    res = int(((t0 <= mid) & (t1 > mid)).sum())
    print("  %6.1f  %5.2f  %5d" % (mid, res / nsimd, int(((t0 >= lo) & (t0 < hi)).sum())))
# per-SIMD finish time
fin = np.zeros(simd_key.max() + 1); np.maximum.at(fin, simd_key, t1)
fin = fin[fin > 0]
print("per-SIMD last finish: p10 %.1f  p50 %.1f  p90 %.1f  max %.1f us" % (*np.percentile(fin, [10, 50, 90]), fin.max()))
np.save(os.path.join(ROOT, "gpurun_out", "bwd_trace_%s.npy" % W), a)
