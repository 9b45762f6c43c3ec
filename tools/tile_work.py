"""Per-tile work of the blend backward on a workload: list entries each tile's wave walks (all earlier chunks + the last
one up to the tile's deepest contributor), its distribution, and what a longest-first launch order would buy."""
import os, sys
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
rs = dgr.GaussianRasterizationSettings(kw["image_height"], kw["image_width"], kw["tanfovx"], kw["tanfovy"], t(kw["bg"]), 1.0, t(kw["viewmatrix"]),
                                       t(kw["projmatrix"]), scene.sh_degree, t(kw["campos"]), False, True)
color, radii, fr = dgr.rasterize_forward(t(kw["means3D"]), t(kw["shs"]), None, t(kw["opacities"]), t(kw["scales"]), t(kw["rotations"]), None, rs)
torch.cuda.synchronize()
v = N.debug_views(fr.desc, fr.geom_ws, fr.binning_ws, fr.image_ws, fr.plan)
Gx, Gy = (kw['image_width'] + 15) // 16, (kw['image_height'] + 15) // 16
rng = v["ranges"].long()[:fr.plan.chunks_run]
lens = (rng[..., 1] - rng[..., 0]).cpu().numpy()                  # [chunks, tiles]
enc = v["n_contrib"].cpu().numpy().astype(np.int64)
shift = int(os.environ.get("LAST_SHIFT", "26"))
c_last, n_last = (enc >> shift) - 1, enc & ((1 << shift) - 1)
H, Wd = enc.shape
pad = np.zeros((Gy * 16, Gx * 16), np.int64); padc = np.full((Gy * 16, Gx * 16), -1, np.int64)
pad[:H, :Wd] = n_last; padc[:H, :Wd] = c_last
key = (padc + 1) * (1 << shift) + pad
tk = key.reshape(Gy, 16, Gx, 16).max(axis=(1, 3)).reshape(-1)
tc, tn = (tk >> shift) - 1, tk & ((1 << shift) - 1)
work = np.zeros(Gx * Gy, np.int64)
for c in range(lens.shape[0]):
    work += np.where(c < tc, lens[c], np.where(c == tc, np.minimum(tn, lens[c]), 0))
print("chunks_run", fr.plan.chunks_run, "tiles", Gx * Gy, "entries walked", int(work.sum()), "mean", work.mean(), "max", work.max())
print("percentiles 50/90/99/99.9:", np.percentile(work, [50, 90, 99, 99.9]))
# list scheduling on 1024 SIMDs x 4 slots, processor sharing approximated as: a SIMD's finish time = sum of its tiles
def makespan(order, slots=4096):
    import heapq
    simd = [0.0] * 1024
    # tiles go to the SIMD that frees a slot first: model each SIMD as 4 slots sharing its rate -> finish = total work / 1
    heap = [(0.0, i) for i in range(1024)]
    load = np.zeros(1024)
    for w in order:
        l, i = heapq.heappop(heap)
        load[i] += w
        heapq.heappush(heap, (load[i], i))
    return load.max(), load.mean()
q, r = divmod(Gx * Gy, 8)
blocks = np.arange(Gx * Gy)
x, i = blocks & 7, blocks >> 3
tile_of_block = x * q + np.minimum(x, r) + i
mx, mean = makespan(work[tile_of_block])
print("launch order now : makespan %.0f  mean load %.0f  ratio %.3f" % (mx, mean, mx / mean))
mx, mean = makespan(np.sort(work)[::-1])
print("longest first    : makespan %.0f  mean load %.0f  ratio %.3f" % (mx, mean, mx / mean))
np.save(os.path.join(ROOT, "gpurun_out", "tile_work_%s.npy" % W), work.reshape(Gy, Gx))
# how well does the list LENGTH (known before the blend) predict the walked depth (known after)?
l0 = lens.sum(axis=0).astype(float)
print("list length per tile: mean %.1f max %d; corr(length, walked) = %.3f" % (l0.mean(), l0.max(), np.corrcoef(l0, work)[0, 1]))
fw = v["tile_walk"][:fr.plan.chunks_run].sum(0).cpu().numpy().astype(float)
print("forward-recorded tile_work vs this estimate: max abs diff", np.abs(fw - work).max())
