"""Quadrant-mask statistics of the walked tile-list entries of a workload and two cost models of the blend backward:
'lockstep' (one entry at a time for the whole wave, a pass per half that has a wanted quadrant) and 'groups' (four 16-lane groups, one per
quadrant, each walking the entries of its own quadrant; a batch of 64 entries ends when the slowest group is through)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "structured-gaussian-splatting_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import torch
import scene_synth as S
import diff_gaussian_rasterization as dgr
from diff_gaussian_rasterization import _native as N
from util import raster_kwargs

W = sys.argv[1] if len(sys.argv) > 1 else "cfg3n"
scene, cam = S.make_config(W)
kw = raster_kwargs(scene, cam, as_numpy=False)
dev = "cuda:0"
t = lambda x: x.to(dev).contiguous()
rs = dgr.GaussianRasterizationSettings(kw["image_height"], kw["image_width"], kw["tanfovx"], kw["tanfovy"], t(kw["bg"]), 1.0, t(kw["viewmatrix"]),
                                       t(kw["projmatrix"]), scene.sh_degree, t(kw["campos"]), False, True)
color, radii, fr = dgr.rasterize_forward(t(kw["means3D"]), t(kw["shs"]), None, t(kw["opacities"]), t(kw["scales"]), t(kw["rotations"]), None, rs)
torch.cuda.synchronize()
v = N.debug_views(fr.desc, fr.geom_ws, fr.binning_ws, fr.image_ws, fr.plan)
nc = fr.plan.chunks_run
rng = v["ranges"].long()[:nc].cpu().numpy()
walk = v["tile_walk"].long()[:nc].cpu().numpy()
q = v["sorted_quadrants"].cpu().numpy().astype(np.int64)
print("chunks_run", nc, "list entries", len(q))
if nc != 1:
    print("(only the last chunk's list is resident: statistics are of that chunk)")
c = nc - 1
b0, n = rng[c, :, 0], np.minimum(walk[c], rng[c, :, 1] - rng[c, :, 0])
tile_of = np.repeat(np.arange(len(b0)), n)
pos = np.arange(n.sum()) - np.repeat(np.cumsum(n) - n, n)
m = q[np.repeat(b0, n) + pos]
hist = np.bincount(m, minlength=16)
pc = np.array([bin(i).count("1") for i in range(16)])
halves = np.array([(1 if i & 3 else 0) + (1 if i & 12 else 0) for i in range(16)])
print("walked entries", len(m), " mask histogram", hist.tolist())
print("mean quadrants per entry %.3f   mean halves %.3f   entries with no quadrant %.3f" % ((hist * pc).sum() / len(m), (hist * halves).sum() / len(m), hist[0] / len(m)))
for k in range(5):
    print("  %d quadrants: %.3f" % (k, hist[pc == k].sum() / len(m)))
# cost models (cycles per wave): lockstep = per entry 200 + 174 per half; groups = per step 530, steps of a batch = max over quadrants
HEAD, HALF, STEP = float(os.environ.get("HEAD", 200)), float(os.environ.get("HALF", 174)), float(os.environ.get("STEP", 530))
lock = ((m > 0) * HEAD + halves[m] * HALF).sum()
BATCH_N = int(os.environ.get("BATCH_N", 64))
batch = tile_of * 4096 + pos // BATCH_N
ub, inv = np.unique(batch, return_inverse=True)
cnt = np.stack([np.bincount(inv, weights=((m >> g) & 1).astype(float), minlength=len(ub)) for g in range(4)], 1)
steps = cnt.max(1)
print("batches", len(ub), " group items %.0f  steps %.0f  (items / 4 = %.0f: imbalance %.3f)" % (cnt.sum(), steps.sum(), cnt.sum() / 4, steps.sum() / (cnt.sum() / 4)))
grp = steps.sum() * STEP + len(ub) * float(os.environ.get("BATCH", 600))
print("model cycles: lockstep %.3e   groups %.3e   ratio %.3f" % (lock, grp, grp / lock))
