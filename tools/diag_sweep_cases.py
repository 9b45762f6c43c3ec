import os, sys
ROOT = "/root/repo" if os.path.exists("/root/repo/tests") else os.getcwd()
sys.path[:0] = [ROOT, os.path.join(ROOT, "structured-gaussian-splatting_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch, oracle
import scene_synth as S
import test_gpu_parity as T
from util import raster_kwargs
for (k, P, W, H, D, lo, hi, zmin) in ((22, 64, 142, 150, 2, 0.005, 0.010, 1.0), (30, 1000, 301, 124, 1, 0.01, 0.02, 0.0), (3, 3000, 233, 74, 3, 0.002, 0.010, 2.0)):
    scene, cam = S.make_scene(P, W, H, D, 1000 + k, scale_lo=lo, scale_hi=hi, zmin=zmin), S.make_camera(W, H)
    kw = raster_kwargs(scene, cam)
    fr64 = oracle.rasterize(dtype=np.float64, **kw)
    fr32 = oracle.rasterize(dtype=np.float32, **kw)
    color, radii, _ = T._run_gpu(kw, None) if False else T._run_gpu(kw, S.make_grad_image(W, H, 1).numpy())
    strict = T._strict_pixels(fr64, radii, True)
    e_gpu = np.abs(color.astype(np.float64) - fr64.color).max(0)
    e_32 = np.abs(fr32.color.astype(np.float64) - fr64.color).max(0)
    bad = np.argwhere((e_gpu > 1e-5) & strict)
    print(f"case {k}: strict {strict.mean():.4f}  gpu max {e_gpu[strict].max():.3e} ({len(bad)} px > 1e-5)   fp32 oracle max {e_32[strict].max():.3e} ({int(((e_32 > 1e-5) & strict).sum())} px > 1e-5)")
    for y, x in bad[:4]:
        print("   px", x, y, "gpu err %.3e  fp32-oracle err %.3e  n_contrib %d" % (e_gpu[y, x], e_32[y, x], fr64.n_contrib[y, x] if hasattr(fr64, "n_contrib") else -1))

# case 30 in detail: the GPU's final transmittance and last contributor at the deviating pixels against the oracle's
import diff_gaussian_rasterization as dgr
from diff_gaussian_rasterization import _native as N
k, P, W, H, D, lo, hi, zmin = 30, 1000, 301, 124, 1, 0.01, 0.02, 0.0
scene, cam = S.make_scene(P, W, H, D, 1000 + k, scale_lo=lo, scale_hi=hi, zmin=zmin), S.make_camera(W, H)
kw = raster_kwargs(scene, cam)
fr64 = oracle.rasterize(dtype=np.float64, **kw)
kt = raster_kwargs(scene, cam, as_numpy=False)
dev = "cuda:0"; t = lambda x: x.to(dev).contiguous()
rs = dgr.GaussianRasterizationSettings(kt["image_height"], kt["image_width"], kt["tanfovx"], kt["tanfovy"], t(kt["bg"]), 1.0, t(kt["viewmatrix"]), t(kt["projmatrix"]), scene.sh_degree, t(kt["campos"]), False, True)
color, radii, fr = dgr.rasterize_forward(t(kt["means3D"]), t(kt["shs"]), None, t(kt["opacities"]), t(kt["scales"]), t(kt["rotations"]), None, rs)
torch.cuda.synchronize()
v = N.debug_views(fr.desc, fr.geom_ws, fr.binning_ws, fr.image_ws, fr.plan)
for (x, y) in ((280, 88), (278, 88), (279, 88), (100, 50)):
    print("px", x, y, "final_T gpu %.9f oracle %.9f  rel diff %.2e | colour gpu" % (abs(float(v["final_T"][y, x])), fr64.final_T[y, x], abs(abs(float(v["final_T"][y, x])) - fr64.final_T[y, x]) / fr64.final_T[y, x]),
          color[:, y, x].cpu().numpy(), "oracle", fr64.color[:, y, x])
