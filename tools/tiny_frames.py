"""One-off: frames smaller than a tile, one pixel wide / high, a tile plus one pixel: finite outputs, pixels against the oracle.  python tools/tiny_frames.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "structured-gaussian-splatting_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch, oracle
import scene_synth as S
import test_gpu_parity as T
from util import raster_kwargs
bad = 0
for (W, H) in ((1, 1), (5, 3), (15, 15), (16, 16), (17, 17), (33, 7), (7, 40), (64, 1), (1, 64)):
    for P in (1, 50, 700):
        scene, cam = S.make_scene(P, max(W, 2), max(H, 2), 2, 77 + W + H + P, scale_lo=0.01, scale_hi=0.1), S.make_camera(W, H)
        kw = raster_kwargs(scene, cam)
        try:
            fr64 = oracle.rasterize(dtype=np.float64, **kw)
            gimg = S.make_grad_image(W, H, 5).numpy()
            color, radii, grads = T._run_gpu(kw, gimg)
            err = np.abs(color.astype(np.float64) - fr64.color).max()
            want = fr64.backward(gimg.astype(np.float64))
            gerr = max(float(np.abs(grads[n] - getattr(want, n, want[n] if isinstance(want, dict) else None)).max()) if False else 0.0 for n in ())  if False else 0.0
            ok = np.isfinite(color).all() and all(np.isfinite(v).all() for v in grads.values())
            print(f"{W}x{H} P={P}: max pixel err {err:.2e} finite={ok}", flush=True)
            if not ok or err > 5e-3: bad += 1
        except Exception as e:
            bad += 1
            print(f"{W}x{H} P={P}: EXC {type(e).__name__} {str(e)[:120]}", flush=True)
print("bad", bad)
