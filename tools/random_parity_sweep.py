"""One-off confidence sweep: random small frames (odd sizes, tiny to mid splats, every SH degree) through the strict parity check of
tests/test_gpu_parity.py (forward 1e-5 on strict pixels, every Gaussian's gradient at 1e-4).  python tools/random_parity_sweep.py [n] [seed0]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "structured-gaussian-splatting_amd"), os.path.join(ROOT, "tests")]
import numpy as np
import torch
import oracle
import scene_synth as S
import test_gpu_parity as T
from util import raster_kwargs
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
rng = np.random.default_rng(seed0)
bad = 0
for k in range(n):
    P = int(rng.choice([1, 3, 17, 64, 200, 1000, 3000, 6000]))
    W, H = int(rng.integers(17, 330)), int(rng.integers(17, 250))
    D = int(rng.integers(0, 4))
    lo = float(rng.choice([0.002, 0.005, 0.01, 0.03]))
    hi = lo * float(rng.choice([2.0, 5.0, 12.0]))
    zmin = float(rng.choice([0.0, 1.0, 2.0]))
    scene, cam = S.make_scene(P, W, H, D, seed0 + k, scale_lo=lo, scale_hi=hi, zmin=zmin), S.make_camera(W, H)
    kw = raster_kwargs(scene, cam)
    try:
        fr64 = oracle.rasterize(dtype=np.float64, **kw)
        gimg = S.make_grad_image(W, H, seed0 + k).numpy()
        T._forward_backward_strict(kw, fr64, gimg, label=f"sweep {k}")
        print(f"{k}: P={P} {W}x{H} D={D} scales {lo}-{hi:.3f} zmin={zmin}: ok", flush=True)
    except AssertionError as e:
        msg = str(e).splitlines()[0][:160]
        if "too many fragile" in msg:
            print(f"{k}: P={P} {W}x{H} D={D} scales {lo}-{hi:.3f} zmin={zmin}: skipped ({msg})", flush=True)
        else:
            bad += 1
            print(f"{k}: P={P} {W}x{H} D={D} scales {lo}-{hi:.3f} zmin={zmin}: FAILED {msg}", flush=True)
print("failures:", bad)
sys.exit(1 if bad else 0)
