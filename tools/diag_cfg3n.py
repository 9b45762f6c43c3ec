"""Diagnostic: cfg3n forward through (a) the standard API with fp32-activated inputs vs oracle(f64 of the same fp32 inputs), and
(b) the raw path vs oracle.rasterize_raw; prints the distribution of pixel errors on strict pixels."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "structured-gaussian-splatting_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np, torch
import oracle, scene_synth as S
from util import raster_kwargs
import test_gpu_parity as T
import test_gpu_timed_path as TP

wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3n"
scene, cam = S.make_config(wl)
kw = raster_kwargs(scene, cam)
fr_a = oracle.rasterize(dtype=np.float64, parallel=True, **kw)
color, radii, _ = T._run_gpu(kw)
def report(tag, fr, color, radii):
    strict = T._strict_pixels(fr, radii, exact_radii=False)
    err = np.abs(color.astype(np.float64) - fr.color).max(0)
    bad = (err > 1e-5) & strict
    print(tag, "radii differ:", int((radii != fr.radii).sum()), "strict frac", strict.mean(), "bad strict px:", int(bad.sum()),
          "max", err[strict].max(), "fragile px", int((fr.fragile_px != 0).sum()))
    ys, xs = np.nonzero(bad)
    for y, x in list(zip(ys, xs))[:10]:
        print("   px", x, y, "err", err[y, x], "T", fr.final_T[y, x], "n_contrib", fr.n_contrib[y, x])
    return bad
report("standard", fr_a, color, radii)
fr_r = TP._oracle_raw(scene, cam)
c2, r2, _ = TP._render_timed_path(scene, cam, (0, 0, 0), lambda c, r: np.zeros((3, cam.image_height, cam.image_width), np.float32))
bad = report("raw", fr_r, c2, r2)
print("raw vs standard image diff max", np.abs(c2 - color).max(), "radii differ", int((r2 != radii).sum()))
print("oracle raw vs oracle act diff max", np.abs(fr_r.color - fr_a.color).max())
