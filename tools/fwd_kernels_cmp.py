"""Renders one frame and saves the image: run with GSR_FWD_GROUPS=0 and =1 to compare the two blend forward kernels bit for bit
(python tools/fwd_kernels_cmp.py fix|cfg2|cfg3n out.npy; tests/test_gpu_parity.py::test_both_blend_forward_kernels_render_the_same_bits)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "structured-gaussian-splatting_amd"), os.path.join(ROOT, "tests")]
import numpy as np, torch
import scene_synth as S
import diff_gaussian_rasterization as dgr
from util import raster_kwargs
W = sys.argv[1]
if W == "fix":
    scene = S.make_scene(5000, 256, 192, 3, 109); cam = S.make_camera(256, 192)
else:
    scene, cam = S.make_config(W)
kw = raster_kwargs(scene, cam, as_numpy=False)
dev = "cuda:0"; t = lambda x: x.to(dev).contiguous()
rs = dgr.GaussianRasterizationSettings(kw["image_height"], kw["image_width"], kw["tanfovx"], kw["tanfovy"], t(kw["bg"]), 1.0, t(kw["viewmatrix"]), t(kw["projmatrix"]), scene.sh_degree, t(kw["campos"]), False, False)
color, radii, fr = dgr.rasterize_forward(t(kw["means3D"]), t(kw["shs"]), None, t(kw["opacities"]), t(kw["scales"]), t(kw["rotations"]), None, rs)
torch.cuda.synchronize()
np.save(sys.argv[2], color.cpu().numpy())
