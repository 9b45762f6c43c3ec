#!/usr/bin/env python3
"""Profiling aid: N iterations of the full training loop at cfg3 (what `bench.py --train-loop` times), nothing else,
so that `rocprofv3 --kernel-trace --stats -- python3 profiles/train_loop_probe.py 100 1` shows its kernels only.
argv: iterations, fused_activations (0/1)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "structured-gaussian-splatting_amd"))
import torch

import scene_synth as S
from dataclasses import replace
from gaussian_params import Pipe
from gaussian_renderer import render
from scene import GaussianModel, OptimizationDefaults
from train_loop import train

iters, fused = int(sys.argv[1]), bool(int(sys.argv[2]))
dev = torch.device("cuda", 0)
cfg = S.CONFIGS["cfg3"]
cams = [c.to(dev) for c in S.arc_cameras(cfg["W"], cfg["H"], 8)]
bg = torch.zeros(3, device=dev)
truth = GaussianModel(cfg["D"])
truth.adopt_scene(S.make_scene(cfg["P"], cfg["W"], cfg["H"], cfg["D"], 30), device=dev)
with torch.no_grad():
    targets = [render(c, truth, Pipe(), bg)["render"].clone() for c in cams]
del truth
gm = GaussianModel(cfg["D"])
gm.adopt_scene(S.make_scene(cfg["P"], cfg["W"], cfg["H"], cfg["D"], cfg["seed"]), device=dev)
opt = replace(OptimizationDefaults(), densify_from_iter=0)
gm.training_setup(opt)
pipe = Pipe()
pipe.fused_activations = fused
train(gm, cams, targets, opt, pipe, bg, iterations=100, scene_extent=6.0)      # includes one densify + prune
torch.cuda.synchronize()
t0 = time.perf_counter()
train(gm, cams, targets, opt, pipe, bg, iterations=100 + iters, first_iter=101, scene_extent=6.0)
torch.cuda.synchronize()
print(f"{iters} iterations from 101 (densify every 100), fused={fused}: {1e3 * (time.perf_counter() - t0) / iters:.3f} ms/it")
