"""The reference's training iteration (train.py:63-146) around the drop-in rasterizer, for synthetic scenes:
LR schedule, SH-degree ramp, render -> L1/D-SSIM -> backward, densification bookkeeping, densify/prune,
opacity reset, Adam step.  SURVEY 8f row f1."""
import random

import torch

from gaussian_renderer import render
from loss_utils import training_loss


def train(gaussians, cameras, targets, opt, pipe, background, iterations, first_iter=1, scene_extent=5.0,
          white_background=False, on_iteration=None):
    """cameras / targets: lists of equal length (Camera-like objects and [3,H,W] ground-truth images)."""
    stack = []
    for it in range(first_iter, iterations + 1):
        gaussians.update_learning_rate(it)
        if it % 1000 == 0:
            gaussians.oneupSHdegree()
        if not stack:
            stack = list(range(len(cameras)))
        idx = stack.pop(random.randint(0, len(stack) - 1))
        bg = torch.rand(3, device=background.device) if opt.random_background else background
        pkg = render(cameras[idx], gaussians, pipe, bg)
        image, vsp, vis, radii = pkg["render"], pkg["viewspace_points"], pkg["visibility_filter"], pkg["radii"]
        loss = training_loss(image, targets[idx], opt.lambda_dssim)
        loss.backward()
        with torch.no_grad():
            if it < opt.densify_until_iter:
                gaussians.update_densification_stats(vsp, radii)       # train.py:127-130, one native pass
                if it > opt.densify_from_iter and it % opt.densification_interval == 0:
                    size_threshold = 20 if it > opt.opacity_reset_interval else None
                    gaussians.densify_and_prune(opt.densify_grad_threshold, 0.005, scene_extent, size_threshold)
                if it % opt.opacity_reset_interval == 0 or (white_background and it == opt.densify_from_iter):
                    gaussians.reset_opacity()
            if it < iterations:
                gaussians.optimizer.step()
                gaussians.optimizer.zero_grad(set_to_none=True)
        if on_iteration is not None:
            on_iteration(it, loss, gaussians)
    return gaussians
