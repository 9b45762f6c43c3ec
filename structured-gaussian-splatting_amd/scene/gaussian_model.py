"""GaussianModel — host-side mirror of the reference's parameter store (scene/gaussian_model.py), the producer
of the rasterizer's inputs and the consumer of its `means2D.grad` / `radii` outputs.

Same public surface and semantics as the reference class (getters :101-128, create_from_pcd :134-157,
training_setup :159-177, update_learning_rate :179-185, PLY IO :201-266, reset_opacity :220-223,
densify_and_prune :399-413, add_densification_stats :415-417, capture/restore :67-99), organised differently:
the per-Gaussian tensors live in ONE table, and every structural edit (prune, clone, split, opacity reset) goes
through a single `_rebuild` that rewrites parameters and Adam moments together.

Two things are laid out for the rasterizer rather than as the reference has them (the producer
side of the hot path; 0.45 ms of a 1.4 ms training step at 1e6 Gaussians when done as torch ops):
  * `_features_dc` [P,1,3] and `_features_rest` [P,M-1,3] are the two column ranges of ONE leaf tensor `_features`
    [P,M,3], so `get_features` (the reference's torch.cat, 384 MB of traffic per call at P = 1e6, and as much again
    in its backward) is that tensor itself: no copy forward, and the rasterizer's dL/dshs IS its gradient.  The
    optimizer keeps the reference's six named groups; "f_dc" holds `_features` and steps its first column with its
    own lr and the others with the lr of the (parameter-less) group "f_rest" — element for element what Adam on the
    two separate tensors does (fused_adam.py, gsr_adam_step_split).
  * `get_scaling`, `get_rotation`, `get_opacity` share one native evaluation (scene/activations.py: one HIP launch
    forward, one backward, for all three) instead of ~20 torch kernels.
Runs on whatever device the tensors live on (host tensors use the reference's torch ops); the native pieces it
calls (activations, optimizer step, rasterizer, distCUDA2) are HIP-only.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass

import numpy as np
import torch
from torch import nn

from . import ply_io
from .activations import ActivationCache

SH_C0 = 0.28209479177387814
GROUPS = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")       # optimizer group order of the reference
TABLE = ("xyz", "features", "opacity", "scaling", "rotation")              # this class's leaf tensors (features = f_dc | f_rest)


def _pack(t: dict) -> dict:
    """Reference-named tensors (f_dc [n,1,3], f_rest [n,M-1,3]) -> table-named ones (features [n,M,3])."""
    if "features" in t:
        return {k: t[k] for k in TABLE}
    out = {k: t[k] for k in TABLE if k != "features"}
    out["features"] = torch.cat((t["f_dc"], t["f_rest"]), dim=1)
    return out


@dataclass
class OptimizationDefaults:
    """arguments/__init__.py:76-95 (pinned by tests/golden/params.json)."""
    iterations: int = 30_000
    position_lr_init: float = 0.00016
    position_lr_final: float = 0.0000016
    position_lr_delay_mult: float = 0.01
    position_lr_max_steps: int = 30_000
    feature_lr: float = 0.0025
    opacity_lr: float = 0.05
    scaling_lr: float = 0.005
    rotation_lr: float = 0.001
    percent_dense: float = 0.01
    lambda_dssim: float = 0.2
    densification_interval: int = 100
    opacity_reset_interval: int = 3000
    densify_from_iter: int = 500
    densify_until_iter: int = 15_000
    densify_grad_threshold: float = 0.0002
    random_background: bool = False


def expon_lr(lr_init, lr_final, lr_delay_steps=0, lr_delay_mult=1.0, max_steps=1_000_000):
    """Log-linear interpolation lr_init -> lr_final with an optional warm-up (utils/general_utils.py:29-62;
    pinned by tests/golden/lr.npz)."""
    def at(step):
        if step < 0 or (lr_init == 0.0 and lr_final == 0.0):
            return 0.0
        warm = 1.0
        if lr_delay_steps > 0:
            warm = lr_delay_mult + (1 - lr_delay_mult) * math.sin(0.5 * math.pi * min(max(step / lr_delay_steps, 0), 1))
        t = min(max(step / max_steps, 0), 1)
        return warm * math.exp(math.log(lr_init) * (1 - t) + math.log(lr_final) * t)
    return at


def inverse_sigmoid(x):
    return torch.log(x / (1 - x))


def quat_to_rotmat(q):
    """Normalised quaternion (r,x,y,z) -> R, as utils/general_utils.py:78-99."""
    q = q / q.norm(dim=1, keepdim=True)
    r, x, y, z = q.unbind(1)
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).view(-1, 3, 3)


class GaussianModel:
    def __init__(self, sh_degree: int):
        self.active_sh_degree = 0
        self.max_sh_degree = sh_degree
        self._t = {k: torch.empty(0) for k in TABLE}
        self._acts = ActivationCache()
        self.max_radii2D = torch.empty(0)
        self.xyz_gradient_accum = torch.empty(0)
        self.denom = torch.empty(0)
        self.optimizer = None
        self.percent_dense = 0.0
        self.spatial_lr_scale = 0.0
        self.freeze_means = self.freeze_scales = self.freeze_rotations = self.freeze_opacities = False

    # ---- the reference's attribute names -------------------------------------------------------------------
    packed_features = True                                   # _features_dc / _features_rest are views of _features
    _xyz = property(lambda s: s._t["xyz"])
    _features = property(lambda s: s._t["features"])
    _features_dc = property(lambda s: s._t["features"][:, :1])
    _features_rest = property(lambda s: s._t["features"][:, 1:])
    _opacity = property(lambda s: s._t["opacity"])
    _scaling = property(lambda s: s._t["scaling"])
    _rotation = property(lambda s: s._t["rotation"])

    @property
    def get_xyz(self):
        return self._xyz.detach() if self.freeze_means else self._xyz

    @property
    def get_scaling(self):
        s = self._acts.get(0, self._scaling, self._rotation, self._opacity)
        return s.detach() if self.freeze_scales else s

    @property
    def get_rotation(self):
        r = self._acts.get(1, self._scaling, self._rotation, self._opacity)
        return r.detach() if self.freeze_rotations else r

    @property
    def get_opacity(self):
        o = self._acts.get(2, self._scaling, self._rotation, self._opacity)
        return o.detach() if self.freeze_opacities else o

    @property
    def get_features(self):
        return self._t["features"]                           # = torch.cat((_features_dc, _features_rest), dim=1), see the header

    def get_covariance(self, scaling_modifier=1):
        L = quat_to_rotmat(self._rotation) @ torch.diag_embed(self.get_scaling * scaling_modifier)
        S = L @ L.transpose(1, 2)
        return torch.stack([S[:, 0, 0], S[:, 0, 1], S[:, 0, 2], S[:, 1, 1], S[:, 1, 2], S[:, 2, 2]], 1)

    def oneupSHdegree(self):
        self.active_sh_degree = min(self.active_sh_degree + 1, self.max_sh_degree)

    @property
    def device(self):
        return self._xyz.device

    # ---- initialisation ------------------------------------------------------------------------------------
    def _adopt(self, tensors: dict):
        tensors = _pack(tensors)
        self._t = {k: nn.Parameter(tensors[k].detach().clone().contiguous().requires_grad_(True)) for k in TABLE}
        self._acts.invalidate()
        P = self._xyz.shape[0]
        self.max_radii2D = torch.zeros(P, device=self.device)

    def create_from_pcd(self, pcd, spatial_lr_scale: float, device="cuda"):
        """pcd: anything with `.points` [P,3] and `.colors` [P,3] in [0,1] (utils/graphics_utils.py:17-20)."""
        from simple_knn._C import distCUDA2
        self.spatial_lr_scale = spatial_lr_scale
        pts = torch.as_tensor(np.asarray(pcd.points), dtype=torch.float32, device=device)
        rgb = torch.as_tensor(np.asarray(pcd.colors), dtype=torch.float32, device=device)
        P, M = pts.shape[0], (self.max_sh_degree + 1) ** 2
        sh = torch.zeros(P, M, 3, device=device)
        sh[:, 0, :] = (rgb - 0.5) / SH_C0
        d2 = torch.clamp_min(distCUDA2(pts), 1e-7)
        rot = torch.zeros(P, 4, device=device)
        rot[:, 0] = 1
        self._adopt(dict(xyz=pts, f_dc=sh[:, :1], f_rest=sh[:, 1:], scaling=torch.log(torch.sqrt(d2))[:, None].repeat(1, 3),
                         rotation=rot, opacity=inverse_sigmoid(torch.full((P, 1), 0.1, device=device))))

    def adopt_scene(self, scene, device="cuda"):
        """Start from a scene_synth.Scene (raw parameters) instead of a point cloud."""
        self.spatial_lr_scale = 1.0
        self.active_sh_degree = scene.sh_degree
        self._adopt({k: v.to(device) for k, v in dict(
            xyz=scene.means3D, f_dc=scene.shs[:, :1], f_rest=scene.shs[:, 1:], scaling=scene.log_scales,
            rotation=scene.raw_rotations, opacity=scene.opacity_logits).items()})

    def training_setup(self, opt):
        P = self._xyz.shape[0]
        self.percent_dense = opt.percent_dense
        self.xyz_gradient_accum = torch.zeros(P, 1, device=self.device)
        self.denom = torch.zeros(P, 1, device=self.device)
        lrs = dict(xyz=opt.position_lr_init * self.spatial_lr_scale, f_dc=opt.feature_lr, f_rest=opt.feature_lr / 20.0,
                   opacity=opt.opacity_lr, scaling=opt.scaling_lr, rotation=opt.rotation_lr)
        # the reference's six groups, in its order; "f_dc" carries the interleaved SH table and takes the lr of its columns
        # 1.. from the parameter-less group "f_rest" (see the header)
        groups = []
        for k in GROUPS:
            if k == "f_dc":
                groups.append({"params": [self._t["features"]], "lr": lrs[k], "name": k, "head_cols": 1, "tail": "f_rest"})
            elif k == "f_rest":
                groups.append({"params": [], "lr": lrs[k], "name": k})
            else:
                groups.append({"params": [self._t[k]], "lr": lrs[k], "name": k})
        from fused_adam import FusedAdam              # torch.optim.Adam's arithmetic and state layout; one HIP kernel per tensor
        self.optimizer = FusedAdam(groups, lr=0.0, eps=1e-15, native=getattr(opt, "fused_adam", True) and self.device.type == "cuda")
        self._xyz_lr = expon_lr(opt.position_lr_init * self.spatial_lr_scale, opt.position_lr_final * self.spatial_lr_scale,
                                lr_delay_mult=opt.position_lr_delay_mult, max_steps=opt.position_lr_max_steps)

    def update_learning_rate(self, iteration):
        for g in self.optimizer.param_groups:
            if g["name"] == "xyz":
                g["lr"] = self._xyz_lr(iteration)
                return g["lr"]

    # ---- one primitive for every structural edit -------------------------------------------------------------
    def _rebuild(self, keep=None, extra=None, replace=None):
        """New parameter table = old rows selected by boolean `keep` (all if None), then `extra` rows appended
        (`extra` names its tensors as the table does, or as the reference does with f_dc / f_rest);
        `replace` = {name: tensor} swaps a whole tensor (fresh Adam moments).  Adam moments follow their rows;
        appended rows start with zero moments (the reference's _prune_optimizer / cat_tensors_to_optimizer /
        replace_tensor_to_optimizer in one pass)."""
        if extra is not None:
            extra = _pack(extra)
        done = set()
        for grp in self.optimizer.param_groups if self.optimizer is not None else []:
            if not grp["params"]:
                continue                                      # "f_rest": its columns live in the "f_dc" group's tensor
            name, old = ("features" if grp["name"] == "f_dc" else grp["name"]), grp["params"][0]
            state = self.optimizer.state.pop(old, None)
            if replace is not None and name in replace:
                new = replace[name]
                if state is not None:
                    state["exp_avg"], state["exp_avg_sq"] = torch.zeros_like(new), torch.zeros_like(new)
            else:
                rows = old.detach() if keep is None else old.detach()[keep]
                add = None if extra is None else extra[name]
                new = rows if add is None else torch.cat((rows, add), 0)
                if state is not None:
                    for m in ("exp_avg", "exp_avg_sq"):
                        kept = state[m] if keep is None else state[m][keep]
                        state[m] = kept if add is None else torch.cat((kept, torch.zeros_like(add)), 0)
            new = nn.Parameter(new.contiguous().requires_grad_(True))
            grp["params"][0] = new
            if state is not None:
                self.optimizer.state[new] = state
            self._t[name] = new
            done.add(name)
        for name in TABLE:
            if name in done:
                continue
            if replace is not None and name in replace:
                rows = replace[name]
            else:
                rows = self._t[name].detach() if keep is None else self._t[name].detach()[keep]
                if extra is not None:
                    rows = torch.cat((rows, extra[name]), 0)
            self._t[name] = nn.Parameter(rows.contiguous().requires_grad_(True))
        self._acts.invalidate()

    def _reset_stats(self):
        P = self._xyz.shape[0]
        self.xyz_gradient_accum = torch.zeros(P, 1, device=self.device)
        self.denom = torch.zeros(P, 1, device=self.device)
        self.max_radii2D = torch.zeros(P, device=self.device)

    def prune_points(self, mask):
        keep = ~mask
        self._rebuild(keep=keep)
        self.xyz_gradient_accum, self.denom, self.max_radii2D = self.xyz_gradient_accum[keep], self.denom[keep], self.max_radii2D[keep]

    def reset_opacity(self):
        capped = inverse_sigmoid(torch.min(self.get_opacity.detach(), torch.full_like(self._opacity, 0.01)))
        self._rebuild(replace={"opacity": capped})

    def add_densification_stats(self, viewspace_point_tensor, update_filter):
        self.xyz_gradient_accum[update_filter] += torch.norm(viewspace_point_tensor.grad[update_filter, :2], dim=-1, keepdim=True)
        self.denom[update_filter] += 1

    def update_densification_stats(self, viewspace_point_tensor, radii):
        """train.py:127-130 in one native pass (gsr_densify_stats): max_radii2D and add_densification_stats for the
        visible set `radii > 0`, without the four boolean-mask compactions (each a host synchronisation)."""
        grad = viewspace_point_tensor.grad
        if grad is None or not self.max_radii2D.is_cuda:
            vis = radii > 0
            self.max_radii2D[vis] = torch.max(self.max_radii2D[vis], radii[vis])
            self.add_densification_stats(viewspace_point_tensor, vis)
            return
        from diff_gaussian_rasterization import _native
        with torch.cuda.device(radii.device):
            _native.densify_stats(radii.contiguous(), grad.contiguous(), self.max_radii2D, self.xyz_gradient_accum, self.denom)

    def densify_and_prune(self, max_grad, min_opacity, extent, max_screen_size):
        grads = self.xyz_gradient_accum / self.denom
        grads[grads.isnan()] = 0.0
        big = torch.max(self.get_scaling.detach(), dim=1).values > self.percent_dense * extent
        hot = torch.norm(grads, dim=-1) >= max_grad
        # clone: small Gaussians with a large view-space gradient are duplicated in place
        sel = hot & ~big
        self._rebuild(extra={k: self._t[k].detach()[sel] for k in TABLE})
        n_after_clone = self._xyz.shape[0]
        self._reset_stats()
        # split: large ones are replaced by N = 2 samples from themselves, 1.6x smaller
        N = 2
        sel = torch.zeros(n_after_clone, dtype=torch.bool, device=self.device)
        sel[:grads.shape[0]] = hot & big
        scale_sel = self.get_scaling.detach()[sel].repeat(N, 1)
        offs = torch.normal(mean=torch.zeros_like(scale_sel), std=scale_sel)
        R = quat_to_rotmat(self._rotation.detach()[sel]).repeat(N, 1, 1)
        new = {k: self._t[k].detach()[sel].repeat(N, *([1] * (self._t[k].dim() - 1))) for k in TABLE}
        new["xyz"] = torch.bmm(R, offs.unsqueeze(-1)).squeeze(-1) + self._xyz.detach()[sel].repeat(N, 1)
        new["scaling"] = torch.log(scale_sel / (0.8 * N))
        self._rebuild(extra=new)
        self._reset_stats()
        self.prune_points(torch.cat((sel, torch.zeros(N * int(sel.sum()), dtype=torch.bool, device=self.device))))
        # prune: transparent, or too large on screen / in the world
        drop = (self.get_opacity.detach() < min_opacity).squeeze(-1)
        if max_screen_size:
            drop = drop | (self.max_radii2D > max_screen_size) | (self.get_scaling.detach().max(dim=1).values > 0.1 * extent)
        self.prune_points(drop)

    # ---- persistence -----------------------------------------------------------------------------------------
    # The checkpoint tuple is the reference's (scene/gaussian_model.py:67-99, saved by train.py as chkpnt*.pth), its optimizer
    # state_dict included: six groups with one tensor each (params 0..5 = xyz, f_dc, f_rest, opacity, scaling, rotation), every
    # group carrying torch.optim.Adam's own keys.  In memory f_dc and f_rest share one table; the two layouts are converted here,
    # at the boundary, so that checkpoints move both ways between this class and the reference's.
    def _optimizer_state_reference_layout(self):
        sd = self.optimizer.state_dict()
        defaults = {k: v for k, v in torch.optim.Adam([torch.zeros(1)], lr=0.0, eps=1e-15).param_groups[0].items() if k != "params"}
        groups, state, new_id = [], {}, 0
        for g in sd["param_groups"]:
            hyper = {**defaults, **{k: v for k, v in g.items() if k not in ("params", "head_cols", "tail")}}
            if g.get("name") == "f_rest":
                continue                                         # emitted together with "f_dc" below
            st = sd["state"].get(g["params"][0]) if g["params"] else None
            if g.get("name") == "f_dc":
                rest = next(x for x in sd["param_groups"] if x.get("name") == "f_rest")
                for hyp, cols in ((hyper, slice(0, 1)), ({**defaults, **{k: v for k, v in rest.items() if k != "params"}}, slice(1, None))):
                    groups.append({**hyp, "params": [new_id]})
                    if st is not None:
                        state[new_id] = {"step": st["step"].clone(), "exp_avg": st["exp_avg"][:, cols].contiguous(),      # (two tensors: own counters)
                                         "exp_avg_sq": st["exp_avg_sq"][:, cols].contiguous()}
                    new_id += 1
                continue
            groups.append({**hyper, "params": [new_id]})
            if st is not None:
                state[new_id] = st
            new_id += 1
        return {"state": state, "param_groups": groups}

    def _load_optimizer_state(self, sd):
        """Accepts the reference's six-tensor layout (what capture() emits, what the reference's own checkpoints hold) and this
        class's packed one (five tensors; checkpoints written by earlier versions of this package)."""
        groups = sd["param_groups"]
        mine = self.optimizer.state_dict()["param_groups"]
        if len(groups) == 6 and all(len(g["params"]) == 1 for g in groups):
            by_name = {g.get("name", GROUPS[i]): g for i, g in enumerate(groups)}
            state, new_groups = {}, []
            for tmpl in mine:
                src = by_name[tmpl["name"]]
                new_groups.append({**{k: v for k, v in src.items() if k != "params"}, **{k: tmpl[k] for k in ("head_cols", "tail") if k in tmpl},
                                   "params": list(tmpl["params"])})
                if not tmpl["params"]:
                    continue
                if tmpl["name"] == "f_dc":
                    dc, rest = sd["state"].get(by_name["f_dc"]["params"][0]), sd["state"].get(by_name["f_rest"]["params"][0])
                    if dc is not None:
                        cat = lambda k: dc[k] if rest is None or rest[k].numel() == 0 else torch.cat((dc[k], rest[k]), dim=1)
                        state[tmpl["params"][0]] = {"step": dc["step"], "exp_avg": cat("exp_avg"), "exp_avg_sq": cat("exp_avg_sq")}
                else:
                    st = sd["state"].get(src["params"][0])
                    if st is not None:
                        state[tmpl["params"][0]] = st
            sd = {"state": state, "param_groups": new_groups}
        self.optimizer.load_state_dict(sd)

    def capture(self):
        return (self.active_sh_degree, self._xyz, self._features_dc, self._features_rest, self._scaling, self._rotation,
                self._opacity, self.max_radii2D, self.xyz_gradient_accum, self.denom, self._optimizer_state_reference_layout(),
                self.spatial_lr_scale)

    def restore(self, model_args, training_args):
        (self.active_sh_degree, xyz, fdc, frest, scaling, rotation, opacity, max_radii, accum, denom, opt_state,
         self.spatial_lr_scale) = model_args
        self._adopt(dict(xyz=xyz, f_dc=fdc, f_rest=frest, scaling=scaling, rotation=rotation, opacity=opacity))
        self.max_radii2D = max_radii
        self.training_setup(training_args)
        self.xyz_gradient_accum, self.denom = accum, denom
        self._load_optimizer_state(opt_state)

    def save_ply(self, path):
        os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
        n = lambda t: t.detach().cpu().numpy()
        ply_io.write_gaussian_ply(path, n(self._xyz), n(self._features_dc), n(self._features_rest), n(self._opacity),
                                  n(self._scaling), n(self._rotation))

    def load_ply(self, path, device="cuda"):
        d = ply_io.read_gaussian_ply(path)
        assert d["f_rest"].shape[1] == (self.max_sh_degree + 1) ** 2 - 1, "PLY SH degree does not match the model"
        t = lambda a: torch.tensor(a, dtype=torch.float32, device=device)
        self._adopt(dict(xyz=t(d["xyz"]), f_dc=t(d["f_dc"]), f_rest=t(d["f_rest"]), opacity=t(d["opacity"]),
                         scaling=t(d["scaling"]), rotation=t(d["rotation"])))
        self.active_sh_degree = self.max_sh_degree
