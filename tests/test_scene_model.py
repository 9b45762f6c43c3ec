"""CPU tests of the f1/f4 host mirrors: PLY wire format, LR schedule vs the reference's golden values, and the
parameter-store surgery (clone / split / prune / opacity reset keep parameters and Adam moments aligned)."""
import os

import numpy as np
import torch

import scene_synth as S
from scene import GaussianModel, OptimizationDefaults
from scene import ply_io
from scene.gaussian_model import expon_lr

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_lr_schedule_matches_reference():
    gold = np.load(os.path.join(G, "lr.npz"))
    f = expon_lr(0.00016, 0.0000016, lr_delay_mult=0.01, max_steps=30000)
    np.testing.assert_allclose([f(int(s)) for s in gold["steps"]], gold["lr"], rtol=1e-12)


def test_optimization_defaults_match_reference():
    import json
    ref = json.load(open(os.path.join(G, "params.json")))["optimization"]
    d = OptimizationDefaults()
    for k, v in ref.items():
        if hasattr(d, k) and k != "iterations":
            assert getattr(d, k) == v, k


def _model(P=500, D=3, device="cpu"):
    gm = GaussianModel(D)
    gm.adopt_scene(S.make_scene(P, 64, 64, D, 7), device=device)
    gm.training_setup(OptimizationDefaults())
    return gm


def test_ply_round_trip_and_layout(tmp_path):
    gm = _model()
    path = str(tmp_path / "pc" / "point_cloud.ply")
    gm.save_ply(path)
    head = open(path, "rb").read(2000).split(b"end_header")[0].decode()
    props = [l.split()[2] for l in head.splitlines() if l.startswith("property")]
    assert props == ply_io.property_names(45) and props[:9] == ["x", "y", "z", "nx", "ny", "nz", "f_dc_0", "f_dc_1", "f_dc_2"]
    assert "format binary_little_endian 1.0" in head and "element vertex 500" in head
    raw = np.frombuffer(open(path, "rb").read().split(b"end_header\n", 1)[1], "<f4").reshape(500, -1)
    # f_rest is channel-major: f_rest_0..14 = red coefficients 1..15
    np.testing.assert_array_equal(raw[:, 9:9 + 15], gm._features_rest.detach().numpy()[:, :, 0])
    gm2 = GaussianModel(3)
    gm2.load_ply(path, device="cpu")
    for k in ("_xyz", "_features_dc", "_features_rest", "_opacity", "_scaling", "_rotation"):
        assert torch.equal(getattr(gm, k).detach(), getattr(gm2, k).detach()), k
    assert gm2.active_sh_degree == 3


def test_structural_edits_keep_adam_state_aligned():
    torch.manual_seed(0)
    gm = _model(400)
    groups = [g for g in gm.optimizer.param_groups if g["params"]]     # "f_rest" holds no tensor: its columns live in "f_dc"
    assert [g["name"] for g in gm.optimizer.param_groups] == ["xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"]
    table = lambda g: gm._t["features" if g["name"] == "f_dc" else g["name"]]
    for p in groups:                                         # one Adam step so that moments exist
        p["params"][0].grad = torch.randn_like(p["params"][0])
    gm.optimizer.step()
    tag = gm._xyz.detach().clone()
    m_before = gm.optimizer.state[gm._xyz]["exp_avg"].clone()
    # prune
    mask = torch.zeros(400, dtype=torch.bool); mask[::4] = True
    gm.prune_points(mask)
    assert gm._xyz.shape[0] == 300 and gm.denom.shape[0] == 300 and gm.max_radii2D.shape[0] == 300
    assert torch.equal(gm._xyz.detach(), tag[~mask])
    assert torch.equal(gm.optimizer.state[gm._xyz]["exp_avg"], m_before[~mask])
    for g in groups:
        p = g["params"][0]
        assert p.shape[0] == 300 and gm.optimizer.state[p]["exp_avg_sq"].shape == p.shape and p is table(g)
    # opacity reset: opacities capped at 0.01, fresh moments
    gm.reset_opacity()
    assert float(gm.get_opacity.detach().max()) <= 0.01 + 1e-6
    assert float(gm.optimizer.state[gm._opacity]["exp_avg"].abs().max()) == 0.0
    # densify: clone small-hot, split big-hot (2 children replace the parent), prune transparent
    gm.xyz_gradient_accum[:] = 0.0; gm.denom[:] = 1.0
    gm.xyz_gradient_accum[:40] = 1.0                         # 40 "hot" Gaussians
    with torch.no_grad():
        gm._t["scaling"][:20] = np.log(1.0)                  # 20 of them big (> percent_dense * extent = 0.05)
        gm._t["opacity"][:] = 2.0                            # nothing is transparent
        gm._t["opacity"][100:110] = -10.0                    # ... except these ten
    n0 = gm._xyz.shape[0]
    gm.densify_and_prune(0.0002, 0.005, extent=5.0, max_screen_size=None)
    assert gm._xyz.shape[0] == n0 + 20 + 2 * 20 - 20 - 10    # +clones, +children, -parents, -transparent
    for g in groups:
        p = g["params"][0]
        assert gm.optimizer.state[p]["exp_avg"].shape == p.shape and p is table(g)
    assert gm.xyz_gradient_accum.shape[0] == gm._xyz.shape[0] == gm.max_radii2D.shape[0]
    # capture / restore
    snap = gm.capture()
    gm3 = GaussianModel(3)
    gm3.restore(snap, OptimizationDefaults())
    assert torch.equal(gm3._xyz.detach(), gm._xyz.detach()) and gm3.active_sh_degree == gm.active_sh_degree


def _reference_style_adam(tensors):
    """torch.optim.Adam built the way the reference's training_setup builds it (scene/gaussian_model.py:159-177)."""
    names = ("xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation")
    return torch.optim.Adam([{"params": [tensors[n]], "lr": 1e-3 * (i + 1), "name": n} for i, n in enumerate(names)], lr=0.0, eps=1e-15)


def test_checkpoint_optimizer_state_is_the_reference_layout_both_ways():
    """capture() emits the reference's six-tensor Adam state_dict (chkpnt*.pth of train.py), restore() reads it: a state saved by a
    torch.optim.Adam built the reference way loads into this class, and this class's captured state loads into such an optimizer."""
    torch.manual_seed(3)
    gm = _model(200)
    for g in gm.optimizer.param_groups:
        for p in g["params"]:
            p.grad = torch.randn_like(p)
    gm.optimizer.step()
    snap = gm.capture()
    sd = snap[10]
    assert [g["name"] for g in sd["param_groups"]] == ["xyz", "f_dc", "f_rest", "opacity", "scaling", "rotation"]
    assert [g["params"] for g in sd["param_groups"]] == [[0], [1], [2], [3], [4], [5]]
    assert sd["state"][1]["exp_avg"].shape == (200, 1, 3) and sd["state"][2]["exp_avg"].shape == (200, 15, 3)
    # (a) this class's captured state -> an optimizer built the reference way
    ref_t = {n: torch.nn.Parameter(t.detach().clone()) for n, t in (("xyz", gm._xyz), ("f_dc", gm._features_dc.contiguous()),
             ("f_rest", gm._features_rest.contiguous()), ("opacity", gm._opacity), ("scaling", gm._scaling), ("rotation", gm._rotation))}
    ref = _reference_style_adam(ref_t)
    ref.load_state_dict(sd)
    feats = gm.optimizer.state[gm._features]
    assert torch.equal(ref.state[ref_t["f_dc"]]["exp_avg"], feats["exp_avg"][:, :1])
    assert torch.equal(ref.state[ref_t["f_rest"]]["exp_avg_sq"], feats["exp_avg_sq"][:, 1:])
    assert torch.equal(ref.state[ref_t["rotation"]]["exp_avg"], gm.optimizer.state[gm._rotation]["exp_avg"])
    for p in ref_t.values():                                  # and it steps (every Adam key is there)
        p.grad = torch.randn_like(p)
    ref.step()
    # (b) a state_dict written by the reference-style optimizer -> this class
    ref.param_groups[2]["lr"], ref.param_groups[5]["lr"] = 3e-3, 6e-3
    snap_ref = snap[:10] + (ref.state_dict(),) + snap[11:]
    gm2 = GaussianModel(3)
    gm2.restore(snap_ref, OptimizationDefaults())
    f2 = gm2.optimizer.state[gm2._features]
    assert torch.equal(f2["exp_avg"][:, :1], ref.state[ref_t["f_dc"]]["exp_avg"])
    assert torch.equal(f2["exp_avg_sq"][:, 1:], ref.state[ref_t["f_rest"]]["exp_avg_sq"])
    assert float(f2["step"]) == float(ref.state[ref_t["f_dc"]]["step"]) == 2.0
    assert torch.equal(gm2.optimizer.state[gm2._opacity]["exp_avg"], ref.state[ref_t["opacity"]]["exp_avg"])
    lrs = {g["name"]: g["lr"] for g in gm2.optimizer.param_groups}
    assert lrs["f_rest"] == 3e-3 and lrs["rotation"] == 6e-3      # hyper-parameters come from the loaded state, as torch's do
    for g in gm2.optimizer.param_groups:                      # and the restored model keeps stepping
        for p in g["params"]:
            p.grad = torch.randn_like(p)
    gm2.optimizer.step()
    # (c) the packed layout of earlier versions of this package still loads
    gm3 = GaussianModel(3)
    gm3.restore(snap[:10] + (gm.optimizer.state_dict(),) + snap[11:], OptimizationDefaults())
    assert torch.equal(gm3.optimizer.state[gm3._features]["exp_avg"], feats["exp_avg"])


def test_packed_features_behave_as_the_reference_two_tensors():
    """get_features is the interleaved table itself (= torch.cat of its two column ranges, scene/gaussian_model.py:118-121);
    _features_dc / _features_rest are views with the reference's shapes, and gradients through either reach the table."""
    gm = _model(64)
    F = gm.get_features
    assert F is gm._features and F.shape == (64, 16, 3) and F.is_leaf and F.requires_grad
    assert gm._features_dc.shape == (64, 1, 3) and gm._features_rest.shape == (64, 15, 3)
    assert torch.equal(torch.cat((gm._features_dc, gm._features_rest), dim=1), F)
    w = torch.randn(64, 16, 3)
    (torch.cat((gm._features_dc, gm._features_rest), dim=1) * w).sum().backward()
    assert torch.equal(F.grad, w)


def test_adam_on_the_packed_table_equals_torch_adam_on_two_tensors():
    """One optimizer step of the model (FusedAdam's torch arithmetic on host tensors; groups "f_dc" lr and "f_rest" lr
    over one table) against torch.optim.Adam over the reference's six separate tensors (scene/gaussian_model.py:159-177)."""
    torch.manual_seed(1)
    gm = _model(128)
    opt = OptimizationDefaults()
    ref = {k: v.detach().clone().requires_grad_(True) for k, v in dict(
        xyz=gm._xyz, f_dc=gm._features_dc, f_rest=gm._features_rest, opacity=gm._opacity, scaling=gm._scaling,
        rotation=gm._rotation).items()}
    lrs = dict(xyz=opt.position_lr_init * gm.spatial_lr_scale, f_dc=opt.feature_lr, f_rest=opt.feature_lr / 20.0,
               opacity=opt.opacity_lr, scaling=opt.scaling_lr, rotation=opt.rotation_lr)
    ref_opt = torch.optim.Adam([{"params": [ref[k]], "lr": lrs[k], "name": k} for k in ref], lr=0.0, eps=1e-15)
    for it in range(3):
        grads = {k: torch.randn_like(v) * 10.0 ** float(torch.randint(-6, 1, ())) for k, v in ref.items()}
        for k, v in ref.items():
            v.grad = grads[k].clone()
        for k in ("xyz", "opacity", "scaling", "rotation"):
            gm._t[k].grad = grads[k].clone()
        gm._features.grad = torch.cat((grads["f_dc"], grads["f_rest"]), dim=1)
        ref_opt.step(); gm.optimizer.step()
        for k, attr in (("xyz", "_xyz"), ("f_dc", "_features_dc"), ("f_rest", "_features_rest"), ("opacity", "_opacity"),
                        ("scaling", "_scaling"), ("rotation", "_rotation")):
            assert torch.equal(getattr(gm, attr).detach(), ref[k].detach()), (it, k)
    m = gm.optimizer.state[gm._features]["exp_avg"]
    assert torch.equal(m[:, :1], ref_opt.state[ref["f_dc"]]["exp_avg"]) and torch.equal(m[:, 1:], ref_opt.state[ref["f_rest"]]["exp_avg"])


def test_getters_on_host_tensors_are_the_reference_torch_ops():
    gm = _model(50)
    assert torch.equal(gm.get_scaling, torch.exp(gm._scaling)) and torch.equal(gm.get_opacity, torch.sigmoid(gm._opacity))
    assert torch.equal(gm.get_rotation, torch.nn.functional.normalize(gm._rotation))
    gm.freeze_scales = True
    assert not gm.get_scaling.requires_grad and gm.get_rotation.requires_grad
