"""CPU tests of host-side mirrors against the reference's golden vectors (tests/golden)."""
import json
import os

import numpy as np
import torch

import loss_utils
import scene_synth as S

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_loss_matches_reference():
    gold = np.load(os.path.join(G, "loss.npz"))
    for i in range(2):
        a = torch.tensor(gold[f"a{i}"], requires_grad=True)
        b = torch.tensor(gold[f"b{i}"])
        l1, s = loss_utils.l1_loss(a, b), loss_utils.ssim(a, b)
        loss = loss_utils.training_loss(a, b)
        loss.backward()
        assert abs(l1.item() - float(gold[f"l1_{i}"])) < 1e-7
        assert abs(s.item() - float(gold[f"ssim_{i}"])) < 1e-6
        assert abs(loss.item() - float(gold[f"loss_{i}"])) < 1e-6
        np.testing.assert_allclose(a.grad.numpy(), gold[f"grad_a{i}"], atol=1e-8, rtol=1e-4)
        np.testing.assert_allclose(loss_utils.psnr(a.detach(), b).numpy(), gold[f"psnr_{i}"], rtol=1e-6)


def test_render_python_sh_matches_reference_eval_sh():
    from gaussian_renderer import eval_sh
    gold = np.load(os.path.join(G, "sh_eval.npz"))
    pos, sh, cam = torch.tensor(gold["pos"]), torch.tensor(gold["sh"]), torch.tensor(gold["campos"])
    d = pos - cam
    d = d / d.norm(dim=1, keepdim=True)
    for D in range(4):
        raw = eval_sh(D, sh.transpose(1, 2), d)
        np.testing.assert_allclose(raw.numpy(), gold[f"raw_D{D}"], atol=1e-13)


def test_gaussian_params_getters_and_defaults():
    from gaussian_params import GaussianParams, Pipe
    scene = S.make_scene(100, 64, 64, 3, 5)
    gp = GaussianParams(scene)
    a = scene.activated()
    assert torch.equal(gp.get_scaling, a["scales"]) and torch.equal(gp.get_rotation, a["rotations"])
    assert torch.equal(gp.get_opacity, a["opacities"]) and torch.equal(gp.get_features, a["shs"])
    params = json.load(open(os.path.join(G, "params.json")))
    for k in ("convert_SHs_python", "compute_cov3D_python", "debug"):
        assert getattr(Pipe, k) == params["pipeline"][k]
    assert params["optimization"]["lambda_dssim"] == 0.2


def test_getter_fusion_matcher_accepts_only_the_reference_getters():
    """diff_gaussian_rasterization._match_getters (opt-in FUSE_GETTERS): the autograd-history patterns of the reference's
    GaussianModel getters are recognised exactly; look-alikes are refused (then the plain path runs)."""
    import torch
    import torch.nn.functional as F
    import diff_gaussian_rasterization as d
    x = torch.randn(7, 4, requires_grad=True)
    s = torch.randn(7, 3, requires_grad=True)
    o = torch.randn(7, 1, requires_grad=True)
    assert d._normalize_getter(F.normalize(x)) is x
    assert d._normalize_getter(x / x.norm(dim=1, keepdim=True)) is None              # no clamp_min: not F.normalize
    assert d._normalize_getter(F.normalize(x, dim=0)) is None and d._normalize_getter(F.normalize(x, eps=1e-6)) is None
    assert d._normalize_getter(F.normalize(x * 1.0)) is None                           # not a leaf behind it
    assert d._unary_getter(torch.exp(s), "ExpBackward0") is s and d._unary_getter(torch.sigmoid(o), "SigmoidBackward0") is o
    assert d._unary_getter(torch.exp(s) * 2.0, "ExpBackward0") is None and d._unary_getter(torch.exp(s.detach()), "ExpBackward0") is None
    assert d._unary_getter(torch.sigmoid(o)[:5], "SigmoidBackward0") is None
    # the whole matcher needs device tensors for means3D; on the CPU it declines
    dc, rest = torch.randn(7, 1, 3, requires_grad=True), torch.randn(7, 15, 3, requires_grad=True)
    m = torch.randn(7, 3, requires_grad=True)
    assert d._match_getters(m, torch.sigmoid(o), torch.cat((dc, rest), 1), torch.exp(s), F.normalize(x)) is None
