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
