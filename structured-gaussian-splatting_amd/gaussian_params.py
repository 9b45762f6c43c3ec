"""Minimal parameter store with the getters render() reads from the reference's GaussianModel
(scene/gaussian_model.py:101-128): raw leaf Parameters + the same activations (exp / normalize / sigmoid /
cat).  Used by bench.py and the tests; densification and the optimizer live in the SURVEY 8f "next" rows."""
import torch
from torch import nn


class GaussianParams(nn.Module):
    def __init__(self, scene, max_sh_degree=None):
        super().__init__()
        self.max_sh_degree = scene.sh_degree if max_sh_degree is None else max_sh_degree
        self.active_sh_degree = scene.sh_degree
        self._xyz = nn.Parameter(scene.means3D.clone())
        self._features_dc = nn.Parameter(scene.shs[:, :1, :].clone().contiguous())
        self._features_rest = nn.Parameter(scene.shs[:, 1:, :].clone().contiguous())
        self._scaling = nn.Parameter(scene.log_scales.clone())
        self._rotation = nn.Parameter(scene.raw_rotations.clone())
        self._opacity = nn.Parameter(scene.opacity_logits.clone())

    @property
    def get_xyz(self):
        return self._xyz

    @property
    def get_scaling(self):
        return torch.exp(self._scaling)

    @property
    def get_rotation(self):
        return torch.nn.functional.normalize(self._rotation)

    @property
    def get_opacity(self):
        return torch.sigmoid(self._opacity)

    @property
    def get_features(self):
        return torch.cat((self._features_dc, self._features_rest), dim=1)

    def get_covariance(self, scaling_modifier=1.0):
        s = self.get_scaling * scaling_modifier
        q = self.get_rotation
        r, x, y, z = q.unbind(1)
        R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                         2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                         2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], 1).view(-1, 3, 3)
        L = R @ torch.diag_embed(s)
        Sg = L @ L.transpose(1, 2)
        return torch.stack([Sg[:, 0, 0], Sg[:, 0, 1], Sg[:, 0, 2], Sg[:, 1, 1], Sg[:, 1, 2], Sg[:, 2, 2]], 1)


class Pipe:
    """PipelineParams defaults (arguments/__init__.py:68-74)."""
    convert_SHs_python = False
    compute_cov3D_python = False
    debug = False
    fused_activations = None       # extension, see gaussian_renderer.render: None = only for this package's scene.GaussianModel
