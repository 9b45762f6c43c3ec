"""Host-side mirrors of the reference's scene package that sit on the rasterizer's path:
the Gaussian parameter store with densification (scene/gaussian_model.py) and the PLY wire format."""
from .gaussian_model import GaussianModel, OptimizationDefaults  # noqa: F401
