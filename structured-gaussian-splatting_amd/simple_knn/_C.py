"""`simple_knn._C` drop-in: `distCUDA2(points[P,3]) -> float32[P]`, the mean squared distance of every point
to its 3 nearest other points (call sites scene/gaussian_model.py:144-145,
scene/latent_gaussian_model.py:219-220).  Hand-written HIP in csrc/gsr_knn.hip; no CPU path."""
import torch


def distCUDA2(points: torch.Tensor) -> torch.Tensor:
    if points.device.type != "cuda":
        raise RuntimeError("simple_knn._C.distCUDA2 (MI355X build) needs a tensor on a HIP device; there is no CPU path")
    from diff_gaussian_rasterization import _native
    return _native.dist2_knn3(points)
