"""ctypes binding of oracle/libgsr_oracle.so (the plain-C restatement in gsr_oracle.c).

TEST INFRASTRUCTURE ONLY — see oracle/__init__.py.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
FRAGILE_POS_ULPS = 2.0
_LIB = None


def lib_path() -> str:
    return os.path.join(_HERE, "libgsr_oracle.so")


def build(force: bool = False) -> str:
    """Compile the C oracle with gcc (oracle/Makefile).  No-op when up to date."""
    so = lib_path()
    srcs = [os.path.join(_HERE, f) for f in ("gsr_oracle.c", "gsr_oracle_impl.h")]
    srcs.append(os.path.join(_HERE, "..", "include", "gsr_constants.h"))
    if force or not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libgsr_oracle.so"], stdout=subprocess.DEVNULL)
    return so


def _lib():
    global _LIB
    if _LIB is None:
        so = lib_path()
        if not os.path.exists(so):
            build()
        _LIB = C.CDLL(so)
        for suf in ("f32", "f64"):
            getattr(_LIB, f"gso_forward_{suf}").restype = C.c_void_p
            getattr(_LIB, f"gso_num_rendered_{suf}").restype = C.c_int64
            getattr(_LIB, f"gso_num_pairs_{suf}").restype = C.c_int64
    return _LIB


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class OracleFrame:
    """One forward pass of the oracle and everything the backward needs."""

    def __init__(self, handle, suf, dtype, P, D, M, W, H, has_cov, has_colors):
        self._h, self._suf, self.dtype = handle, suf, dtype
        self.P, self.D, self.M, self.W, self.H = P, D, M, W, H
        self.Gx, self.Gy = (W + 15) // 16, (H + 15) // 16
        self.has_cov, self.has_colors = has_cov, has_colors
        lib = _lib()
        self.num_rendered = int(getattr(lib, f"gso_num_rendered_{suf}")(C.c_void_p(handle)))
        self.num_pairs = int(getattr(lib, f"gso_num_pairs_{suf}")(C.c_void_p(handle)))
        N = W * H
        self.color = np.empty((3, H, W), dtype)
        self.final_T = np.empty((H, W), dtype)
        self.n_contrib = np.empty((H, W), np.int32)
        self.fragile_px = np.empty((H, W), np.uint8)
        getattr(lib, f"gso_get_image_{suf}")(C.c_void_p(handle), _ptr(self.color), _ptr(self.final_T),
                                              _ptr(self.n_contrib), _ptr(self.fragile_px))
        Pn = max(P, 1)
        self.radii = np.zeros(Pn, np.int32)
        self.xy = np.zeros((Pn, 2), dtype)
        self.depth = np.zeros(Pn, dtype)
        self.cov3D = np.zeros((Pn, 6), dtype)
        self.conic_opacity = np.zeros((Pn, 4), dtype)
        self.rgb = np.zeros((Pn, 3), dtype)
        self.clamped = np.zeros((Pn, 3), np.uint8)
        self.rect = np.zeros((Pn, 4), np.int32)
        self.tiles_touched = np.zeros(Pn, np.uint32)
        self.fragile_g = np.zeros(Pn, np.uint8)
        getattr(lib, f"gso_get_geom_{suf}")(C.c_void_p(handle), _ptr(self.radii), _ptr(self.xy), _ptr(self.depth),
                                             _ptr(self.cov3D), _ptr(self.conic_opacity), _ptr(self.rgb),
                                             _ptr(self.clamped), _ptr(self.rect), _ptr(self.tiles_touched),
                                             _ptr(self.fragile_g))
        for name in ("radii", "xy", "depth", "cov3D", "conic_opacity", "rgb", "clamped", "rect",
                     "tiles_touched", "fragile_g"):
            setattr(self, name, getattr(self, name)[:P])
        R = self.num_rendered
        self.keys = np.zeros(max(R, 1), np.uint64)
        self.point_list = np.zeros(max(R, 1), np.uint32)
        self.ranges = np.zeros((self.Gx * self.Gy, 2), np.int64)
        getattr(lib, f"gso_get_binning_{suf}")(C.c_void_p(handle), _ptr(self.keys), _ptr(self.point_list),
                                                _ptr(self.ranges))
        self.keys, self.point_list = self.keys[:R], self.point_list[:R]

    def __del__(self):
        if getattr(self, "_h", None):
            getattr(_lib(), f"gso_free_{self._suf}")(C.c_void_p(self._h))
            self._h = None

    # ---- backward (A.9 then A.10) -------------------------------------------------------------
    def backward_screen(self, dL_dcolor: np.ndarray, parallel: bool = False) -> np.ndarray:
        """Per-Gaussian screen-space gradients [P, 9]:
        (dmean2D.x, dmean2D.y, gA, gB, gC, dopacity, drgb[3])."""
        g = np.ascontiguousarray(dL_dcolor, self.dtype)
        assert g.shape == (3, self.H, self.W)
        screen = np.zeros((max(self.P, 1), 9), self.dtype)
        getattr(_lib(), f"gso_backward_screen_{self._suf}")(C.c_void_p(self._h), _ptr(g), C.c_int(int(parallel)),
                                                             _ptr(screen))
        return screen[:self.P]

    def backward_geom(self, screen: np.ndarray, g0: int = 0, g1: Optional[int] = None) -> dict:
        P, M = self.P, self.M
        g1 = P if g1 is None else g1
        Pn = max(P, 1)
        scr = np.zeros((Pn, 9), self.dtype)
        scr[:P] = screen
        out = dict(means3D=np.zeros((Pn, 3), self.dtype), means2D=np.zeros((Pn, 3), self.dtype),
                   shs=np.zeros((Pn, max(M, 1), 3), self.dtype), colors_precomp=np.zeros((Pn, 3), self.dtype),
                   opacities=np.zeros((Pn, 1), self.dtype), scales=np.zeros((Pn, 3), self.dtype),
                   rotations=np.zeros((Pn, 4), self.dtype), cov3D_precomp=np.zeros((Pn, 6), self.dtype))
        getattr(_lib(), f"gso_backward_geom_{self._suf}")(
            C.c_void_p(self._h), _ptr(scr), C.c_int(g0), C.c_int(g1), _ptr(out["means3D"]), _ptr(out["means2D"]),
            _ptr(out["shs"]), _ptr(out["colors_precomp"]), _ptr(out["opacities"]), _ptr(out["scales"]),
            _ptr(out["rotations"]), _ptr(out["cov3D_precomp"]))
        out = {k: v[:P] for k, v in out.items()}
        out["shs"] = out["shs"][:, :M]
        return out

    def backward(self, dL_dcolor: np.ndarray, parallel: bool = False) -> dict:
        screen = self.backward_screen(dL_dcolor, parallel)
        out = self.backward_geom(screen)
        out["screen"] = screen
        return out


def rasterize(*, image_height, image_width, tanfovx, tanfovy, bg, scale_modifier, viewmatrix, projmatrix,
              sh_degree, campos, means3D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
              cov3D_precomp=None, dtype=np.float32, tile_rows=None, fragile_eps=2e-6, fragile_pos=None,
              parallel=False) -> OracleFrame:
    """Forward pass.  Argument names mirror GaussianRasterizationSettings + GaussianRasterizer.forward
    (reference call site gaussian_renderer/__init__.py:36-49, 85-93).  `tile_rows=(ty0, ty1)` restricts
    binning/rendering to a tile-row slab (multi-GPU sharding)."""
    dtype = np.dtype(dtype)
    suf = {np.dtype(np.float32): "f32", np.dtype(np.float64): "f64"}[dtype]

    def arr(a, shape=None):
        if a is None:
            return None
        a = np.ascontiguousarray(np.asarray(a, dtype=dtype))
        if shape is not None:
            a = a.reshape(shape)
        return a

    means3D = arr(means3D)
    P = means3D.shape[0] if means3D.size else 0
    if (shs is None) == (colors_precomp is None):
        raise Exception("Please provide excatly one of either SHs or precomputed colors!")
    if ((scales is None or rotations is None) and cov3D_precomp is None) or \
            ((scales is not None or rotations is not None) and cov3D_precomp is not None):
        raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
    shs = arr(shs)
    M = 0 if shs is None else (shs.shape[1] if shs.ndim == 3 else 0)
    D = int(sh_degree)
    if shs is not None:
        assert (D + 1) ** 2 <= M, "active SH degree exceeds stored coefficients"
    ty0, ty1 = (-1, -1) if tile_rows is None else tile_rows
    if fragile_pos is None:          # pixels: a splat's screen position in binary32 is good to a few ulps of the image size
        fragile_pos = FRAGILE_POS_ULPS * 2.0 ** -24 * max(int(image_width), int(image_height))
    lib = _lib()
    bg_, view_, proj_, cam_ = arr(bg, (3,)), arr(viewmatrix, (16,)), arr(projmatrix, (16,)), arr(campos, (3,))
    op_ = arr(opacities, (P,)) if P else arr(np.zeros(0))
    args = [arr(colors_precomp), arr(scales), arr(rotations), arr(cov3D_precomp)]
    h = getattr(lib, f"gso_forward_{suf}")(
        C.c_int(P), C.c_int(D), C.c_int(M), C.c_int(int(image_width)), C.c_int(int(image_height)),
        C.c_double(float(tanfovx)), C.c_double(float(tanfovy)), C.c_double(float(scale_modifier)),
        C.c_int(ty0), C.c_int(ty1), C.c_double(fragile_eps), C.c_double(fragile_pos), C.c_int(int(parallel)),
        _ptr(bg_), _ptr(view_), _ptr(proj_), _ptr(cam_), _ptr(means3D), _ptr(shs), _ptr(args[0]), _ptr(op_),
        _ptr(args[1]), _ptr(args[2]), _ptr(args[3]))
    return OracleFrame(h, suf, dtype, P, D, M, int(image_width), int(image_height),
                       cov3D_precomp is not None, colors_precomp is not None)


def dist2_knn3(xyz: np.ndarray) -> np.ndarray:
    """Mean squared distance to the 3 nearest other points (A.12; scene/gaussian_model.py:144)."""
    xyz = np.ascontiguousarray(xyz, np.float32)
    out = np.zeros(xyz.shape[0], np.float32)
    _lib().gso_dist2_knn3(C.c_int(xyz.shape[0]), _ptr(xyz), _ptr(out))
    return out


# ---- raw-leaves entry: the parameter store's activations in front of the rasterizer, their chain rule behind it -------------
class OracleRawFrame:
    """`rasterize_raw`'s result: the OracleFrame of the ACTIVATED inputs (binary64 activations of the raw leaves) plus what
    the chain rule back to the raw leaves needs.  `backward()` returns gradients named like the leaves of the reference's
    GaussianModel (scene/gaussian_model.py:47-55): _xyz, _features ([P,M,3]: _features_dc | _features_rest), _opacity,
    _scaling, _rotation — and means2D (the `viewspace_points` gradient, gaussian_renderer/__init__.py:26-30)."""

    def __init__(self, frame: OracleFrame, scales, rotations, opacities, raw_rotations, norm):
        self.frame = frame
        self._scales, self._rot, self._op, self._raw_rot, self._norm = scales, rotations, opacities, raw_rotations, norm

    def __getattr__(self, name):                     # color, radii, fragile_px, xy, Gx, ... : the activated frame's
        return getattr(self.frame, name)

    def backward(self, dL_dcolor: np.ndarray, parallel: bool = False) -> dict:
        act = self.frame.backward(dL_dcolor, parallel=parallel)
        P = self.frame.P
        # get_scaling = exp(_scaling)                      (scene/gaussian_model.py:101-104)
        d_scaling = act["scales"] * self._scales
        # get_opacity = sigmoid(_opacity)                  (:123-125)
        d_opacity = act["opacities"].reshape(P, 1) * (self._op * (1.0 - self._op)).reshape(P, 1)
        # get_rotation = normalize(_rotation) = q / max(|q|, 1e-12)   (:106-109): d/dq = (g - qhat (qhat . g)) / |q|
        g = act["rotations"]
        d_rotation = (g - self._rot * (self._rot * g).sum(1, keepdims=True)) / self._norm
        # get_features = cat(_features_dc, _features_rest) (:116-120): the gradient of the table is dL/dshs itself
        return {"_xyz": act["means3D"], "_features": act["shs"], "_opacity": d_opacity, "_scaling": d_scaling,
                "_rotation": d_rotation, "means2D": act["means2D"], "screen": act["screen"], "activated": act}


def rasterize_raw(*, means3D, features, opacity_logits, log_scales, raw_rotations, dtype=np.float64, **settings) -> OracleRawFrame:
    """The rasterizer as the reference's training step reaches it (train.py:99 -> gaussian_renderer/__init__.py:53-93 with
    `pipe.compute_cov3D_python = pipe.convert_SHs_python = False`): the raw leaves of the parameter store go through the
    getters of scene/gaussian_model.py:101-125 — exp, normalize (torch.nn.functional.normalize: eps 1e-12), sigmoid, cat —
    evaluated in `dtype` (binary64 by default), then through `rasterize`.  `features` is the [P, M, 3] table
    (= cat(_features_dc, _features_rest, dim=1)); `settings` are rasterize's other keyword arguments."""
    dt = np.dtype(dtype)
    raw_s = np.asarray(log_scales, dt)
    raw_q = np.asarray(raw_rotations, dt)
    raw_o = np.asarray(opacity_logits, dt).reshape(-1, 1)
    scales = np.exp(raw_s)
    norm = np.maximum(np.sqrt((raw_q * raw_q).sum(1, keepdims=True)), 1e-12)
    rotations = raw_q / norm
    opacities = 1.0 / (1.0 + np.exp(-raw_o))
    fr = rasterize(means3D=np.asarray(means3D, dt), shs=np.asarray(features, dt), opacities=opacities, scales=scales,
                   rotations=rotations, dtype=dt, **settings)
    return OracleRawFrame(fr, scales, rotations, opacities, raw_q, norm)
