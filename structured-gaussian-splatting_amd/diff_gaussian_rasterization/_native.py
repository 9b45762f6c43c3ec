"""ctypes binding of libgsrast.so (C ABI: include/gsrast.h).

There is NO fallback: if the HIP library is missing or a call fails, this module raises.  The product
path never imports the CPU oracle (oracle/ is test infrastructure).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import torch

_PKG_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB_PATH = os.environ.get("GSR_LIB_PATH") or os.path.join(_PKG_ROOT, "lib", "libgsrast.so")      # override: kernel-variant experiments
_CSRC = os.path.join(_PKG_ROOT, "csrc")
SCREEN_GRAD_STRIDE = 12

_f32p = C.c_void_p


class FrameDesc(C.Structure):
    _fields_ = [("P", C.c_int32), ("sh_degree", C.c_int32), ("sh_coeffs", C.c_int32), ("width", C.c_int32),
                ("height", C.c_int32), ("tanfovx", C.c_float), ("tanfovy", C.c_float), ("scale_modifier", C.c_float),
                ("prefiltered", C.c_int32), ("debug", C.c_int32), ("tile_row_begin", C.c_int32),
                ("tile_row_end", C.c_int32)]


class Camera(C.Structure):
    _fields_ = [("bg", _f32p), ("viewmatrix", _f32p), ("projmatrix", _f32p), ("campos", _f32p)]


class Gaussians(C.Structure):
    _fields_ = [("means3D", _f32p), ("shs", _f32p), ("colors_precomp", _f32p), ("opacities", _f32p),
                ("scales", _f32p), ("rotations", _f32p), ("cov3D_precomp", _f32p),
                ("shs_rest", _f32p), ("raw", C.c_int32)]      # raw-parameter mode (a14): see include/gsrast.h


class Grads(C.Structure):
    _fields_ = [("means3D", _f32p), ("means2D", _f32p), ("shs", _f32p), ("colors_precomp", _f32p),
                ("opacities", _f32p), ("scales", _f32p), ("rotations", _f32p), ("cov3D_precomp", _f32p),
                ("shs_rest", _f32p), ("prezeroed", C.c_int32)]


MAX_CHUNKS = 8
LAST_SHIFT = 26


class FramePlan(C.Structure):
    _fields_ = [("num_rendered", C.c_int64), ("num_visible", C.c_int32), ("num_chunks", C.c_int32),
                ("chunk_rank_begin", C.c_int32 * (MAX_CHUNKS + 1)), ("chunk_instances_max", C.c_int64 * MAX_CHUNKS),
                ("chunks_run", C.c_int32), ("sort_result", C.c_int32), ("instances_emitted", C.c_int64),
                ("binning_initialised", C.c_int32), ("screen_prezeroed", C.c_int32), ("binning_capacity", C.c_int64),
                ("chunk_key_end", C.c_uint32 * MAX_CHUNKS), ("chunks_sorted", C.c_int32), ("chunks_filtered", C.c_int32),
                ("tile_order_ready", C.c_int32), ("key_max", C.c_uint32)]


class DebugViews(C.Structure):
    _fields_ = [("splat_records", C.c_void_p), ("tiles_touched", C.c_void_p), ("depth_order", C.c_void_p),
                ("point_offsets", C.c_void_p), ("clamped", C.c_void_p), ("sorted_gaussian", C.c_void_p),
                ("ranges", C.c_void_p), ("final_T", C.c_void_p), ("n_contrib", C.c_void_p), ("tile_walk", C.c_void_p),
                ("bwd_units", C.c_void_p), ("bwd_unit_count", C.c_void_p), ("bwd_unit_cap_full", C.c_uint32), ("bwd_unit_cap_part", C.c_uint32)]


EXPORTS = ("gsr_version", "gsr_last_error", "gsr_workspace_sizes", "gsr_binning_size", "gsr_binning_first_chunk_capacity", "gsr_forward_preprocess", "gsr_forward",
           "gsr_forward_render", "gsr_bwd_segment_entries", "gsr_backward_rows_size", "gsr_backward_prepare", "gsr_backward_render", "gsr_backward_geom", "gsr_backward_geom_rows", "gsr_frame_arrays", "gsr_exchange_rows_gather", "gsr_exchange_rows_scatter", "gsr_mark_visible", "gsr_debug_get_views", "gsr_profile_enable",
           "gsr_profile_read", "gsr_loss_workspace_size", "gsr_loss_l1_ssim_forward", "gsr_loss_l1_ssim_backward", "gsr_loss_l1_ssim_forward_rows", "gsr_loss_l1_ssim_backward_rows", "gsr_loss_l1_backward",
           "gsr_debug_sort_temp_bytes", "gsr_debug_sort_pairs", "gsr_dist2_workspace_size", "gsr_dist2_knn3", "gsr_adam_step", "gsr_adam_step_split", "gsr_adam_step_multi", "gsr_densify_stats",
           "gsr_activations_forward", "gsr_activations_backward")

_lib = None


def lib_path() -> str:
    return _LIB_PATH


def build(force: bool = False) -> str:
    """Compile libgsrast.so for gfx950 with hipcc (csrc/Makefile).  Cross-compiles without a GPU."""
    if force:
        subprocess.check_call(["make", "-C", _CSRC, "clean"], stdout=subprocess.DEVNULL)
    subprocess.check_call(["make", "-C", _CSRC, "-j4"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            raise RuntimeError(
                f"libgsrast.so not found at {_LIB_PATH}: the HIP extension is required (no CPU fallback exists). "
                f"Build it with `make -C {_CSRC}` or `python -c 'import __graft_entry__ as g; g.build()'`.")
        lib = C.CDLL(_LIB_PATH)
        lib.gsr_last_error.restype = C.c_char_p
        for name in EXPORTS:
            if not hasattr(lib, name):
                raise RuntimeError(f"libgsrast.so does not export {name}")
        _lib = lib
    return _lib


class GsrError(RuntimeError):
    status = 0


ERR_WORKSPACE = -4


def _check(rc: int, what: str):
    if rc != 0:
        msg = load().gsr_last_error().decode("utf-8", "replace")
        e = GsrError(f"{what} failed (status {rc}): {msg}")
        e.status = rc
        raise e


def _ptr(t: Optional[torch.Tensor]):
    if t is None or t.numel() == 0:
        return None
    return C.c_void_p(t.data_ptr())


def _stream(device) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def make_desc(P, sh_degree, sh_coeffs, width, height, tanfovx, tanfovy, scale_modifier, prefiltered, debug,
              tile_rows=None) -> FrameDesc:
    ty0, ty1 = (0, 0) if tile_rows is None else (int(tile_rows[0]), int(tile_rows[1]))
    return FrameDesc(int(P), int(sh_degree), int(sh_coeffs), int(width), int(height), float(tanfovx), float(tanfovy),
                     float(scale_modifier), int(bool(prefiltered)), int(bool(debug)), ty0, ty1)


def workspace_sizes(desc: FrameDesc):
    g, i = C.c_size_t(0), C.c_size_t(0)
    _check(load().gsr_workspace_sizes(C.byref(desc), C.byref(g), C.byref(i)), "gsr_workspace_sizes")
    return g.value, i.value


def binning_size(desc: FrameDesc, R: int) -> int:
    b = C.c_size_t(0)
    _check(load().gsr_binning_size(C.byref(desc), C.c_int64(R), C.byref(b)), "gsr_binning_size")
    return b.value


def binning_first_chunk_capacity(plan: FramePlan) -> int:
    n = C.c_int64(0)
    _check(load().gsr_binning_first_chunk_capacity(C.byref(plan), C.byref(n)), "gsr_binning_first_chunk_capacity")
    return n.value


def forward_preprocess(desc, cam: Camera, g: Gaussians, geom_ws, radii, device, image_ws=None) -> FramePlan:
    plan = FramePlan()
    _check(load().gsr_forward_preprocess(C.byref(desc), C.byref(cam), C.byref(g), _ptr(geom_ws), _ptr(image_ws), _ptr(radii),
                                         C.byref(plan), _stream(device)), "gsr_forward_preprocess")
    return plan


def forward_both(desc, cam: Camera, g: Gaussians, geom_ws, image_ws, radii, binning_ws, binning_capacity, out_color, device,
                 early_fill: Optional[Grads] = None):
    """gsr_forward: both stages in one call.  Returns (plan, done): done = False when the binning workspace was too small for the
    first chunk (nothing of stage 2 ran: allocate and call forward_render)."""
    plan = FramePlan()
    rc = load().gsr_forward(C.byref(desc), C.byref(cam), C.byref(g), _ptr(geom_ws), _ptr(image_ws), _ptr(radii), C.byref(plan),
                            _ptr(binning_ws), C.c_int64(int(binning_capacity)), _ptr(out_color),
                            None if early_fill is None else C.byref(early_fill), _stream(device))
    if rc == ERR_WORKSPACE:           # the guessed workspace did not do (for the first chunk, or for a later one)
        return plan, False
    _check(rc, "gsr_forward")
    return plan, True


def forward_render(desc, cam: Camera, g: Gaussians, geom_ws, binning_ws, image_ws, plan: FramePlan, out_color, device):
    _check(load().gsr_forward_render(C.byref(desc), C.byref(cam), C.byref(g), _ptr(geom_ws), _ptr(binning_ws), _ptr(image_ws),
                                     C.byref(plan), _ptr(out_color), _stream(device)), "gsr_forward_render")


def backward_rows_size(desc: FrameDesc, plan: FramePlan) -> int:
    b = C.c_size_t(0)
    _check(load().gsr_backward_rows_size(C.byref(desc), C.byref(plan), C.byref(b)), "gsr_backward_rows_size")
    return b.value


def backward_prepare(desc, g: Gaussians, plan: FramePlan, screen_grads, grads: Grads, device):
    _check(load().gsr_backward_prepare(C.byref(desc), C.byref(g), C.byref(plan), _ptr(screen_grads), C.byref(grads),
                                       _stream(device)), "gsr_backward_prepare")


def backward_render(desc, cam: Camera, geom_ws, binning_ws, image_ws, rows_ws, plan: FramePlan, out_color, dL_dcolor, screen_grads,
                    device):
    _check(load().gsr_backward_render(C.byref(desc), C.byref(cam), _ptr(geom_ws), _ptr(binning_ws), _ptr(image_ws),
                                      _ptr(rows_ws), C.byref(plan), _ptr(out_color), _ptr(dL_dcolor), _ptr(screen_grads), _stream(device)),
           "gsr_backward_render")


_SEG = [0]


def bwd_segment_entries() -> int:
    """List entries per work unit of the blend backward (a build constant of the library)."""
    if not _SEG[0]:
        _SEG[0] = int(load().gsr_bwd_segment_entries())
    return _SEG[0]


def backward_geom(desc, cam: Camera, g: Gaussians, radii, geom_ws, screen_grads, g0, g1, grads: Grads, device,
                  binned_ranks: int = -1, own_plan: Optional[FramePlan] = None):
    """own_plan: the frame's plan when screen_grads are this very frame's (gsr_backward_render of the same plan), else None."""
    _check(load().gsr_backward_geom(C.byref(desc), C.byref(cam), C.byref(g), _ptr(radii), _ptr(geom_ws),
                                    _ptr(screen_grads), C.c_int32(g0), C.c_int32(g1), C.c_int32(binned_ranks),
                                    None if own_plan is None else C.byref(own_plan), C.byref(grads), _stream(device)), "gsr_backward_geom")


def effective_binned_ranks(plan: FramePlan) -> int:
    """csrc/gsr_internal.h effective_binned_ranks: the ranks of the chunks that ran, a live-filtered chunk counted as nothing."""
    if plan.num_rendered <= 0 or plan.chunks_run <= 0:
        return 0
    return sum(int(plan.chunk_rank_begin[c + 1]) - int(plan.chunk_rank_begin[c]) for c in range(int(plan.chunks_run))
               if not (int(plan.chunks_filtered) >> c) & 1)


def backward_geom_rows(desc, cam: Camera, g: Gaussians, radii, geom_ws, screen_grads, rows, grads: Grads, device):
    """Sparse geometry backward over the Gaussians listed in `rows` (int32 device tensor)."""
    _check(load().gsr_backward_geom_rows(C.byref(desc), C.byref(cam), C.byref(g), _ptr(radii), _ptr(geom_ws), _ptr(screen_grads),
                                         _ptr(rows), C.c_int32(rows.numel()), C.byref(grads), _stream(device)), "gsr_backward_geom_rows")


_frame_offsets = {}


def frame_arrays(desc, geom_ws):
    """(depth_keys, depth_order): int32 views [P] into the geometry workspace (keys: bits of the view depth, -1 = invisible)."""
    P = desc.P
    offs = _frame_offsets.get(P)
    if offs is None:
        k, o = C.c_void_p(), C.c_void_p()
        _check(load().gsr_frame_arrays(C.byref(desc), _ptr(geom_ws), C.byref(k), C.byref(o)), "gsr_frame_arrays")
        base = geom_ws.data_ptr()
        offs = _frame_offsets[P] = (int(k.value) - base, int(o.value) - base)
    return tuple(geom_ws[o:o + 4 * P].view(torch.int32) for o in offs)


def exchange_rows_gather(desc, geom_ws, key_max: int, screen, n_rows: int):
    """(rows int32 [n_rows], packed float32 [n_rows, 12]): the Gaussians with depth key <= key_max in index order and their
    screen-gradient rows, packed for one all-reduce."""
    # n_rows comes from ANOTHER rank's count: entries this rank's keys do not fill must be inert (-1 is skipped by the scatter and by
    # the sparse geometry backward; their packed rows are zero), not whatever the allocator left there
    rows = torch.full((n_rows,), -1, dtype=torch.int32, device=screen.device)
    packed = torch.zeros(n_rows, SCREEN_GRAD_STRIDE, dtype=torch.float32, device=screen.device)
    _check(load().gsr_exchange_rows_gather(C.byref(desc), _ptr(geom_ws), C.c_uint32(int(key_max) & 0xFFFFFFFF), _ptr(screen), C.c_int32(n_rows),
                                           _ptr(rows), _ptr(packed), _stream(screen.device)), "gsr_exchange_rows_gather")
    return rows, packed


def exchange_rows_scatter(desc, rows, packed, screen):
    _check(load().gsr_exchange_rows_scatter(C.byref(desc), C.c_int32(rows.numel()), _ptr(rows), _ptr(packed), _ptr(screen),
                                            _stream(screen.device)), "gsr_exchange_rows_scatter")


def mark_visible(means3D, viewmatrix, projmatrix, present):
    _check(load().gsr_mark_visible(C.c_int32(means3D.shape[0]), _ptr(means3D), _ptr(viewmatrix), _ptr(projmatrix),
                                   _ptr(present), _stream(means3D.device)), "gsr_mark_visible")


def debug_views(desc, geom_ws, binning_ws, image_ws, plan: FramePlan) -> dict:
    """Intermediate arrays as torch views INTO the workspaces (tests / profiling)."""
    v = DebugViews()
    _check(load().gsr_debug_get_views(C.byref(desc), _ptr(geom_ws), _ptr(binning_ws), _ptr(image_ws), C.byref(plan),
                                      C.byref(v)), "gsr_debug_get_views")
    P, N, R = desc.P, desc.width * desc.height, int(plan.binning_capacity or plan.num_rendered)
    Tn = ((desc.width + 15) // 16) * ((desc.height + 15) // 16)

    def view(ws, addr, nbytes, dtype, shape):
        if not addr or ws is None or nbytes == 0:
            return None
        off = addr - ws.data_ptr()
        return ws[off:off + nbytes].view(dtype).view(shape)
    return dict(
        splat_records=view(geom_ws, v.splat_records, P * 48, torch.float32, (P, 12)),
        tiles_touched=(lambda t: None if t is None else t[:, 0])(view(geom_ws, v.tiles_touched, P * 8, torch.int32, (P, 2))),
        optical_mass=(lambda t: None if t is None else t[:, 1])(view(geom_ws, v.tiles_touched, P * 8, torch.int32, (P, 2))),
        depth_order=view(geom_ws, v.depth_order, P * 4, torch.int32, (P,)),
        point_offsets=view(geom_ws, v.point_offsets, P * 4, torch.int32, (P,)),
        clamped=view(geom_ws, v.clamped, P, torch.uint8, (P,)),
        # bits 0..27 Gaussian index, bits 28..31 quadrant mask (copies: the workspace word holds both)
        sorted_gaussian=(lambda t: None if t is None else t & 0x0FFFFFFF)(view(binning_ws, v.sorted_gaussian, R * 4, torch.int32, (R,))),
        sorted_quadrants=(lambda t: None if t is None else (t >> 28) & 0xF)(view(binning_ws, v.sorted_gaussian, R * 4, torch.int32, (R,))),
        ranges=view(image_ws, v.ranges, MAX_CHUNKS * Tn * 8, torch.int32, (MAX_CHUNKS, Tn, 2)),
        final_T=view(image_ws, v.final_T, N * 4, torch.float32, (desc.height, desc.width)),
        n_contrib=view(image_ws, v.n_contrib, N * 4, torch.int32, (desc.height, desc.width)),
        tile_walk=view(image_ws, v.tile_walk, MAX_CHUNKS * Tn * 4, torch.int32, (MAX_CHUNKS, Tn)),
        # (tile | chunk << 24, segment) per work unit of the blend backward: per shard 17 lists [cap_full + 16 cap_part], their lengths
        bwd_units=view(binning_ws, v.bwd_units, 8 * (int(v.bwd_unit_cap_full) + 16 * int(v.bwd_unit_cap_part)) * 8, torch.int32,
                       (8, int(v.bwd_unit_cap_full) + 16 * int(v.bwd_unit_cap_part), 2)),
        bwd_unit_caps=(int(v.bwd_unit_cap_full), int(v.bwd_unit_cap_part)),
        bwd_unit_count=view(image_ws, v.bwd_unit_count, 8 * 17 * 4, torch.int32, (8, 17)))


def profile_enable(on: bool):
    """Per-kernel hipEvent timing inside the library (thread-local; resets the accumulators)."""
    _check(load().gsr_profile_enable(C.c_int(int(on))), "gsr_profile_enable")


def profile_read() -> dict:
    """{kernel name: (total_ms, launches)} since profile_enable(True); synchronises the recorded events."""
    n_max = 32
    names = ((C.c_char * 32) * n_max)()
    ms = (C.c_float * n_max)()
    cnt = (C.c_int32 * n_max)()
    n = load().gsr_profile_read(n_max, names, ms, cnt)
    return {names[i].value.decode(): (float(ms[i]), int(cnt[i])) for i in range(n)}


def loss_workspace_size(C_, H, W) -> int:
    b = C.c_size_t(0)
    _check(load().gsr_loss_workspace_size(C.c_int32(C_), C.c_int32(H), C.c_int32(W), C.byref(b)), "gsr_loss_workspace_size")
    return b.value


def loss_forward(image, target, lam, workspace, out3):
    Cn, H, W = image.shape
    _check(load().gsr_loss_l1_ssim_forward(C.c_int32(Cn), C.c_int32(H), C.c_int32(W), C.c_float(lam), _ptr(image), _ptr(target),
                                           _ptr(workspace), _ptr(out3), _stream(image.device)), "gsr_loss_l1_ssim_forward")


def loss_backward(image, target, lam, upstream, workspace, grad_image):
    Cn, H, W = image.shape
    _check(load().gsr_loss_l1_ssim_backward(C.c_int32(Cn), C.c_int32(H), C.c_int32(W), C.c_float(lam), _ptr(upstream),
                                            _ptr(image), _ptr(target), _ptr(workspace), _ptr(grad_image),
                                            _stream(image.device)), "gsr_loss_l1_ssim_backward")


def loss_l1_backward(image, target, upstream, grad_image):
    _check(load().gsr_loss_l1_backward(C.c_int64(image.numel()), _ptr(upstream), _ptr(image), _ptr(target), _ptr(grad_image),
                                       _stream(image.device)), "gsr_loss_l1_backward")


def loss_forward_rows(image, target, workspace, out2, row_begin, row_end):
    Cn, H, W = image.shape
    _check(load().gsr_loss_l1_ssim_forward_rows(C.c_int32(Cn), C.c_int32(H), C.c_int32(W), _ptr(image), _ptr(target),
                                                _ptr(workspace), _ptr(out2), C.c_int32(row_begin), C.c_int32(row_end),
                                                _stream(image.device)), "gsr_loss_l1_ssim_forward_rows")


def loss_backward_rows(image, target, lam, upstream, workspace, grad_image, row_begin, row_end):
    Cn, H, W = image.shape
    _check(load().gsr_loss_l1_ssim_backward_rows(C.c_int32(Cn), C.c_int32(H), C.c_int32(W), C.c_float(lam), _ptr(upstream),
                                                 _ptr(image), _ptr(target), _ptr(workspace), _ptr(grad_image),
                                                 C.c_int32(row_begin), C.c_int32(row_end), _stream(image.device)),
           "gsr_loss_l1_ssim_backward_rows")


def debug_sort_pairs(keys: torch.Tensor, vals: torch.Tensor, end_bit: int, count_on_device: bool = False):
    """Stable radix sort of (int32-viewed-as-u32 key, value) pairs through the library's own sort (test hook)."""
    n = keys.numel()
    k = [keys.contiguous().clone(), torch.empty_like(keys)]
    v = [vals.contiguous().clone(), torch.empty_like(vals)]
    b = C.c_size_t(0)
    _check(load().gsr_debug_sort_temp_bytes(C.byref(b)), "gsr_debug_sort_temp_bytes")
    temp = torch.empty(b.value, dtype=torch.uint8, device=keys.device)
    res = C.c_int32(0)
    with torch.cuda.device(keys.device):
        _check(load().gsr_debug_sort_pairs(_ptr(k[0]) or C.c_void_p(temp.data_ptr()), _ptr(k[1]) or C.c_void_p(temp.data_ptr()),
                                           _ptr(v[0]) or C.c_void_p(temp.data_ptr()), _ptr(v[1]) or C.c_void_p(temp.data_ptr()),
                                           C.c_int64(n), C.c_int32(end_bit), C.c_int32(int(count_on_device)), _ptr(temp),
                                           C.byref(res), _stream(keys.device)), "gsr_debug_sort_pairs")
    return k[res.value], v[res.value]


def dist2_knn3(xyz: torch.Tensor) -> torch.Tensor:
    """Mean squared distance to the 3 nearest other points (gsr_dist2_knn3)."""
    P = int(xyz.shape[0])
    pts = xyz.detach().float().contiguous()
    out = torch.zeros(P, dtype=torch.float32, device=pts.device)
    if P == 0:
        return out
    b = C.c_size_t(0)
    _check(load().gsr_dist2_workspace_size(C.c_int32(P), C.byref(b)), "gsr_dist2_workspace_size")
    ws = torch.empty(b.value, dtype=torch.uint8, device=pts.device)
    with torch.cuda.device(pts.device):
        _check(load().gsr_dist2_knn3(C.c_int32(P), _ptr(pts), _ptr(out), _ptr(ws), _stream(pts.device)), "gsr_dist2_knn3")
    return out


def adam_step(param, grad, exp_avg, exp_avg_sq, lr, beta1, beta2, eps, step):
    _check(load().gsr_adam_step(C.c_int64(param.numel()), _ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), C.c_float(lr),
                                C.c_float(beta1), C.c_float(beta2), C.c_float(eps), C.c_int64(int(step)),
                                _stream(param.device)), "gsr_adam_step")


def adam_step_split(param, grad, exp_avg, exp_avg_sq, split, lr_head, lr_tail, beta1, beta2, eps, step):
    """param [rows, ...] contiguous; the first `split` elements of every row step with lr_head, the rest with lr_tail."""
    rows = param.shape[0]
    row_len = param.numel() // max(rows, 1)
    _check(load().gsr_adam_step_split(C.c_int64(rows), C.c_int32(row_len), C.c_int32(int(split)), _ptr(param), _ptr(grad), _ptr(exp_avg),
                                      _ptr(exp_avg_sq), C.c_float(lr_head), C.c_float(lr_tail), C.c_float(beta1), C.c_float(beta2),
                                      C.c_float(eps), C.c_int64(int(step)), _stream(param.device)), "gsr_adam_step_split")


class AdamTensor(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("exp_avg", C.c_void_p), ("exp_avg_sq", C.c_void_p), ("n", C.c_int64),
                ("lr", C.c_float), ("lr_tail", C.c_float), ("step", C.c_int64), ("row_len", C.c_int32), ("split", C.c_int32)]


ADAM_MAX_TENSORS = 8


def adam_step_multi(items, beta1, beta2, eps):
    """items: up to ADAM_MAX_TENSORS tuples (param, grad, exp_avg, exp_avg_sq, lr, lr_tail, step, row_len, split) — the whole
    optimizer step in one launch (row_len = 0: one learning rate for the tensor)."""
    arr = (AdamTensor * len(items))()
    for a, (p, g, m, v, lr, lr_tail, step, row_len, split) in zip(arr, items):
        a.param, a.grad, a.exp_avg, a.exp_avg_sq, a.n = _ptr(p), _ptr(g), _ptr(m), _ptr(v), p.numel()
        a.lr, a.lr_tail, a.step, a.row_len, a.split = float(lr), float(lr_tail), int(step), int(row_len), int(split)
    _check(load().gsr_adam_step_multi(C.c_int32(len(items)), arr, C.c_float(beta1), C.c_float(beta2), C.c_float(eps),
                                      _stream(items[0][0].device)), "gsr_adam_step_multi")


def activations_forward(scaling_raw, rotation_raw, opacity_raw):
    """(exp, normalize, sigmoid) of the three raw tensors in one launch; any may be None."""
    some = next(t for t in (scaling_raw, rotation_raw, opacity_raw) if t is not None)
    P = some.shape[0]
    outs = [None if t is None else torch.empty_like(t) for t in (scaling_raw, rotation_raw, opacity_raw)]
    with torch.cuda.device(some.device):
        _check(load().gsr_activations_forward(C.c_int64(P), _ptr(scaling_raw), _ptr(rotation_raw), _ptr(opacity_raw), _ptr(outs[0]),
                                              _ptr(outs[1]), _ptr(outs[2]), _stream(some.device)), "gsr_activations_forward")
    return tuple(outs)


def activations_backward(scales, rotation_raw, opacities, g_scales, g_rotations, g_opacities):
    """Gradients w.r.t. the raw tensors; a None incoming gradient gives a None result."""
    some = next(t for t in (scales, rotation_raw, opacities) if t is not None)
    P = some.shape[0]
    outs = [None if g is None else torch.empty_like(g) for g in (g_scales, g_rotations, g_opacities)]
    with torch.cuda.device(some.device):
        _check(load().gsr_activations_backward(C.c_int64(P), _ptr(scales), _ptr(rotation_raw), _ptr(opacities), _ptr(g_scales),
                                               _ptr(g_rotations), _ptr(g_opacities), _ptr(outs[0]), _ptr(outs[1]), _ptr(outs[2]),
                                               _stream(some.device)), "gsr_activations_backward")
    return tuple(outs)


def densify_stats(radii, viewspace_grad, max_radii2D, xyz_gradient_accum, denom):
    _check(load().gsr_densify_stats(C.c_int32(radii.shape[0]), _ptr(radii), _ptr(viewspace_grad), _ptr(max_radii2D),
                                    _ptr(xyz_gradient_accum), _ptr(denom), _stream(radii.device)), "gsr_densify_stats")
