"""Drop-in for the reference's `diff_gaussian_rasterization` package, MI355X-native.

Public surface = what the reference imports and calls (gaussian_renderer/__init__.py:14,36-51,85-93):

    GaussianRasterizationSettings(image_height, image_width, tanfovx, tanfovy, bg, scale_modifier,
                                  viewmatrix, projmatrix, sh_degree, campos, prefiltered, debug)
    GaussianRasterizer(raster_settings).forward(means3D, means2D, opacities, shs=None, colors_precomp=None,
                                                scales=None, rotations=None, cov3D_precomp=None)
        -> (color[3,H,W] float32, radii[P] int32)
    GaussianRasterizer.markVisible(positions) -> bool[P]

Autograd contract (train.py:106, scene/gaussian_model.py:415-417): gradients for
(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3D_precomp, None) in that order;
`means2D.grad[:, :2]` is dL/d(NDC position) with the W/2, H/2 pixel scale folded in, `[:, 2] = 0`.

Everything below the autograd.Function is the C ABI of libgsrast.so (include/gsrast.h): hand-written
gfx950 HIP kernels.  There is no CPU path — missing library => RuntimeError at first use.
"""
from __future__ import annotations

from typing import NamedTuple, Optional

import torch
from torch import nn

from . import _native as N


class GaussianRasterizationSettings(NamedTuple):
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


def _f32c(t: Optional[torch.Tensor], device=None) -> Optional[torch.Tensor]:
    """Contiguous fp32 view/copy on `device`, or None for missing / empty tensors (the reference passes
    empty tensors for absent optionals; callers may hand in non-contiguous or detached views:
    scene/latent_gaussian_model.py:193-199, scene/gaussian_model.py:104-125)."""
    if t is None or t.numel() == 0:
        return None
    if t.dtype != torch.float32:
        t = t.float()
    if device is not None and t.device != device:
        raise ValueError(f"tensor on {t.device}, rasterizer inputs live on {device}")
    return t.contiguous()


def _workspace(nbytes: int, device) -> torch.Tensor:
    """Opaque byte workspace from torch's caching allocator.  Sizes are rounded up to 1/8-octave steps (<= 12.5 %
    slack) so that frames whose P or R differ a little — another camera, a densification step — hit the same cached
    block instead of a fresh hipMalloc (measured: 35 ms stalls after every densify without it)."""
    n = max(int(nbytes), 1)
    step = 1 << max(n.bit_length() - 4, 12)
    return torch.empty(-(-n // step) * step, dtype=torch.uint8, device=device)


def _binning_bytes(n: int, tiles: int) -> int:
    """gsr_binning_size(desc, n) without the call (csrc/gsr_binning.hip carve_binning; tests/test_abi.py holds the two together):
    six u32 arrays and one byte array of max(n, 1) entries, one 4 KB checkpoint per `seg` instances (+ 2) and the blend backward's
    unit lists (per shard n / seg + 1 full and 16 x 8 (tiles / 8 + 1) partial units of 8 bytes), each block padded to 256 bytes."""
    n, seg, up = max(int(n), 1), N.bwd_segment_entries(), lambda b: (b + 255) // 256 * 256
    return 6 * up(4 * n) + up(n) + up((n // seg + 2) * 4096) + up(8 * (n // seg + 1 + 16 * N.MAX_CHUNKS * (int(tiles) // 8 + 1)) * 8)


_binning_guess = {}        # (P, W, H, slab, device) -> instances the binning workspace of that frame shape last had to hold


class _Frame:
    """Native handles of one forward pass, kept alive by autograd's ctx for the backward."""
    __slots__ = ("desc", "cam", "keep", "plan", "geom_ws", "binning_ws", "image_ws", "radii", "color", "gauss", "M", "device", "raw", "pre")

    @property
    def R(self):
        """num_rendered of the reference (upper bound of the instances this frame binned)."""
        return int(self.plan.num_rendered)


def _camera(rs: GaussianRasterizationSettings, device):
    bg, vm, pm, cp = (_f32c(rs.bg, device), _f32c(rs.viewmatrix, device), _f32c(rs.projmatrix, device),
                      _f32c(rs.campos, device))
    if bg is None or vm is None or pm is None or cp is None:
        raise ValueError("bg, viewmatrix, projmatrix and campos must be non-empty device tensors")
    return N.Camera(N._ptr(bg), N._ptr(vm), N._ptr(pm), N._ptr(cp)), (bg, vm, pm, cp)


# torch.autograd.Function.forward runs with grad mode OFF, so "will a backward follow?" must be asked by the public entry
# points BEFORE they call .apply(): they leave the caller's grad mode here (thread-local: the library is re-entrant).
import threading as _threading

_tls = _threading.local()


def _apply(fn, *args):
    _tls.caller_grad = torch.is_grad_enabled()
    try:
        return fn.apply(*args)
    finally:
        _tls.caller_grad = None


def caller_grad_enabled() -> bool:
    g = getattr(_tls, "caller_grad", None)
    return torch.is_grad_enabled() if g is None else g


def rasterize_forward(means3D, sh, colors_precomp, opacities, scales, rotations, cov3D_precomp,
                      rs: GaussianRasterizationSettings, tile_rows=None, out_color=None, sh_rest=None, raw=False,
                      prepare_needs=None):
    """Stage 1 + stage 2 of the native forward.  Returns (color, radii, frame).
    raw=True: the tensors are the parameter store's RAW values (sh = _features_dc, sh_rest = _features_rest,
    opacities = logits, scales = log-scales, rotations = unnormalised) and the activations run in the kernels.
    prepare_needs: which gradients a backward will want (as rasterize_backward_geom's `needs`), when one will follow: the
    backward's output tensors are then allocated up front, while the host waits for the plan anyway."""
    device = means3D.device
    if device.type != "cuda":
        raise RuntimeError("diff_gaussian_rasterization (MI355X build) needs tensors on a HIP device; "
                           "there is no CPU path")
    P = int(means3D.shape[0])
    H, W = int(rs.image_height), int(rs.image_width)
    means3D = _f32c(means3D, device)
    sh, colors_precomp = _f32c(sh, device), _f32c(colors_precomp, device)
    opacities = _f32c(opacities, device)
    scales, rotations, cov3D_precomp = _f32c(scales, device), _f32c(rotations, device), _f32c(cov3D_precomp, device)
    sh_rest = _f32c(sh_rest, device)
    M = int(sh.shape[1]) if sh is not None else 0
    raw = int(raw)                   # 0: activated inputs; 1: raw, SH as features_dc + features_rest; 2: raw, SH as one [P,M,3] table
    if raw:
        if sh is None or scales is None or rotations is None or colors_precomp is not None or cov3D_precomp is not None:
            raise ValueError("raw mode takes SH coefficients, log-scales and raw rotations (no precomputed colours / covariances)")
        if raw == 1:
            if M != 1:
                raise ValueError("raw mode takes features_dc [P,1,3], features_rest [P,M-1,3], log-scales and raw rotations")
            M = 1 + (int(sh_rest.shape[1]) if sh_rest is not None else 0)
        elif raw != 2 or sh_rest is not None:
            raise ValueError("raw=2 takes the interleaved SH table [P,M,3] as `sh` and no sh_rest")
    fr = _Frame()
    fr.device, fr.M = device, M
    fr.desc = N.make_desc(P, int(rs.sh_degree), M, W, H, rs.tanfovx, rs.tanfovy, rs.scale_modifier, rs.prefiltered,
                          rs.debug, tile_rows)
    fr.cam, cam_keep = _camera(rs, device)
    fr.gauss = N.Gaussians(N._ptr(means3D), N._ptr(sh), N._ptr(colors_precomp), N._ptr(opacities), N._ptr(scales),
                           N._ptr(rotations), N._ptr(cov3D_precomp), N._ptr(sh_rest), raw)
    fr.keep = (cam_keep, means3D, sh, colors_precomp, opacities, scales, rotations, cov3D_precomp, sh_rest)
    fr.raw = raw
    fr.pre = None
    geom_bytes, image_bytes = N.workspace_sizes(fr.desc)
    fr.geom_ws = _workspace(geom_bytes, device)
    fr.image_ws = _workspace(image_bytes, device)
    fr.radii = torch.empty(P, dtype=torch.int32, device=device)            # every entry is written by the preprocess kernel
    with torch.cuda.device(device):
        # Everything that can be allocated before the plan readback is: the stream is idle while the host waits for R,
        # so the first kernels of the second stage should follow the readback as closely as possible.  The binning
        # workspace is guessed from the last frame of the same shape and re-allocated only when R outgrew it.
        if out_color is not None:
            color = out_color
        else:       # a full-frame render writes every pixel; a slab leaves the other rows untouched, so those start at 0
            color = (torch.empty if tile_rows is None else torch.zeros)(3, H, W, dtype=torch.float32, device=device)
        guess_key = (P, W, H, None if tile_rows is None else tuple(int(v) for v in tile_rows), device.index)
        guess = _binning_guess.get(guess_key, 0)
        tiles = ((W + 15) // 16) * ((H + 15) // 16)
        binning = _workspace(_binning_bytes(guess, tiles), device) if guess > 0 else None
        early = None
        if prepare_needs is not None and P > 0 and any(prepare_needs) and caller_grad_enabled():
            needs = tuple(bool(x) for x in prepare_needs)
            early = (needs,) + _alloc_grads(fr, needs, torch.empty) + (torch.empty(P, N.SCREEN_GRAD_STRIDE, dtype=torch.float32, device=device),)
        # With a workspace guessed from the last frame of this shape both stages run in ONE native call (gsr_forward): the
        # stream is idle from the plan readback until stage 2's first launch, and a trip through the interpreter in between
        # doubles that gap.  The call also enqueues the backward's zero fill itself, right after stage 2's last readback.
        done = False
        if binning is not None:
            plan, done = N.forward_both(fr.desc, fr.cam, fr.gauss, fr.geom_ws, fr.image_ws, fr.radii, binning, guess, color, device,
                                        early_fill=early[2] if early is not None else None)
        else:
            plan = N.forward_preprocess(fr.desc, fr.cam, fr.gauss, fr.geom_ws, fr.radii, device, image_ws=fr.image_ws)
        fr.plan = plan
        if early is not None:          # handed to the backward; prepare_backward() zero-fills them on sparse frames
            needs, tensors, grads, screen = early
            fr.pre = {"needs": needs, "tensors": tensors, "grads": grads, "screen": screen}
        # The binning workspace (24 B per instance) is sized for what the FIRST depth chunk can emit, not for the upper bound R
        # of every chunk (11.8 GB at 5e6 Gaussians / 4K): frames whose tiles saturate never get past that chunk.  A frame that
        # does need more stops with GSR_ERR_WORKSPACE before it writes anything it has no room for, and is re-run once with a
        # workspace for R; frames of that shape then start with R.  (Plain arithmetic here, no library calls: the stream is
        # idle between the plan readback and the first launch of stage 2.)
        R = int(plan.num_rendered)
        if done:
            capacity = int(plan.binning_capacity) if plan.binning_capacity > 0 else guess
            fr.binning_ws = binning
        else:
            first = int(plan.chunk_instances_max[0])
            capacity = min(first + first // 4 + (1 << 20), R) if plan.num_chunks > 1 else R        # = gsr_binning_first_chunk_capacity
            capacity = max(capacity, min(guess, R))
            if binning is not None and capacity <= guess:
                capacity = R                                   # the guessed workspace covered the first chunk, a later one did not fit
        while not done:
            need = _binning_bytes(capacity, tiles)
            if binning is None or binning.numel() < need:
                binning = _workspace(need, device)
            plan.binning_capacity = capacity
            fr.binning_ws = binning
            try:
                N.forward_render(fr.desc, fr.cam, fr.gauss, fr.geom_ws, binning, fr.image_ws, plan, color, device)
                break
            except N.GsrError as e:
                if e.status != N.ERR_WORKSPACE or capacity >= R:
                    raise
                capacity, binning = R, None
        _binning_guess[guess_key] = capacity
        if len(_binning_guess) > 64:
            _binning_guess.pop(next(iter(_binning_guess)))
    fr.color = color               # the blend backward walks the lists front to back and needs the final pixel
    return color, fr.radii, fr


def rasterize_backward_screen(fr: "_Frame", grad_color: torch.Tensor, only_for_own_geom: bool = False) -> torch.Tensor:
    """K7 + deterministic per-Gaussian reduction -> screen-space gradients [P, 12].
    only_for_own_geom: the tensor goes straight into this frame's geometry backward and nowhere else (the autograd path): rows it
    never reads may stay undefined.  The dense geometry backward reads the rows of VISIBLE Gaussians; when every planned chunk ran,
    all of them sit in the binned prefix, whose rows the reduction writes (zeros included): no 48 P-byte memset."""
    P = fr.desc.P
    grad_color = _f32c(grad_color, fr.device)
    screen = fr.pre.pop("screen", None) if fr.pre is not None else None        # allocated (and, on sparse frames, zero-filled) by the forward
    if screen is None:
        screen = torch.empty(max(P, 1), N.SCREEN_GRAD_STRIDE, dtype=torch.float32, device=fr.device)
        covered = fr.plan.num_rendered > 0 and fr.plan.chunks_run == fr.plan.num_chunks
        fr.plan.screen_prezeroed = 2 if (only_for_own_geom and covered) else 0            # (1: set only by prepare_backward(), for the very tensor it filled)
    if grad_color is None:
        return screen.zero_()[:P]
    with torch.cuda.device(fr.device):
        # one 48-B gradient row per EMITTED instance: backward-only scratch, returned to the allocator on exit
        rows = _workspace(N.backward_rows_size(fr.desc, fr.plan), fr.device)
        N.backward_render(fr.desc, fr.cam, fr.geom_ws, fr.binning_ws, fr.image_ws, rows, fr.plan, fr.color, grad_color, screen,
                          fr.device)
        fr.plan.screen_prezeroed = 0
    return screen[:P]


def _alloc_grads(fr: "_Frame", needs, alloc):
    """Gradient tensors of one frame (None where not wanted / not applicable) and their gsr_grads struct."""
    P, M, dev = fr.desc.P, fr.M, fr.device
    (_, means3D, sh, colors_precomp, opacities, scales, rotations, cov3D_precomp, sh_rest) = fr.keep

    def mk(flag, present, *shape):
        return alloc(*shape, dtype=torch.float32, device=dev) if (flag and present) else None
    t = (mk(needs[0], True, P, 3), mk(needs[1], True, P, 3), mk(needs[2], sh is not None, P, 1 if fr.raw == 1 else M, 3),
         mk(needs[3], colors_precomp is not None, P, 3), mk(needs[4], True, P, 1), mk(needs[5], scales is not None, P, 3),
         mk(needs[6], rotations is not None, P, 4), mk(needs[7], cov3D_precomp is not None, P, 6),
         mk(fr.raw == 1 and len(needs) > 8 and needs[8], sh_rest is not None, P, M - 1, 3))
    grads = N.Grads(N._ptr(t[0]), N._ptr(t[1]), N._ptr(t[2]), N._ptr(t[3]), N._ptr(t[4]), N._ptr(t[5]), N._ptr(t[6]),
                    N._ptr(t[7]), N._ptr(t[8]), 0)
    return t, grads


def prepare_backward(fr: "_Frame", needs, screen_prefix_only: bool = False) -> None:
    """Called right after the forward when a backward will follow: the stream is idle while the host walks back
    through the caller's code to the loss, so the backward's zero fills (screen-space gradients + the parameter
    gradients of the sparse geometry backward, ~280 MB at 1e6 Gaussians) are enqueued NOW, in one launch, into the
    tensors the forward allocated up front (rasterize_forward(prepare_needs=...)) or into fresh ones.
    (Measured and dropped: the same fill on a second stream beside the stage-2 kernels or beside k_render_bwd — the
    fill takes 65-86 us instead of 53 and the kernels it runs beside slow down by as much: no step time is gained.)"""
    P, plan = fr.desc.P, fr.plan
    # ctx.needs_input_grad stays True for leaf parameters under torch.no_grad() (eval / test-view renders): no backward can follow
    if P == 0 or plan.num_rendered <= 0 or plan.chunks_run <= 0 or not any(needs) or not caller_grad_enabled():
        return
    if N.effective_binned_ranks(plan) * 4 >= P:
        return                                  # the dense geometry backward writes every row itself
    needs = tuple(bool(x) for x in needs)
    if fr.pre is None or fr.pre["needs"] != needs:
        tensors, grads = _alloc_grads(fr, needs, torch.empty)
        fr.pre = {"needs": needs, "tensors": tensors, "grads": grads,
                  "screen": torch.empty(P, N.SCREEN_GRAD_STRIDE, dtype=torch.float32, device=fr.device)}
    if fr.pre["grads"].prezeroed:              # gsr_forward has already enqueued the fill (parameter gradients only)
        if screen_prefix_only:
            plan.screen_prezeroed = 2
        return
    with torch.cuda.device(fr.device):
        # screen_prefix_only: the screen-space gradients go straight into this frame's sparse geometry backward, which reads
        # the rows of the binned prefix only (the autograd path): that tensor needs no clearing at all
        N.backward_prepare(fr.desc, fr.gauss, plan, None if screen_prefix_only else fr.pre["screen"], fr.pre["grads"], fr.device)
        if screen_prefix_only:
            plan.screen_prezeroed = 2


def rasterize_backward_geom(fr: "_Frame", screen: torch.Tensor, needs, g0: int = 0, g1: Optional[int] = None,
                            binned_ranks: Optional[int] = None, rows: Optional[torch.Tensor] = None):
    """K8 + K9 on Gaussians [g0, g1).  `needs` = (means3D, means2D, sh, colors, opacities, scales, rotations,
    cov3D[, sh_rest]) booleans.  Returns the 8 gradient tensors (None where not needed / not applicable); for a
    raw-mode frame 9: sh is then d/d_features_dc [P,1,3] and the ninth d/d_features_rest [P,M-1,3].
    rows (int32 device tensor, whole range only): the Gaussians that can have a non-zero screen gradient, when that is not
    the frame's own binned prefix (a sharded render: the union over the ranks)."""
    P, dev = fr.desc.P, fr.device
    g1 = P if g1 is None else g1
    partial = (g0, g1) != (0, P)
    pre, fr.pre = fr.pre, None
    if pre is not None and not partial and tuple(bool(x) for x in needs) == pre["needs"]:
        tensors, grads = pre["tensors"], pre["grads"]                  # allocated and zero-filled right after the forward; used once
    else:
        tensors, grads = _alloc_grads(fr, needs, torch.zeros if partial else torch.empty)
    g_means3D, g_means2D, g_sh, g_col, g_op, g_sc, g_rot, g_cov, g_rest = tensors
    if P > 0 and g1 > g0:
        with torch.cuda.device(dev):
            if rows is not None and not partial:
                N.backward_geom_rows(fr.desc, fr.cam, fr.gauss, fr.radii, fr.geom_ws, screen, rows, grads, dev)
                binned_ranks = -2
            own = None
            if rows is None and binned_ranks is None:        # gradients of this very frame: its own binned depth prefix
                plan = own = fr.plan
                binned_ranks = int(plan.chunk_rank_begin[plan.chunks_run]) if plan.num_rendered > 0 and plan.chunks_run > 0 else 0
            if binned_ranks != -2:
                N.backward_geom(fr.desc, fr.cam, fr.gauss, fr.radii, fr.geom_ws, screen, g0, g1, grads, dev, binned_ranks,
                                own_plan=None if partial else own)
    if fr.raw:
        return g_means3D, g_means2D, g_sh, g_col, g_op, g_sc, g_rot, g_cov, g_rest
    return g_means3D, g_means2D, g_sh, g_col, g_op, g_sc, g_rot, g_cov


def _dump(path, payload):
    try:
        torch.save(payload, path)
    except Exception:      # the dump is a debugging aid; never mask the original error
        pass


_FRAME_TENSORS = ("geom_ws", "binning_ws", "image_ws", "radii", "color")       # color: the forward's OUTPUT, saved like the upstream
# extension saves its buffers — a caller that modifies the rendered image in place before backward() gets autograd's usual error


def _stash_frame(ctx, frame):
    """Hand the frame's device buffers (workspaces, radii, the contiguous input copies) to autograd's saved-tensor slots:
    like the upstream extension's saved buffers they are released right after backward() — or kept, and a second
    backward() allowed, under retain_graph=True; a second backward() without it raises autograd's usual error."""
    keep = [t for t in frame.keep[1:] if t is not None]
    ctx.save_for_backward(*(getattr(frame, n) for n in _FRAME_TENSORS), *frame.keep[0], *keep)
    ctx.frame_keep_mask = tuple(t is not None for t in frame.keep[1:])
    for n in _FRAME_TENSORS:
        setattr(frame, n, None)
    frame.keep = None
    ctx.frame = frame


def _unstash_frame(ctx):
    fr = ctx.frame
    saved = ctx.saved_tensors                      # raises if the graph's buffers were already freed
    for n, t in zip(_FRAME_TENSORS, saved[:5]):
        setattr(fr, n, t)
    it = iter(saved[9:])
    fr.keep = (tuple(saved[5:9]),) + tuple(next(it) if m else None for m in ctx.frame_keep_mask)
    return fr


def _restash_frame(fr):
    for n in _FRAME_TENSORS:
        setattr(fr, n, None)
    fr.keep = None
    fr.plan.screen_prezeroed = 0


class _RasterizeGaussians(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                raster_settings):
        rs = raster_settings
        args = (means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp)
        if rs.debug:
            cpu_args = tuple(a.detach().cpu().clone() for a in args)      # README.md:147-150 semantics
            try:
                color, radii, frame = rasterize_forward(*args, rs)
                torch.cuda.synchronize(means3D.device)
            except Exception:
                _dump("snapshot_fw.dump", (cpu_args, tuple(rs)))
                print("\nAn error occured in forward. Please forward snapshot_fw.dump for debugging.")
                raise
        else:
            color, radii, frame = rasterize_forward(*args, rs, prepare_needs=tuple(ctx.needs_input_grad[:8]))
            prepare_backward(frame, tuple(ctx.needs_input_grad[:8]), screen_prefix_only=True)
        _stash_frame(ctx, frame)
        ctx.raster_settings = rs
        ctx.shapes = (means2D.shape, opacities.shape)
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)         # radii's "gradient" would arrive as a zero-filled int32 [P] tensor: one launch per step
        return color, radii

    @staticmethod
    def backward(ctx, grad_out_color, _grad_radii):
        fr, rs = _unstash_frame(ctx), ctx.raster_settings
        needs = tuple(ctx.needs_input_grad[:8])
        # input order: means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp
        order = (needs[0], needs[1], needs[2], needs[3], needs[4], needs[5], needs[6], needs[7])

        def run():
            screen = rasterize_backward_screen(fr, grad_out_color, only_for_own_geom=True)
            return rasterize_backward_geom(fr, screen, order)
        if rs.debug:
            try:
                out = run()
                torch.cuda.synchronize(fr.device)
            except Exception:
                _dump("snapshot_bw.dump", (tuple(None if k is None else k.detach().cpu() for k in fr.keep[1:]),
                                           grad_out_color.detach().cpu(), tuple(rs)))
                print("\nAn error occured in backward. Writing snapshot_bw.dump for debugging.\n")
                raise
        else:
            out = run()
        g_means3D, g_means2D, g_sh, g_col, g_op, g_sc, g_rot, g_cov = out
        if g_op is not None:
            g_op = g_op.view(ctx.shapes[1])
        if g_means2D is not None:
            g_means2D = g_means2D.view(ctx.shapes[0])
        _restash_frame(fr)
        return g_means3D, g_means2D, g_sh, g_col, g_op, g_sc, g_rot, g_cov, None


class _RasterizeGaussiansRaw(torch.autograd.Function):
    """SURVEY 8a row a14 fused: same rasterizer, fed with the parameter store's RAW tensors.  The activations of
    scene/gaussian_model.py:47-60 (exp, sigmoid, normalize) and the torch.cat of get_features (:120-123) run inside
    the preprocess kernel, their chain rule inside the geometry backward: gradients arrive on the raw parameters."""

    @staticmethod
    def forward(ctx, xyz, means2D, features_dc, features_rest, opacity_logits, log_scales, raw_rotations, raster_settings):
        rs = raster_settings
        n = ctx.needs_input_grad       # xyz, means2D, f_dc, f_rest, opacity, scales, rotations
        needs = (n[0], n[1], n[2], False, n[4], n[5], n[6], False, n[3])
        # features_rest None: features_dc is the whole interleaved table [P,M,3] (scene.GaussianModel's packed leaf)
        color, radii, frame = rasterize_forward(xyz, features_dc, None, opacity_logits, log_scales, raw_rotations, None, rs,
                                                sh_rest=features_rest, raw=2 if features_rest is None else 1,
                                                prepare_needs=None if rs.debug else needs)
        if rs.debug:
            torch.cuda.synchronize(xyz.device)
        prepare_backward(frame, needs, screen_prefix_only=True)
        _stash_frame(ctx, frame)
        ctx.shapes = (means2D.shape, opacity_logits.shape)
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)         # radii's "gradient" would arrive as a zero-filled int32 [P] tensor: one launch per step
        return color, radii

    @staticmethod
    def backward(ctx, grad_out_color, _grad_radii):
        fr = _unstash_frame(ctx)
        n = ctx.needs_input_grad       # xyz, means2D, f_dc, f_rest, opacity, scales, rotations
        needs = (n[0], n[1], n[2], False, n[4], n[5], n[6], False, n[3])
        screen = rasterize_backward_screen(fr, grad_out_color, only_for_own_geom=True)
        g_xyz, g_means2D, g_dc, _, g_op, g_sc, g_rot, _, g_rest = rasterize_backward_geom(fr, screen, needs)
        if g_op is not None:
            g_op = g_op.view(ctx.shapes[1])
        if g_means2D is not None:
            g_means2D = g_means2D.view(ctx.shapes[0])
        _restash_frame(fr)
        return g_xyz, g_means2D, g_dc, g_rest, g_op, g_sc, g_rot, None


def rasterize_gaussians(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                        raster_settings):
    return _apply(_RasterizeGaussians, means3D, means2D, sh, colors_precomp, opacities, scales, rotations,
                  cov3Ds_precomp, raster_settings)


# ---- opt-in: fuse the reference's getters without touching the caller -------------------------------------------
# With FUSE_GETTERS on (module attribute, or GSR_FUSE_GETTERS=1 in the environment) GaussianRasterizer.forward looks at
# the autograd history of its arguments.  If they are exactly what the reference's GaussianModel getters produce from
# leaf parameters — opacities = sigmoid(leaf), scales = exp(leaf), rotations = F.normalize(leaf),
# shs = cat((leaf_dc, leaf_rest), 1), means3D a leaf (scene/gaussian_model.py:101-125) — the frame is rendered
# from those leaves with the activations inside the kernels (forward_raw).  Pixels are the same; the gradients reach
# the same leaves with the same values to fp32 rounding, but the ~15 torch kernels of the getters' backward (and the
# 2 x 192 MB copies of cat's) never run.  Anything that does not match falls back to the plain path.  Not default:
# hooks / retain_grad() on the intermediate activated tensors would not fire.
import os as _os

FUSE_GETTERS = _os.environ.get("GSR_FUSE_GETTERS", "0") not in ("", "0")


def _leaf_of(fn, index=0):
    """The leaf tensor behind input `index` of autograd node `fn`, or None."""
    nxt = fn.next_functions
    if index >= len(nxt) or nxt[index][0] is None or type(nxt[index][0]).__name__ != "AccumulateGrad":
        return None
    return nxt[index][0].variable


def _unary_getter(t, node_name):
    fn = t.grad_fn
    if fn is None or type(fn).__name__ != node_name or len(fn.next_functions) != 1:
        return None
    leaf = _leaf_of(fn)
    return leaf if leaf is not None and leaf.shape == t.shape and leaf.dtype == torch.float32 else None


def _normalize_getter(t):
    """leaf such that t = torch.nn.functional.normalize(leaf) (p = 2, dim = 1, eps = 1e-12), or None."""
    fn = t.grad_fn
    if fn is None or type(fn).__name__ != "DivBackward0" or len(fn.next_functions) != 2:
        return None
    leaf = _leaf_of(fn, 0)
    node = fn.next_functions[1][0]
    for name in ("ExpandBackward0", "ClampMinBackward0", "LinalgVectorNormBackward0"):
        if node is None or type(node).__name__ != name:
            return None
        if name == "ClampMinBackward0" and abs(float(node._saved_min) - 1e-12) > 1e-18:
            return None
        if name == "LinalgVectorNormBackward0":
            if float(node._saved_ord) != 2.0 or tuple(node._saved_dim) != (1,) or not node._saved_keepdim:
                return None
            if _leaf_of(node) is not leaf:
                return None
            break
        node = node.next_functions[0][0] if len(node.next_functions) == 1 else None
    return leaf if leaf is not None and leaf.shape == t.shape and leaf.dtype == torch.float32 else None


def _match_getters(means3D, opacities, shs, scales, rotations):
    """(xyz, f_dc, f_rest, opacity, scaling, rotation) leaves when the arguments are the reference getters' outputs."""
    try:
        if not (means3D.is_leaf and means3D.dtype == torch.float32 and means3D.is_cuda):
            return None
        op, sc, rot = _unary_getter(opacities, "SigmoidBackward0"), _unary_getter(scales, "ExpBackward0"), _normalize_getter(rotations)
        fn = shs.grad_fn
        if op is None or sc is None or rot is None or fn is None or type(fn).__name__ != "CatBackward0":
            return None
        if int(fn._saved_dim) != 1 or len(fn.next_functions) != 2:
            return None
        dc, rest = _leaf_of(fn, 0), _leaf_of(fn, 1)
        P = means3D.shape[0]
        if dc is None or rest is None or tuple(dc.shape) != (P, 1, 3) or rest.dim() != 3 or rest.shape[0] != P or rest.shape[2] != 3:
            return None
        if dc.dtype != torch.float32 or rest.dtype != torch.float32 or rest.shape[1] + 1 != shs.shape[1] or rest.shape[1] == 0:
            return None
        return means3D, dc, rest, op, sc, rot
    except (AttributeError, RuntimeError, TypeError):
        return None


class GaussianRasterizer(nn.Module):
    def __init__(self, raster_settings: GaussianRasterizationSettings):
        super().__init__()
        self.raster_settings = raster_settings

    def markVisible(self, positions: torch.Tensor) -> torch.Tensor:
        """Frustum test per point (the reference's `_C.mark_visible`; unused in-tree, kept for API parity)."""
        rs = self.raster_settings
        with torch.no_grad():
            pos = _f32c(positions, positions.device)
            present = torch.zeros(positions.shape[0], dtype=torch.uint8, device=positions.device)
            if pos is not None:
                vm, pm = _f32c(rs.viewmatrix, positions.device), _f32c(rs.projmatrix, positions.device)
                with torch.cuda.device(positions.device):
                    N.mark_visible(pos, vm, pm, present)
        return present.bool()

    def forward(self, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                cov3D_precomp=None):
        rs = self.raster_settings
        if (shs is None and colors_precomp is None) or (shs is not None and colors_precomp is not None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        empty = torch.empty(0, dtype=torch.float32, device=means3D.device)
        shs = empty if shs is None else shs
        colors_precomp = empty if colors_precomp is None else colors_precomp
        scales = empty if scales is None else scales
        rotations = empty if rotations is None else rotations
        cov3D_precomp = empty if cov3D_precomp is None else cov3D_precomp
        rs = rs._replace(sh_degree=int(rs.sh_degree), image_height=int(rs.image_height),
                         image_width=int(rs.image_width))
        if FUSE_GETTERS and torch.is_grad_enabled() and not rs.debug and shs.numel() and scales.numel() and rotations.numel():
            leaves = _match_getters(means3D, opacities, shs, scales, rotations)
            if leaves is not None:
                xyz, dc, rest, op, sc, rot = leaves
                return _apply(_RasterizeGaussiansRaw, xyz, means2D, dc, rest, op, sc, rot, rs)
        return rasterize_gaussians(means3D, means2D, shs, colors_precomp, opacities, scales, rotations,
                                   cov3D_precomp, rs)

    def forward_raw(self, xyz, means2D, features_dc, features_rest, opacity_logits, log_scales, raw_rotations):
        """Extension (not in the reference API): render straight from the reference GaussianModel's raw parameters
        (`_xyz, _features_dc, _features_rest, _opacity, _scaling, _rotation`), activations fused into the kernels.
        Same pixels and radii as forward(get_xyz, ..., get_opacity, get_features, get_scaling, get_rotation) and the
        same gradients on the raw parameters as autograd through those getters, to fp32 rounding."""
        rs = self.raster_settings
        rs = rs._replace(sh_degree=int(rs.sh_degree), image_height=int(rs.image_height), image_width=int(rs.image_width))
        packed = features_rest is None and features_dc.dim() == 3 and features_dc.shape[1] > 1      # one [P,M,3] table
        if not packed and (features_rest is None or features_rest.numel() == 0):
            features_rest = torch.empty(0, dtype=torch.float32, device=xyz.device)
        return _apply(_RasterizeGaussiansRaw, xyz, means2D, features_dc, features_rest, opacity_logits, log_scales, raw_rotations, rs)
