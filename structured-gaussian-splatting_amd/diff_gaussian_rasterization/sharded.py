"""Multi-GPU rasterization by tile-row slabs — one process per GPU, torch.distributed
over RCCL/xGMI ("nccl" backend on ROCm).

The reference has no distributed code; this is net-new.  Partitioning:

  * image: contiguous slabs of 16-px tile rows, one per rank; Gaussian parameters replicated.
    Every rank runs the per-Gaussian preprocess for all P (cheap, no exchange), bins and blends only the
    tiles of its slab (gsr_frame_desc.tile_row_begin/end).
  * forward exchange: ALL-GATHER of the rendered slabs -> full image on every rank (the loss needs an
    11x11 SSIM window across slab borders).
  * backward exchange, default ("allreduce_screen"): ALL-REDUCE (sum) of the per-Gaussian SCREEN-SPACE
    gradients (12 floats / Gaussian, 9 used), after which EVERY rank runs the whole per-Gaussian geometry
    backward and so holds the full parameter gradients — no parameter-gradient collective at all.  The
    geometry backward costs ~0.1 ms for 1e6 Gaussians on an MI355X, an all-gather of the 236 MB of parameter
    gradients over xGMI costs milliseconds.  Only Gaussians that some rank actually binned can have a
    non-zero screen gradient, and the progressive pipeline bins a PREFIX of the (rank-independent) depth
    order: the exchanged block is screen[depth_order[:max over ranks of the prefix length]] — ~1 MB instead
    of 48 MB on the depth-complex benchmark scene.
  * backward exchange, alternative ("reduce_scatter", the plan of SURVEY 8e (ii)): REDUCE-SCATTER of the
    screen-space gradients over Gaussian shards, geometry backward on each rank's shard, ALL-GATHER of the
    parameter gradients.  Right when the optimizer is sharded by Gaussian and wants sharded gradients.

xGMI is point-to-point: per-GPU messages are kept large and few (one collective per direction);
sizes at P = 1e6: slabs 24.9 MB total, screen grads <= 48 MB, parameter grads 236 MB (alternative only).

`backend` is the compute provider: the native HIP library by default (fails loudly if missing).  Tests
inject a CPU provider to exercise this file's partitioning/collective logic under gloo.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch

SCREEN_STRIDE = 12


def slab_bounds(tile_rows: int, world: int, weights: Optional[List[float]] = None) -> List[Tuple[int, int]]:
    """Split tile rows [0, tile_rows) into `world` contiguous slabs.  Without weights: as even as possible.
    With per-tile-row weights (e.g. splat instances per row): boundaries at equal cumulative weight
    (SURVEY 7 "slab load balance").  Slabs may be empty when world > tile_rows."""
    if weights is None:
        base, rem = divmod(tile_rows, world)
        out, y = [], 0
        for r in range(world):
            n = base + (1 if r < rem else 0)
            out.append((y, y + n))
            y += n
        return out
    assert len(weights) == tile_rows
    cum, acc = [], 0.0
    for w in weights:
        acc += float(w)
        cum.append(acc)
    total = acc or 1.0
    bounds = [0]
    for r in range(1, world):
        target = total * r / world
        y = bounds[-1]
        while y < tile_rows and cum[y] <= target:
            y += 1
        bounds.append(min(max(y, bounds[-1]), tile_rows))
    bounds.append(tile_rows)
    return [(bounds[r], bounds[r + 1]) for r in range(world)]


def gaussian_shard(P: int, world: int, rank: int) -> Tuple[int, int, int]:
    """Rank's Gaussian range [g0, g1) and the padded shard length (equal on all ranks)."""
    shard = (P + world - 1) // world
    g0 = min(rank * shard, P)
    return g0, min(g0 + shard, P), shard


class NativeBackend:
    """Compute provider = libgsrast.so through the drop-in package (the product path)."""

    def forward(self, means3D, sh, colors_precomp, opacities, scales, rotations, cov3D_precomp, rs, tile_rows, out_color,
                sh_rest=None, raw=False):
        from . import rasterize_forward
        return rasterize_forward(means3D, sh, colors_precomp, opacities, scales, rotations, cov3D_precomp, rs,
                                 tile_rows=tile_rows, out_color=out_color, sh_rest=sh_rest, raw=raw)

    def backward_screen(self, frame, grad_color):
        from . import rasterize_backward_screen
        return rasterize_backward_screen(frame, grad_color)

    def backward_geom(self, frame, screen, needs, g0, g1, rows=None):
        """rows: Gaussian indices (device tensor) that may carry a screen gradient, None = any (dense)."""
        from . import rasterize_backward_geom
        if rows is None:
            return rasterize_backward_geom(frame, screen, needs, g0, g1, binned_ranks=-1)
        return rasterize_backward_geom(frame, screen, needs, g0, g1, rows=rows.to(torch.int32))

    def gather_rows(self, frame, k_max, n_max, partial):
        """(idx int32 [n_max], packed [n_max, 12]): the Gaussians with 0 <= key <= k_max in index order and their rows of
        `partial`, in two native launches (gsr_exchange_rows_gather)."""
        from . import _native as N
        with torch.cuda.device(partial.device):
            return N.exchange_rows_gather(frame.desc, frame.geom_ws, k_max, partial, n_max)

    def scatter_rows(self, frame, idx, packed, screen):
        from . import _native as N
        with torch.cuda.device(screen.device):
            N.exchange_rows_scatter(frame.desc, idx, packed, screen)

    def prepare_backward(self, frame, needs):
        """Early zero fill of the backward's outputs (gsr_backward_prepare) while the stream waits for the all-gather."""
        from . import prepare_backward
        prepare_backward(frame, needs)

    def row_work(self, frame, Gx, Gy):
        """[Gy] float tensor: instances this frame binned per 16-px tile row (zero outside its slab) — the slab
        balancer's measure of work."""
        from . import _native as N
        v = N.debug_views(frame.desc, frame.geom_ws, frame.binning_ws, frame.image_ws, frame.plan)
        rng = v["ranges"][:max(int(frame.plan.chunks_run), 1)].long()             # [chunks, Tn, 2]
        return (rng[..., 1] - rng[..., 0]).sum(0).view(Gy, Gx).sum(1).to(torch.float32)

    def binned_prefix(self, frame):
        """(keys, key_end, n): keys [P] = a sort key per Gaussian that is the SAME on every rank (bits of the view depth;
        negative = not visible); this rank binned exactly the n Gaussians with 0 <= key <= key_end, so only they can own
        gradient rows here.  Chunks are selected by depth key, so 'everything up to the largest key_end of any rank' is
        the same set on every rank."""
        from . import _native as N
        plan = frame.plan
        run = int(plan.chunks_run) if plan.num_rendered > 0 else 0
        keys, _ = N.frame_arrays(frame.desc, frame.geom_ws)
        if run <= 0:
            return keys, -1, 0
        # (the last chunk's end is 'everything': clamp to the largest key an int32 view can hold)
        return keys, min(int(plan.chunk_key_end[run - 1]), 0x7FFFFFFF), int(plan.chunk_rank_begin[run])


class _Comm:
    """The three collectives of the path, with a gloo-compatible form for the CPU tests."""

    def __init__(self, dist, world, rank, group=None, async_gather=True):
        self.dist, self.world, self.rank, self.group = dist, world, rank, group
        self.async_gather = async_gather                   # False: the plain blocking all_gather (A/B switch, see ShardedRenderer)
        self.gloo = dist.get_backend(group) == "gloo"      # tests only: device tensors are staged through the host
        self.native_rs = not self.gloo
        self._pinned, self._pinned_turn = {}, 0

    def all_gather(self, shard: torch.Tensor) -> torch.Tensor:
        """[n, ...] per rank -> [world * n, ...]."""
        dev = shard.device
        if self.gloo and dev.type != "cpu":
            return self.all_gather(shard.cpu()).to(dev)
        out = torch.empty((self.world * shard.shape[0],) + tuple(shard.shape[1:]), dtype=shard.dtype, device=dev)
        self.dist.all_gather_into_tensor(out, shard.contiguous(), group=self.group)
        return out

    def all_gather_begin(self, shard: torch.Tensor):
        """all_gather in two halves: the collective is started here (on RCCL's own stream), the returned callable makes the
        current stream wait for it and hands out the result — what the caller enqueues in between runs beside the transfer."""
        if self.gloo or shard.device.type == "cpu" or not self.async_gather:
            out = self.all_gather(shard)
            return lambda: out
        out = torch.empty((self.world * shard.shape[0],) + tuple(shard.shape[1:]), dtype=shard.dtype, device=shard.device)
        work = self.dist.all_gather_into_tensor(out, shard.contiguous(), group=self.group, async_op=True)

        def finish():
            work.wait()                    # stream-level wait (no host block)
            return out
        return finish

    def all_reduce_sum(self, t: torch.Tensor) -> torch.Tensor:
        if self.gloo and t.device.type != "cpu":
            return self.all_reduce_sum(t.cpu()).to(t.device)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group)
        return t

    def widest_prefix(self, key_end: int, n: int, device):
        """(max over ranks of key_end, the n of a rank that holds it): one small all-gather (the path without a forward
        piggyback)."""
        t = torch.tensor([[int(key_end), int(n)]], dtype=torch.int64, device="cpu" if self.gloo else device)
        allr = self.all_gather(t)
        return max((int(k), int(m)) for k, m in allr.tolist())

    def pinned_scratch(self, n: int, dtype) -> torch.Tensor:
        """A pinned host vector of n elements from a ring of eight (the caller fills it and enqueues ONE non-blocking copy to
        the device; eight frames later, long after that copy has run, the buffer comes round again)."""
        ring = self._pinned.setdefault(("scratch", n, dtype), [])
        if len(ring) < 8:
            ring.append(torch.empty(n, dtype=dtype).pin_memory())
            return ring[-1]
        self._scratch_turn = getattr(self, "_scratch_turn", 0) + 1
        return ring[self._scratch_turn % 8]

    def read_halves_later(self, halves: torch.Tensor):
        """halves [world, 4] = (key_end + 1 >> 16, key_end + 1 & 0xFFFF, n >> 16, n & 0xFFFF) per rank as floats: one copy to
        pinned memory + an event now; the returned callable waits for that event only and returns (largest key_end over
        the ranks, the n of a rank that holds it).  (A blocking .item() in the backward would drain the whole forward +
        loss from the stream.)"""
        pick = lambda rows: max((int(a) * 65536 + int(b) - 1, int(c) * 65536 + int(d)) for a, b, c, d in rows)
        if halves.device.type != "cuda":
            v = pick(halves.tolist())
            return lambda: v
        # (pinned buffers are recycled: pin_memory() is a host allocation of tens of microseconds; two in rotation, a frame's
        # backward reads its buffer before the next-but-one forward overwrites it)
        key = (tuple(halves.shape), halves.dtype)
        ring = self._pinned.setdefault(key, [])
        if len(ring) < 2:
            ring.append(torch.empty(halves.shape, dtype=halves.dtype).pin_memory())
        host = ring[self._pinned_turn % len(ring)] if len(ring) == 2 else ring[-1]
        self._pinned_turn += 1
        host.copy_(halves, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(halves.device))

        def read():
            ev.synchronize()
            return pick(host.tolist())
        return read

    def reduce_scatter_sum(self, full: torch.Tensor) -> torch.Tensor:
        """[world * n, ...] per rank -> [n, ...] = sum over ranks of this rank's block."""
        n = full.shape[0] // self.world
        if self.native_rs:
            out = torch.empty((n,) + tuple(full.shape[1:]), dtype=full.dtype, device=full.device)
            self.dist.reduce_scatter_tensor(out, full.contiguous(), op=self.dist.ReduceOp.SUM, group=self.group)
            return out
        dev = full.device                           # gloo has no reduce_scatter: all_reduce + slice (tests only)
        full = full.contiguous().cpu().clone() if dev.type != "cpu" else full.contiguous().clone()
        self.dist.all_reduce(full, op=self.dist.ReduceOp.SUM, group=self.group)
        return full[self.rank * n:(self.rank + 1) * n].clone().to(dev)


class _ShardedRasterize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, rs, shard, sh_rest=None,
                packed_raw=False):
        # sh_rest given = raw-parameter mode (SURVEY a14 fused): sh is _features_dc, sh_rest _features_rest, opacities /
        # scales / rotations the raw parameters; the activations run inside the kernels on every rank.  packed_raw: the same with
        # the SH coefficients as ONE interleaved table in `sh` (scene.GaussianModel's leaf; raw mode 2 of the ABI)
        comm, backend = shard.comm, shard.backend
        raw = 2 if packed_raw else (1 if sh_rest is not None else 0)
        ctx.raw = raw
        H, W = int(rs.image_height), int(rs.image_width)
        Gy = (H + 15) // 16
        # slab balancer: adopts weights measured on an earlier frame BEFORE this frame's slabs are cut (so that
        # pixel_rows() / training_loss() after this render see the same slabs), and says whether this frame measures
        balance = shard.balance_now(Gy) and hasattr(backend, "row_work")
        slabs = shard.slabs(Gy)
        ty0, ty1 = slabs[comm.rank]
        rows_max = max(min(b * 16, H) - min(a * 16, H) for a, b in slabs)
        # render own slab straight into the full-size frame (rows of the other slabs arrive with the all-gather)
        full = torch.empty(3, H, W, dtype=means3D.dtype, device=means3D.device)
        slab = (ty0, ty1) if ty1 > ty0 else (Gy, Gy)
        if raw:
            color, radii, frame = backend.forward(means3D, sh, None, opacities, scales, rotations, None, rs, slab, full,
                                                  sh_rest=sh_rest, raw=raw)
        else:
            color, radii, frame = backend.forward(means3D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp,
                                                  rs, slab, full)
        y0, y1 = min(ty0 * 16, H), min(ty1 * 16, H)
        # all-gather payload per rank: its slab padded to rows_max rows, plus four trailing words that carry the depth key up to
        # which it binned and how many Gaussians that is (as 16-bit halves, exact in any float type) so that the widest
        # prefix over the ranks needs no collective of its own
        n_words = 3 * rows_max * W
        mine = torch.empty(1, n_words + 4 + (Gy if balance else 0), dtype=full.dtype, device=full.device)
        if y1 > y0:
            mine[0, :n_words].view(3, rows_max, W)[:, :y1 - y0] = full[:, y0:y1]
        needs = tuple(ctx.needs_input_grad[:8]) + ((bool(ctx.needs_input_grad[10]) and raw == 1,) if raw else ())
        ctx.needs = needs
        from . import caller_grad_enabled          # Function.forward itself runs with grad mode off
        want_prefix = shard.backward_mode == "allreduce_screen" and any(needs) and caller_grad_enabled()
        keys, k_mine, n_mine = backend.binned_prefix(frame) if want_prefix else (None, -1, 0)
        halves = ((int(k_mine) + 1) >> 16, (int(k_mine) + 1) & 0xFFFF, int(n_mine) >> 16, int(n_mine) & 0xFFFF)
        if mine.device.type == "cuda":
            # one asynchronous copy from a recycled PINNED buffer (a copy from pageable memory waits for the stream; four scalar
            # fills were four launches in the stretch where the host is what the stream waits for)
            host4 = comm.pinned_scratch(4, mine.dtype)
            for j, v in enumerate(halves):
                host4[j] = float(v)
            mine[0, n_words:n_words + 4].copy_(host4, non_blocking=True)
        else:
            for j, v in enumerate(halves):
                mine[0, n_words + j].fill_(float(v))
        if balance:          # this rank's per-tile-row work rides along; the sum over ranks is next frames' slab weights
            mine[0, n_words + 4:] = backend.row_work(frame, (W + 15) // 16, Gy).to(mine.dtype)
        gather_done = comm.all_gather_begin(mine)               # [world, 3 * rows_max * W + 4 (+ Gy)]
        hook = getattr(backend, "prepare_backward", None)
        if hook is not None and want_prefix:
            hook(frame, needs)             # the backward's zero fill (50 us of stores) runs beside the slabs' transfer
        gathered = gather_done()
        if balance:
            shard.set_row_work(gathered[:, n_words + 4:].sum(0))
        for r, (a, b) in enumerate(slabs):
            a_px, b_px = min(a * 16, H), min(b * 16, H)
            if b_px > a_px and r != comm.rank:
                full[:, a_px:b_px] = gathered[r, :n_words].view(3, rows_max, W)[:, :b_px - a_px]
        ctx.frame, ctx.shard, ctx.rs = frame, shard, rs
        ctx.n_max = None
        if want_prefix:
            # widest binned depth prefix over the ranks: selects and sizes the backward's gradient exchange; on its way to
            # pinned host memory now, so that the backward finds it without a stream drain
            ctx.n_max = (keys, comm.read_halves_later(gathered[:, n_words:n_words + 4]) if keys is not None else None)
            # ... and the host picks the pair up HERE, behind the copies and the fill it has just enqueued (the stream is busy for
            # another ~50 us): waiting for it in the backward stalled the host until the whole forward + loss had drained, and every
            # launch of the backward then arrived late (0.39 ms of idle stream per step at world size 1).
            if ctx.n_max[1] is not None:
                pair = ctx.n_max[1]()
                ctx.n_max = (keys, lambda: pair)
        ctx.shapes = (means2D.shape, opacities.shape)
        ctx.mark_non_differentiable(radii)
        return full, radii

    @staticmethod
    def backward(ctx, grad_color, _):
        frame, shard = ctx.frame, ctx.shard
        comm, backend = shard.comm, shard.backend
        P = frame.desc.P if hasattr(frame, "desc") else frame.P
        needs = ctx.needs
        # (1) my slab's contribution to every Gaussian's screen-space gradient
        partial = backend.backward_screen(frame, grad_color)                       # [P, 12]
        if shard.backward_mode == "allreduce_screen":
            # (2) sum over slabs; only Gaussians some rank binned can be non-zero: those whose key is <= the largest chunk end
            # any rank reached (the same set, in index order, on every rank: keys do not depend on the slab)
            if ctx.n_max is not None:
                keys, read = ctx.n_max
                k_max, n_max = read() if read is not None else (-1, P)
            else:
                keys, k_mine, n_mine = backend.binned_prefix(frame)
                k_max, n_max = comm.widest_prefix(k_mine, n_mine, partial.device) if keys is not None else (-1, P)
            idx = None
            if keys is None or n_max >= P:
                screen = comm.all_reduce_sum(partial.contiguous())
            else:
                screen = partial                                    # rows outside the set are zero on every rank
                if n_max > 0 and k_max >= 0 and hasattr(backend, "gather_rows") and partial.is_cuda and partial.is_contiguous():
                    # native: the list and the packed rows in two launches, the summed rows written back in one
                    idx, packed = backend.gather_rows(frame, k_max, n_max, partial)
                    backend.scatter_rows(frame, idx, comm.all_reduce_sum(packed), screen)
                else:
                    idx = torch.nonzero_static((keys >= 0) & (keys <= k_max), size=n_max).view(-1)  # n_max is exact: no host sync
                    if n_max > 0:
                        screen[idx] = comm.all_reduce_sum(partial[idx].contiguous())   # in place: `partial` is not used again
            # (3) every rank runs the whole geometry backward: full parameter gradients, no further collective
            out = list(backend.backward_geom(frame, screen, needs, 0, P, idx))
        else:
            g0, g1, slen = gaussian_shard(P, comm.world, comm.rank)
            padded = torch.zeros(comm.world * slen, SCREEN_STRIDE, dtype=partial.dtype, device=partial.device)
            padded[:P] = partial
            # (2) sum over slabs, scattered by Gaussian shard
            mine = comm.reduce_scatter_sum(padded)                                 # [slen, 12]
            screen = torch.zeros(max(P, 1), SCREEN_STRIDE, dtype=partial.dtype, device=partial.device)
            if g1 > g0:
                screen[g0:g1] = mine[:g1 - g0]
            # (3) geometry backward on my shard, (4) all-gather of the parameter gradients
            grads = backend.backward_geom(frame, screen[:P], needs, g0, g1)
            out = []
            for g in grads:
                if g is None:
                    out.append(None)
                    continue
                blk = torch.zeros((slen,) + tuple(g.shape[1:]), dtype=g.dtype, device=g.device)
                if g1 > g0:
                    blk[:g1 - g0] = g[g0:g1]
                out.append(comm.all_gather(blk)[:P])
        g_rest = out[8] if ctx.raw else None
        g_means3D, g_means2D, g_sh, g_col, g_op, g_sc, g_rot, g_cov = out[:8]
        if g_op is not None:
            g_op = g_op.reshape(ctx.shapes[1])
        if g_means2D is not None:
            g_means2D = g_means2D.reshape(ctx.shapes[0])
        ctx.frame = None
        return g_means3D, g_means2D, g_sh, g_col, g_op, g_sc, g_rot, g_cov, None, None, g_rest, None


class ShardedRenderer:
    """render()-shaped front end for world_size > 1 (same arguments and returned dict as
    gaussian_renderer.render, reference gaussian_renderer/__init__.py:18-100)."""

    def __init__(self, dist, world: int, rank: int, backend=None, group=None, row_weights=None,
                 backward_mode: str = "allreduce_screen", balance_every: int = 0, async_gather: bool = True):
        """row_weights: fixed per-tile-row weights for the slab boundaries (None = equal rows).  balance_every = k > 0:
        SURVEY 7 "slab load balance" — every k-th frame the ranks' measured per-tile-row work (instances binned) rides
        on the all-gather and the next frames' slabs are cut at equal cumulative work.  async_gather=False: the slabs'
        all-gather as one blocking collective instead of async_op + a stream-level wait with the backward's zero fill enqueued in
        between — the asynchronous form has only ever run at world size 1 on real RCCL (no multi-GPU lease so far): the switch
        lets a 2-GPU run A/B the two (tests/test_gpu_sharded.py::test_two_rank_rccl_equals_single_render does)."""
        assert backward_mode in ("allreduce_screen", "reduce_scatter")
        self.balance_every, self._frames, self._pending = int(balance_every), 0, None
        self.comm = _Comm(dist, world, rank, group, async_gather=async_gather)
        self.backend = NativeBackend() if backend is None else backend
        self.row_weights = row_weights
        self.backward_mode = backward_mode

    def pixel_rows(self, H: int):
        """This rank's image rows [y0, y1) for an H-row frame (its tile-row slab in pixels)."""
        a, b = self.slabs((H + 15) // 16)[self.comm.rank]
        return min(a * 16, H), min(b * 16, H)

    def training_loss(self, image, gt, lambda_dssim: float = 0.2):
        """train.py:104-105 for a frame rendered by this renderer: every rank evaluates the L1 / D-SSIM terms of its
        own rows (fused kernels) and the ranks' partial sums are added; the gradient comes back for the rank's rows only.
        Same value as loss_utils.training_loss on the full frame (summation order aside)."""
        import loss_utils
        if not image.is_cuda:
            return loss_utils.training_loss(image, gt, lambda_dssim)
        return loss_utils.training_loss_rows(image, gt, lambda_dssim, self.pixel_rows(int(image.shape[1])),
                                             lambda t: self.comm.all_reduce_sum(t))

    def balance_now(self, Gy: int) -> bool:
        """Called once per frame by the forward: adopt weights measured earlier, say whether this frame measures."""
        if self.balance_every <= 0:
            return False
        if self._pending is not None:                       # measured on an earlier frame; host copy has long arrived
            w = self._pending()
            self._pending = None
            if len(w) == Gy and sum(w) > 0:
                floor = 0.01 * sum(w) / Gy                  # empty rows still cost a little: keep every weight positive
                self.row_weights = [max(float(x), floor) for x in w]
        self._frames += 1
        return self._frames % self.balance_every == 1 or self.balance_every == 1

    def set_row_work(self, work: torch.Tensor) -> None:
        """work [Gy] (device or host): summed over ranks; reaches the host through pinned memory, read at a later frame."""
        if work.device.type != "cuda":
            w = [float(x) for x in work.tolist()]
            self._pending = lambda: w
            return
        host = torch.empty(work.shape, dtype=torch.float32).pin_memory()
        host.copy_(work.to(torch.float32), non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream(work.device))

        def read():
            ev.synchronize()
            return host.tolist()
        self._pending = read

    def slabs(self, Gy: int):
        return slab_bounds(Gy, self.comm.world, self.row_weights if self.row_weights and len(self.row_weights) == Gy else None)

    def rasterize(self, rs, means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
                  cov3D_precomp=None):
        if (shs is None) == (colors_precomp is None):
            raise Exception("Please provide excatly one of either SHs or precomputed colors!")
        if ((scales is None or rotations is None) and cov3D_precomp is None) or \
                ((scales is not None or rotations is not None) and cov3D_precomp is not None):
            raise Exception("Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!")
        e = torch.empty(0, dtype=torch.float32, device=means3D.device)
        rs = rs._replace(sh_degree=int(rs.sh_degree))
        from . import _apply
        return _apply(_ShardedRasterize, means3D, means2D, e if shs is None else shs,
                      e if colors_precomp is None else colors_precomp, opacities,
                      e if scales is None else scales, e if rotations is None else rotations,
                      e if cov3D_precomp is None else cov3D_precomp, rs, self)

    def render(self, viewpoint_camera, pc, pipe, bg_color, scaling_modifier=1.0, override_color=None):
        import math

        from . import GaussianRasterizationSettings
        xyz = pc.get_xyz
        screenspace_points = torch.zeros_like(xyz, requires_grad=True)     # a leaf keeps its .grad (gaussian_renderer.render)
        rs = GaussianRasterizationSettings(
            image_height=int(viewpoint_camera.image_height), image_width=int(viewpoint_camera.image_width),
            tanfovx=math.tan(viewpoint_camera.FoVx * 0.5), tanfovy=math.tan(viewpoint_camera.FoVy * 0.5), bg=bg_color,
            scale_modifier=scaling_modifier, viewmatrix=viewpoint_camera.world_view_transform,
            projmatrix=viewpoint_camera.full_proj_transform, sh_degree=pc.active_sh_degree,
            campos=viewpoint_camera.camera_center, prefiltered=False, debug=pipe.debug)
        frozen = any(getattr(pc, f, False) for f in ("freeze_means", "freeze_scales", "freeze_rotations", "freeze_opacities"))
        packed = bool(getattr(pc, "packed_features", False))
        fused = getattr(pipe, "fused_activations", None)
        if fused is None:                    # as gaussian_renderer.render: this package's own model renders from its raw leaves
            fused = packed
        if (fused and packed and override_color is None and not pipe.compute_cov3D_python
                and not getattr(pipe, "convert_SHs_python", False) and isinstance(self.backend, NativeBackend) and not frozen):
            rs = rs._replace(sh_degree=int(rs.sh_degree))
            e = torch.empty(0, dtype=torch.float32, device=xyz.device)
            from . import _apply
            image, radii = _apply(_ShardedRasterize, pc._xyz, screenspace_points, pc._features, e, pc._opacity, pc._scaling,
                                  pc._rotation, e, rs, self, None, True)
            return {"render": image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0, "radii": radii}
        if (fused and not packed and override_color is None and not pipe.compute_cov3D_python
                and not getattr(pipe, "convert_SHs_python", False) and hasattr(pc, "_features_rest") and pc._features_rest.numel()
                and isinstance(self.backend, NativeBackend) and not frozen):      # frozen parameters: the getters' detach() must run
            rs = rs._replace(sh_degree=int(rs.sh_degree))
            e = torch.empty(0, dtype=torch.float32, device=xyz.device)
            from . import _apply
            image, radii = _apply(_ShardedRasterize, pc._xyz, screenspace_points, pc._features_dc, e, pc._opacity, pc._scaling,
                                  pc._rotation, e, rs, self, pc._features_rest)
            return {"render": image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0, "radii": radii}
        scales = rotations = cov = None
        if pipe.compute_cov3D_python:
            cov = pc.get_covariance(scaling_modifier)
        else:
            scales, rotations = pc.get_scaling, pc.get_rotation
        shs = colors = None
        if override_color is None:
            shs = pc.get_features
        else:
            colors = override_color
        image, radii = self.rasterize(rs, xyz, screenspace_points, pc.get_opacity, shs=shs, colors_precomp=colors,
                                      scales=scales, rotations=rotations, cov3D_precomp=cov)
        return {"render": image, "viewspace_points": screenspace_points, "visibility_filter": radii > 0, "radii": radii}
