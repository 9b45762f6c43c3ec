#!/usr/bin/env python3
"""bench.py — train-step images/s (fwd+bwd) of the Gaussian-rasterizer hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg3] [--no-cpu-baseline] [--no-secondary]

A "step" is the reference's timed training window (train.py:79-108): render() through the drop-in
rasterizer + L1/D-SSIM loss + backward into the leaf parameters, on BASELINE.json configs[2]'s scene
(1e6 Gaussians, 1920x1080, SH degree 3; SURVEY Appendix B, seed 3).  The parameter store is this package's
scene.GaussianModel (the reference class's surface, getters as native ops: SURVEY 8a row a14).  Inputs are resident in HBM before
the timed region.  For N > 1 the image is split into tile-row slabs: the scene is fixed and each
rank renders its slab of the SAME image, so scaling is "strong".  `python bench.py --gpus N` without a launcher
starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` itself (as a child process).

Prints ONE JSON line on rank 0.  Besides the contract's keys:
  roofline      dominant kernel of the step, timed live with hipEvents inside libgsrast.so; `valu` prices the blend
                kernel's measured instruction mix (profiles/r03_valu_mix.json) with the issue rates measured by
                profiles/valu_microbench on this chip
  cpu_baseline  the CPU oracle (a port, test infrastructure) on rank 0 at N = 1 only; .cfg1 = BASELINE configs[0]
  secondary     the same step on a NON-saturating scene (scene_synth CONFIGS["cfg3n"]: R/P = 3.6, 98 % of the visible
                Gaussians receive a gradient) with its own per-kernel table and roofline, and cfg3 with another seed
  torch_getters the same step over a store whose getters are the reference's torch ops (gaussian_params.GaussianParams):
                what a caller who keeps the reference's own GaussianModel class gets from the drop-in rasterizer alone
  reference_loss_composition   the step with the reference's own two-call loss (l1_loss + ssim: utils/loss_utils.py)
"""
import argparse
import gc
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "structured-gaussian-splatting_amd")
for _p in (ROOT, PKG):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402

_GC_LOG, _GC_T0 = [], [0.0]


def _gc_cb(phase, info):                     # what the interpreter's cyclic collector takes out of a timed window
    if phase == "start":
        _GC_T0[0] = time.perf_counter()
    else:
        _GC_LOG.append(time.perf_counter() - _GC_T0[0])


gc.callbacks.append(_gc_cb)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec
_AFFINITY = {"original": None, "bound": None}      # the process's CPU mask at start / after the NUMA binding


def _set_affinity_all_threads(cpus):
    """Every thread of this process (the autograd engine's worker exists already) onto `cpus`."""
    if not cpus:
        return
    for tid in os.listdir("/proc/self/task"):
        try:
            os.sched_setaffinity(int(tid), cpus)
        except OSError:
            pass


def algorithmic_bytes(P, V, Vb, Re, N, Tn, K, M, Vlive, sparse_geom, prezeroed, C=3, walked=None, units=0):
    """ALGORITHMIC bytes per step and kernel: SURVEY.md 8d's per-unit figures x the units each kernel processes in THIS
    design.  P Gaussians, V visible, Vb = Gaussians of the depth chunks that were binned (the
    depth prefix), Re = instances actually emitted by the progressive binning (the reference would process R =
    num_rendered of them), Vlive = Gaussians with a non-zero screen-space gradient, N pixels, Tn tiles.  Every logical
    array is charged once per read and once per write, to the kernel that moves it: zero rows are charged to
    `zero_outputs` when the early fill runs (then NOT to geom_bwd), SH coefficients to `chunk_colors` (the lazy colour
    pass reads them for binned Gaussians only), never to `preprocess`."""
    walked = Re if walked is None else walked               # list entries in front of their tile's deepest contributor: what K7 walks
    geom_rows = Vb if sparse_geom else P                    # rows the geometry backward visits
    geom_written = Vlive if prezeroed else geom_rows        # after an early fill only live rows are written
    return {
        "preprocess": 52 * P + 67 * V,                      # K1 minus A.6: read 44 + sort key/value 8, write 67 per visible
        "chunk_colors": (12 * K + 12 + 12 + 1) * Vb,        # A.6 for binned Gaussians: SH + mean in, rgb + clamp mask out
        # depth order by selection (csrc/gsr_select.hip): two histogram passes over the keys (+ tiles / mass), the stable
        # partition by chunk (count pass: keys; scatter pass: keys + tiles in, Gaussian / relative key / tiles out per
        # visible Gaussian), then only the BINNED chunks are sorted: one read + one write of (key, Gaussian) + tiles in,
        # offsets out per binned Gaussian
        "depth_hist": 16 * P, "depth_partition": 8 * P + 20 * V, "chunk_sort": 24 * Vb,
        "scan_tiles": 12 * Vb, "open_count": 8 * Tn,        # chunks beyond the one-block sort: the scan gathers tiles[order[r]] itself
        "count_open": 56 * Vb, "scan_open": 8 * Vb,         # rank -> Gaussian 4 + record 48 + count 4
        "emit": 12 * Re + 56 * Vb,                          # K3: key 4 + slot 4 + Gaussian 4 per instance
        # K4 minimum: one read + one write of (tile, slot); sorts of >= 4 M instances carry the Gaussian word along (12 B each way)
        # and K5 then only reads the sorted keys; smaller ones gather it in K5 (slot in, Gaussian in + out)
        "tile_sort": (24 if Re >= (4 << 20) else 16) * Re,
        "ranges": (4 if Re >= (4 << 20) else 16) * Re + 8 * Tn,
        # gather variant of K3-K5 (chunks of few large splats): counts in, ranges out; then per instance the accept bit's
        # mask word 8 + prefix 4 in, (Gaussian, slot) 8 out, per rank 32 B of metadata
        "tile_ranges": 12 * Tn, "tile_gather": 20 * Re + 32 * Vb,
        # a14: exp / normalize / sigmoid of (3 + 4 + 1) floats per Gaussian: 32 in + 32 out; backward reads the 8 incoming
        # gradients, the saved outputs / raw quaternion (8) and writes 8
        "activations_fwd": 64 * P, "activations_bwd": 96 * P,
        # K6: id 4 + record 36 per instance; 20 B/px; per tile its walked depth and its work units out; one 4 KB checkpoint per
        # kSeg = 128 walked entries (32 B per walked instance)
        "render_fwd": 40 * Re + 20 * N + 12 * Tn + 32 * walked,
        # K7 (front to back, one wave per work unit): 40 read + one whole 48-byte row written per walked instance (an upper bound for the
        # rows: the entries no quadrant group attempted are not written); per unit the 256 pixels' 32 B (T, last, dL/dpix, final colour)
        # + its 16 B/px checkpoint
        "render_bwd": 88 * walked + 256 * 48 * units,
        "reduce_rows": Re + 36 * walked + 36 * Vb,          # deterministic reduction (replaces atomic RMW): a valid byte per row, written rows only
        "geom_bwd": 4 * geom_rows + (99 + 12 * K) * Vlive + (40 + 12 * M) * geom_written,        # K8 + K9
        "loss_fwd": 20 * C * N, "loss_bwd": 24 * C * N,
        "zero_outputs": (56 + 12 * M) * P if prezeroed else 0,       # early fill of the parameter gradients (means3D, means2D, opacity,
                                                                      # scales, rotations, SH); the screen-space tensor is never cleared
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)   # 0.8 ms steps: a 20-step window (16 ms) measured the clock ramp, not the step
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="cfg3", choices=["cfg1", "cfg2", "cfg3", "cfg3n", "cfg3b", "cfg5", "cfg5n"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the non-saturating scene and the second seed")
    ap.add_argument("--no-extras", action="store_true", help="skip the comparison passes (torch getters, fused activations, ...) "
                    "and the profiled pass: the timed region only (for external profilers)")
    ap.add_argument("--force-sharded", action="store_true",
                    help="rehearsal on one GPU: run the N > 1 code path (RCCL process group, ShardedRenderer, slab-local loss) "
                         "with world_size 1")
    ap.add_argument("--train-loop", type=int, default=300, metavar="ITERS",
                    help="also time ITERS iterations of the full training loop (Adam, densify every 100) on the workload: BASELINE "
                         "configs[2] taken literally (0 = skip)")
    ap.add_argument("--no-4k", action="store_true", help="skip the 5M-Gaussian / 3840x2160 non-saturating stress (secondary_4k)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` started by hand: become the launcher.  Nothing in this process has touched the GPU
        # (importing torch and counting devices does not initialise HIP); the ranks are fresh children.
        sys.exit(_self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    n_dev = max(torch.cuda.device_count(), 1)
    shared_device = world > n_dev          # rehearsal on a box with fewer GPUs than ranks: ranks share devices
    torch.cuda.set_device(local_rank % n_dev)
    dev = torch.device("cuda", local_rank % n_dev)
    bound = None
    _AFFINITY["original"] = set(os.sched_getaffinity(0))
    if os.environ.get("GSR_BENCH_BIND", "1") != "0":
        from diff_gaussian_rasterization.hostbind import bind_to_gpu_numa_node
        bound = bind_to_gpu_numa_node(local_rank % n_dev, all_threads=os.environ.get("GSR_BENCH_BIND") == "all")
        _AFFINITY["bound"] = set(os.sched_getaffinity(0)) if bound else None
    dist = None
    collectives = None
    if world > 1 or args.force_sharded:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1:
            os.environ.setdefault("MASTER_PORT", "29533")
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
            collectives = "rccl (world 1)"
        elif shared_device:
            # RCCL refuses two ranks on one device: the rehearsal stages the collectives through gloo (sharded._Comm
            # does that when the group's backend is gloo).  Never the case on the driver's N-GPU node.
            dist.init_process_group("gloo")
            collectives = f"gloo (REHEARSAL: {world} ranks share {n_dev} device(s); not a scaling measurement)"
        else:
            dist.init_process_group("nccl", device_id=dev)
            collectives = "rccl"

    ctx = dict(dev=dev, dist=dist, world=world, rank=rank, sharded_on=(world > 1 or args.force_sharded))
    single = world == 1 and not args.force_sharded
    m = measure(args.workload, args.steps, args.warmup, ctx, extras=single and not args.no_extras, traffic=(args.workload == "cfg3" and world == 1),
                profile=not args.no_extras)

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    line = {
        "metric": "train-step images/sec (fwd+bwd) @1080p, 1e6 Gaussians" if args.workload == "cfg3"
        else f"train-step images/sec (fwd+bwd) {args.workload}",
        "value": m["value"], "unit": "images/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": m["ms_per_step"], "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {**m["config"], "parallelism": "single" if not ctx["sharded_on"] else f"tile-row slabs x{world}",
                   **({"collectives": collectives} if collectives else {})},
        "raster_ms_per_step": m["raster_ms_per_step"], "profiled_ms_per_step": m["profiled_ms_per_step"],
        "gc_in_timed_window": m["gc_in_timed_window"],
        "host": {"bound_to_numa_node": bound[0] if bound else None, "cpus": len(os.sched_getaffinity(0))},
        "kernels": m["kernels"], "roofline": m["roofline"], "cpu_baseline": None,
    }
    for k in ("native_getters", "torch_getters", "fused_activations", "getter_fusion", "reference_loss_composition", "unchanged_caller"):
        if m.get(k) is not None:
            line[k] = m[k]
    if single and not args.no_secondary and args.workload == "cfg3":
        sec = measure("cfg3n", args.steps, args.warmup, ctx, extras=False, traffic=False)
        other = measure("cfg3b", max(args.steps // 2, 2), 2, ctx, extras=False, traffic=False, profile=False)
        line["secondary"] = {
            "what": "the same step on a scene that does NOT saturate: the cfg3 generator with z ~ U(2, 6) and scales x0.5 "
                    "(scene_synth.CONFIGS['cfg3n']); every constant of the library at its default",
            "value": sec["value"], "unit": "images/s", "ms_per_step": sec["ms_per_step"], "config": sec["config"],
            "raster_ms_per_step": sec["raster_ms_per_step"], "kernels": sec["kernels"], "roofline": sec["roofline"],
            "other_seed": {"what": "cfg3's generator, seed 33 instead of 3", "value": other["value"],
                           "ms_per_step": other["ms_per_step"], "config": other["config"]},
        }
    if single and not args.no_4k and args.workload == "cfg3":
        k4 = measure("cfg5n", max(args.steps // 8, 5), 3, ctx, extras=False, traffic=False)
        stream = {k: {"ms_per_step": v["ms_per_step"], "alg_GBs": v["alg_GBs"], "hbm_frac": round((v["alg_GBs"] or 0.0) / HBM_PEAK_GBS, 4)}
                  for k, v in k4["kernels"].items() if k in ("preprocess", "chunk_colors", "geom_bwd", "depth_hist", "depth_partition", "reduce_rows",
                                                             "chunk_sort", "tile_sort", "emit", "ranges", "loss_fwd", "loss_bwd")}
        line["secondary_4k"] = {
            "what": "BASELINE configs[4]'s size (5M Gaussians, 3840x2160, SH 3) on ONE GPU with a generator that does not saturate "
                    "(scene_synth.CONFIGS['cfg5n']: the cfg3n recipe): the streaming kernels at the size the config names; hbm_frac = "
                    "algorithmic bytes / time / 8 TB/s per kernel; PMC traffic of the same run: profiles/r03_cfg5n_traffic.json",
            "value": k4["value"], "unit": "images/s", "ms_per_step": k4["ms_per_step"], "steps": max(args.steps // 8, 5), "config": k4["config"],
            "raster_ms_per_step": k4["raster_ms_per_step"], "kernels": k4["kernels"], "streaming": stream, "roofline": k4["roofline"],
            "pmc_traffic": _profiles_json("r03_cfg5n_traffic.json")}
    if single and args.train_loop > 0 and args.workload in ("cfg3", "cfg2", "cfg3n"):
        line["train_loop"] = _train_loop(args.workload, args.train_loop, dev, False)
        if os.environ.get("GSR_BENCH_TRAIN_BOTH"):
            line["train_loop"]["native_getters"] = _train_loop(args.workload, args.train_loop, dev, True)
    if world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = _cpu_baseline(args.workload)
    print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


def measure(workload, steps, warmup, ctx, extras, traffic, profile=True):
    """Time `steps` steps of the training window on `workload`; returns the numbers of one bench object."""
    import scene_synth as S
    from diff_gaussian_rasterization import _native as N
    from gaussian_params import GaussianParams, Pipe
    from gaussian_renderer import render
    from loss_utils import training_loss
    dev, dist, world, rank = ctx["dev"], ctx["dist"], ctx["world"], ctx["rank"]

    cfg = S.CONFIGS[workload]
    scene, cam = S.make_config(workload)
    scene, cam = scene.to(dev), cam.to(dev)
    # The parameter store is this package's scene.GaussianModel: the reference class's public surface (getters, optimizer
    # groups, densification) with the getters as native ops.  `store` is the plain torch-getter
    # store (the reference's own four torch getters, what a caller who keeps the reference's GaussianModel class runs);
    # it is timed beside `value` as "torch_getters".
    from scene import GaussianModel
    model = GaussianModel(scene.sh_degree)
    model.adopt_scene(scene, device=dev)
    store = GaussianParams(scene).to(dev)
    active = {"m": model}
    params = list(model._t.values()) + [p for p in store.parameters()]
    bg = torch.zeros(3, device=dev)
    gt = torch.rand(3, cfg["H"], cfg["W"], generator=torch.Generator().manual_seed(cfg["seed"] + 100)).to(dev)
    pipe = Pipe()
    sharded = None
    if ctx["sharded_on"]:
        from diff_gaussian_rasterization.sharded import ShardedRenderer
        sharded = ShardedRenderer(dist, world, rank)
    loss_fn = {"f": None}

    def step():
        for p in params:
            p.grad = None
        out = render(cam, active["m"], pipe, bg) if sharded is None else sharded.render(cam, active["m"], pipe, bg)
        # same loss; on N > 1 every rank evaluates the terms of its own rows (slab-local, one 8-byte all-reduce)
        if loss_fn["f"] is not None:
            loss = loss_fn["f"](out["render"], gt)
        else:
            loss = training_loss(out["render"], gt) if sharded is None else sharded.training_loss(out["render"], gt)
        loss.backward()
        return out

    def sync():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    gc_log, last_gc = _GC_LOG, {}

    def timed(n, w):
        gc.collect()                 # a generation-2 collection landing inside a 30 ms window is a 40 ms outlier (measured);
        for _ in range(w):           # collect before the warm-up, so that the device is busy again when the clock starts
            step()
        sync()
        gc_log.clear()
        t0 = time.perf_counter()
        per_step = []
        for _ in range(n):
            o = step()
            if os.environ.get("GSR_BENCH_TRACE"):
                torch.cuda.synchronize(dev); per_step.append(round(1e3 * (time.perf_counter() - t0), 2))
        sync()
        elapsed = time.perf_counter() - t0
        last_gc["collections"], last_gc["ms"] = len(gc_log), round(1e3 * sum(gc_log), 3)
        if per_step:
            print("trace", per_step, file=sys.stderr)
        return elapsed, o

    # (1) the timed region: EXACTLY K steps, no instrumentation inside
    elapsed, out = timed(steps, warmup)
    gc_headline = dict(last_gc)
    # (2) the same K steps again with the library's per-kernel hipEvent pairs (roofline); the event packets
    # add ~10 us between kernels, so this pass is reported separately and never feeds `value`
    prof, elapsed_profiled = {}, 0.0
    if profile:
        N.profile_enable(True)
        elapsed_profiled, out = timed(steps, 0)
        prof = N.profile_read()
        N.profile_enable(False)
    res = {}
    if extras:
        # (2a) the timed pass renders this package's GaussianModel from its raw leaves (activations inside the kernels: render()'s
        # default for that class); the same model through its native getters (one launch forward, one backward) beside it
        pipe.fused_activations = False
        el, _ = timed(steps, max(warmup, 1))
        pipe.fused_activations = None
        res["native_getters"] = {
            "value": round(steps / el, 3), "unit": "images/s", "ms_per_step": round(1e3 * el / steps, 4),
            "what": "same step, same scene.GaussianModel, with render(..., pipe.fused_activations=False): exp / normalize / sigmoid as "
                    "the model's native getters (csrc/gsr_activations.hip, one launch forward and one backward over all P) instead "
                    "of inside the preprocess / geometry-backward kernels"}
        # (2b) the same step with the reference's torch getters (exp / normalize / sigmoid / cat and their autograd backward)
        active["m"] = store
        el, _ = timed(steps, max(warmup, 1))
        res["torch_getters"] = {
            "value": round(steps / el, 3), "unit": "images/s", "ms_per_step": round(1e3 * el / steps, 4),
            "what": "same step with a parameter store whose getters are the reference's torch ops (gaussian_params.GaussianParams "
                    "= scene/gaussian_model.py:101-125 verbatim): what a caller who keeps the reference's own GaussianModel class "
                    "gets from the drop-in rasterizer alone"}
        # (3) extension, reported beside `value`, never as it: the activations of the parameter store
        # fused into the HIP kernels (pipe.fused_activations -> GaussianRasterizer.forward_raw)
        pipe.fused_activations = True
        el, _ = timed(steps, max(warmup, 1))
        pipe.fused_activations = None
        res["fused_activations"] = {
            "value": round(steps / el, 3), "unit": "images/s", "ms_per_step": round(1e3 * el / steps, 4),
            "what": "the torch-getter store of (2b) with render(..., pipe.fused_activations=True): its raw tensors (features_dc and "
                    "features_rest as two tensors) go into the kernels; same image and parameter gradients"}
        # (4) the SAME caller code as the timed pass with the rasterizer's opt-in FUSE_GETTERS
        import diff_gaussian_rasterization as _dgr
        _dgr.FUSE_GETTERS = True
        el, _ = timed(steps, max(warmup, 1))
        _dgr.FUSE_GETTERS = False
        active["m"] = model
        res["getter_fusion"] = {
            "value": round(steps / el, 3), "unit": "images/s", "ms_per_step": round(1e3 * el / steps, 4),
            "what": "unchanged caller (reference-style render(): getters + GaussianRasterizer.forward) with "
                    "diff_gaussian_rasterization.FUSE_GETTERS = True (or GSR_FUSE_GETTERS=1); opt-in: matches autograd node "
                    "names, falls back to the plain path on any mismatch"}
        # (5) the reference's own loss composition, train.py:104-105 verbatim: Ll1 = l1_loss(image, gt);
        # loss = (1 - lambda) * Ll1 + lambda * (1 - ssim(image, gt)) with this package's drop-in l1_loss / ssim
        import loss_utils
        if hasattr(loss_utils, "ssim_torch"):
            loss_fn["f"] = lambda image, g: (1.0 - 0.2) * loss_utils.l1_loss(image, g) + 0.2 * (1.0 - loss_utils.ssim(image, g))
            el, _ = timed(steps, max(warmup, 1))
            loss_fn["f"] = lambda image, g: loss_utils.training_loss_torch(image, g)
            n_torch = max(steps // 2, 2)
            el_torch, _ = timed(n_torch, 1)
            loss_fn["f"] = None
            res["reference_loss_composition"] = {
                "value": round(steps / el, 3), "unit": "images/s", "ms_per_step": round(1e3 * el / steps, 4),
                "what": "train.py:104-105 as written (two calls: l1_loss, ssim) with loss_utils' drop-in autograd ops over "
                        "the fused kernels; torch_ops_ms_per_step = the same composition in plain torch ops (5 depthwise "
                        "convolutions and their backward), what an import of the reference's own utils/loss_utils.py gives",
                "torch_ops_ms_per_step": round(1e3 * el_torch / n_torch, 4)}
    if extras:
        # (6) ONE number for the truly unchanged caller (north_star: "consume it unchanged"): the reference's own parameter store
        # (torch getters, scene/gaussian_model.py:101-125), its render() call sequence (gaussian_renderer/__init__.py:53-93), its
        # two-call loss (train.py:104-105 over utils/loss_utils.py:17-63 signatures) — and no host placement: the NUMA binding of
        # this process is undone for the leg (a caller who changes nothing does not bind either).
        import loss_utils
        active["m"] = store
        loss_fn["f"] = lambda image, g: (1.0 - 0.2) * loss_utils.l1_loss(image, g) + 0.2 * (1.0 - loss_utils.ssim(image, g))
        _set_affinity_all_threads(_AFFINITY["original"])
        el, _ = timed(steps, max(warmup, 1))
        _set_affinity_all_threads(_AFFINITY["bound"] or _AFFINITY["original"])
        loss_fn["f"] = None
        active["m"] = model
        res["unchanged_caller"] = {
            "value": round(steps / el, 3), "unit": "images/s", "ms_per_step": round(1e3 * el / steps, 4),
            "what": "the reference's caller, nothing changed but the import: GaussianParams (torch exp / normalize / sigmoid / cat getters "
                    "and their autograd backward) -> render() -> GaussianRasterizer.forward; loss = (1 - 0.2) * l1_loss(image, gt) + 0.2 * "
                    "(1 - ssim(image, gt)) as two calls; the process NOT bound to the GPU's NUMA node"}
    if dist is not None:
        t = torch.tensor([elapsed], device="cpu" if dist.get_backend() == "gloo" else dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- workload statistics for the roofline (measured, not assumed)
    radii = out["radii"]
    P = scene.P
    V = int((radii > 0).sum())
    Npix, Tn = cfg["W"] * cfg["H"], ((cfg["W"] + 15) // 16) * ((cfg["H"] + 15) // 16)
    K = M = (cfg["D"] + 1) ** 2
    stats = _frame_stats(model, cam, bg, pipe)
    R, Re, Vlive, pairs = stats["num_rendered"], stats["emitted"], stats["live"], stats["pairs_bwd"]
    Vb = stats["binned_ranks"]
    sparse = N.effective_binned_ranks(stats["plan"]) * 4 < P
    alg = algorithmic_bytes(P, V, Vb, Re, Npix, Tn, K, M, Vlive, sparse_geom=sparse, prezeroed=sparse, walked=stats["walked"], units=stats["units"])
    per_kernel = {}
    for k, (ms, n) in prof.items():
        per_step_ms = ms / steps
        per_kernel[k] = dict(ms_per_step=per_step_ms, launches_per_step=n / steps,
                             alg_GBs=(alg.get(k, 0) / 1e9) / (per_step_ms / 1e3) if per_step_ms > 0 else None)
        keys = {"chunk_sort": Vb, "tile_sort": Re}.get(k)            # SURVEY 8d secondary rate for K4: keys/s
        if keys is not None and per_step_ms > 0:
            per_kernel[k]["Gkeys_per_s"] = keys / 1e9 / (per_step_ms / 1e3)
    # no kernel may be credited with more algorithmic bytes than the HBM could move in its time (an accounting check: reported,
    # never raised — the line must always come out)
    over = sorted(k for k, v in per_kernel.items() if v["alg_GBs"] is not None and v["alg_GBs"] > HBM_PEAK_GBS)
    raster_ms = sum(ms for ms, _ in prof.values()) / steps
    roofline = None
    if prof:
        dom = max(prof, key=lambda k: prof[k][0])
        launches = max(prof[dom][1], 1)
        dom_ms = prof[dom][0] / launches                              # average duration of one launch
        bytes_per_launch = alg.get(dom, 0) * steps / launches
        achieved = (bytes_per_launch / 1e9) / (dom_ms / 1e3)
        bwd_ms = per_kernel.get("render_bwd", {}).get("ms_per_step", 0)
        ref_bytes = (144 + 12 * M) * P + (182 + 24 * K) * V + 160 * R + 40 * Npix + 8 * Tn
        roofline = dict(bound="hbm", kernel=dom, achieved=round(achieved, 2), peak=HBM_PEAK_GBS, unit="GB/s",
                        frac=round(achieved / HBM_PEAK_GBS, 5), traffic=_traffic_from_profiles(dom) if traffic else None,
                        avg_launch_ms=round(dom_ms, 4), algorithmic_bytes_per_launch=int(bytes_per_launch),
                        step_algorithmic_bytes=int(sum(alg.values())), kernels_over_hbm_peak=over,
                        step_frac=round(sum(alg.values()) / 1e9 / (elapsed / steps) / HBM_PEAK_GBS, 5),
                        # context only: SURVEY 8d's whole-step total evaluated as the REFERENCE's algorithm would move it
                        # (all R = num_rendered duplicates; this design bins `instances_emitted` of them), over the same time
                        reference_formula_step_bytes=int(ref_bytes),
                        reference_formula_step_frac=round(ref_bytes / 1e9 / (elapsed / steps) / HBM_PEAK_GBS, 5),
                        valu=_valu_roofline(dom, dom_ms, workload) if (traffic or workload == "cfg3n") else None,
                        pairs_per_s_G=round(pairs / 1e9 / (bwd_ms / 1e3), 1) if bwd_ms > 0 else None,
                        note="the blend kernels are VALU-issue-bound, not HBM-bound (SURVEY 8d caveat): `valu` is their "
                             "roofline; pairs_per_s_G = (pixel, splat) pairs per second in render_bwd")
    res.update({
        "value": round(steps / elapsed, 3), "ms_per_step": round(1e3 * elapsed / steps, 4),
        "config": {"workload": f"{workload}: {P} Gaussians, {cfg['W']}x{cfg['H']}, SH degree {cfg['D']} "
                               f"(SURVEY Appendix B generator, seed {cfg['seed']}"
                               + (f", z >= {cfg['zmin']}, scales x{cfg['scale_mul']}" if "zmin" in cfg else "")
                               + "); step = render() + L1/D-SSIM loss + backward (train.py:79-108 window)",
                   "visible": V, "num_rendered": R, "instances_emitted": Re, "chunks_run": stats["chunks_run"],
                   "chunks_planned": stats["chunks_planned"], "binned_gaussians": Vb, "gaussians_with_gradient": Vlive,
                   "instances_walked_bwd": stats["walked"], "bwd_work_units": stats["units"],
                   "emitted_over_num_rendered": round(Re / max(R, 1), 4)},
        "raster_ms_per_step": round(raster_ms, 4), "profiled_ms_per_step": round(1e3 * elapsed_profiled / steps, 4),
        "gc_in_timed_window": gc_headline,
        "kernels": {k: {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items()} for k, v in per_kernel.items()},
        "roofline": roofline,
    })
    del model, params, scene, gt, out
    torch.cuda.empty_cache()
    return res


def _self_launch(n: int) -> int:
    """`python bench.py --gpus N` without a launcher: start `python -m torch.distributed.run --nproc-per-node N bench.py
    <same arguments>` as a CHILD process (never exec: see the GPU box rules) and relay its output; rank 0 of the child
    job prints the JSON line.  Returns the child's exit code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC only on this pool (RCCL across processes)
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for ln in p.stdout.splitlines():
        print(ln, flush=True)
    return p.returncode


def _frame_stats(model, cam, bg, pipe):
    """Workload statistics of the benchmark frame (one extra un-timed forward/backward through the functional
    API): R, instances emitted, chunks run, Gaussians with a non-zero screen gradient, (pixel, splat) pairs."""
    import math

    import diff_gaussian_rasterization as dgr
    from diff_gaussian_rasterization import _native as N
    with torch.no_grad():
        rs = dgr.GaussianRasterizationSettings(cam.image_height, cam.image_width, math.tan(cam.FoVx * .5),
                                               math.tan(cam.FoVy * .5), bg, 1.0, cam.world_view_transform,
                                               cam.full_proj_transform, model.active_sh_degree, cam.camera_center,
                                               False, False)
        _, _, fr = dgr.rasterize_forward(model.get_xyz, model.get_features, None, model.get_opacity, model.get_scaling,
                                         model.get_rotation, None, rs)
        v = N.debug_views(fr.desc, fr.geom_ws, fr.binning_ws, fr.image_ws, fr.plan)
        rng = v["ranges"].long()[:fr.plan.chunks_run]
        lens = rng[..., 1] - rng[..., 0]
        emitted = int(lens.sum())
        # pairs the backward evaluates: per pixel, its last contributor's position in the concatenated list
        enc = v["n_contrib"].long()
        c = (enc >> N.LAST_SHIFT) - 1
        pos = enc & ((1 << N.LAST_SHIFT) - 1)
        H, W = enc.shape
        Gx = (W + 15) // 16
        ys, xs = torch.meshgrid(torch.arange(H, device=enc.device), torch.arange(W, device=enc.device), indexing="ij")
        tile = (ys // 16) * Gx + xs // 16
        before = torch.cat([torch.zeros_like(lens[:1]), torch.cumsum(lens, 0)], 0)
        ncontrib = torch.where(c >= 0, before[c.clamp(min=0), tile] + pos, torch.zeros_like(pos))
        g = torch.ones(3, H, W, device=enc.device)
        screen = dgr.rasterize_backward_screen(fr, g)
        live = int((screen.abs().sum(1) > 0).sum())
        walked = int(torch.minimum(v["tile_walk"].long()[:fr.plan.chunks_run], lens).sum())
        units = int(v["bwd_unit_count"].long().sum())
        binned = int(fr.plan.chunk_rank_begin[fr.plan.chunks_run]) if fr.plan.num_rendered > 0 and fr.plan.chunks_run > 0 else 0
    return dict(num_rendered=fr.R, emitted=emitted, chunks_run=int(fr.plan.chunks_run), chunks_planned=int(fr.plan.num_chunks),
                live=live, pairs_bwd=int(ncontrib.sum()), binned_ranks=binned, walked=walked, units=units, plan=fr.plan)


def _train_loop(workload, iters, dev, native_getters):
    """BASELINE configs[2] taken literally: the full train.py loop (LR schedule, render, loss, backward,
    densification statistics, densify/prune every 100 iterations, Adam) starting from the workload's cloud, with
    target views rendered from a second cloud (seed 30) on 8 cameras of a small arc."""
    from dataclasses import replace

    import scene_synth as S
    from gaussian_params import Pipe
    from gaussian_renderer import render
    from scene import GaussianModel, OptimizationDefaults
    from train_loop import train
    cfg = S.CONFIGS[workload]
    cams = [c.to(dev) for c in S.arc_cameras(cfg["W"], cfg["H"], 8)]
    bg = torch.zeros(3, device=dev)
    truth = GaussianModel(cfg["D"])
    truth.adopt_scene(S.make_scene(cfg["P"], cfg["W"], cfg["H"], cfg["D"], 30), device=dev)
    with torch.no_grad():
        targets = [render(c, truth, Pipe(), bg)["render"].clone() for c in cams]
    del truth
    gm = GaussianModel(cfg["D"])
    gm.adopt_scene(S.make_config(workload)[0], device=dev)
    opt = replace(OptimizationDefaults(), densify_from_iter=0)
    gm.training_setup(opt)
    warm = min(20, iters // 4)
    # one-time library initialisation (rocBLAS for the split's bmm, torch's RNG / index kernels: ~0.5 s on first use)
    # on a throw-away 2000-Gaussian model, so that the timed loop's first densification costs what every later one costs
    tiny = GaussianModel(0)
    tiny.adopt_scene(S.make_scene(2000, 64, 64, 0, 1), device=dev)
    tiny.training_setup(opt)
    tiny.xyz_gradient_accum += 1.0
    tiny.denom += 1.0
    with torch.no_grad():
        tiny.densify_and_prune(opt.densify_grad_threshold, 0.005, 6.0, None)
    del tiny
    pipe = Pipe()
    pipe.fused_activations = False if native_getters else None      # None: render()'s default for scene.GaussianModel (raw leaves)
    train(gm, cams, targets, opt, pipe, bg, iterations=warm, scene_extent=6.0)
    with torch.no_grad():       # warm-up includes one full-size densification: the allocator has seen the grown tensors
        gm.densify_and_prune(opt.densify_grad_threshold, 0.005, 6.0, None)
    gc.collect()                 # as in timed(): a full collection (48 ms over torch's import-time objects) is not the loop's cost
    torch.cuda.synchronize(dev)
    n0 = gm._xyz.shape[0]
    _GC_LOG.clear()
    stamps = []
    stat0 = torch.cuda.memory_stats(dev)
    t0 = time.perf_counter()
    train(gm, cams, targets, opt, pipe, bg, iterations=warm + iters, first_iter=warm + 1, scene_extent=6.0,
          on_iteration=lambda it, loss, g: stamps.append(time.perf_counter()))          # host time per iteration (no sync)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    gc_ms = round(1e3 * sum(_GC_LOG), 2)
    stat1 = torch.cuda.memory_stats(dev)
    per_it = [1e3 * (b - a) for a, b in zip([t0] + stamps[:-1], stamps)]
    slow = {str(warm + 1 + i): round(x, 1) for i, x in enumerate(per_it) if x > 20.0}
    return {"iterations": iters, "its_per_s": round(iters / dt, 2), "ms_per_it": round(1e3 * dt / iters, 3), "gc_ms_in_window": gc_ms,
            "host_ms_per_it_median": round(sorted(per_it)[len(per_it) // 2], 3), "iterations_over_20_ms": slow,
            "device_allocations_in_window": int(stat1["num_device_alloc"] - stat0["num_device_alloc"]),
            "gaussians_start": int(n0), "gaussians_end": int(gm._xyz.shape[0]),
            "includes": "LR schedule, render, L1/D-SSIM, backward, densification stats, densify+prune every 100 it, Adam"}


def _profiles_json(name):
    path = os.path.join(ROOT, "profiles", name)
    if os.path.exists(path):
        try:
            return json.load(open(path))
        except Exception:
            return None
    return None


def _traffic_from_profiles(kernel):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/traffic.json: FETCH_SIZE x2 +
    WRITE_SIZE, made by profiles/pmc_to_traffic.py for the default cfg3 single-GPU run), or None."""
    d = _profiles_json("traffic.json")
    return d.get(kernel) if d else None


def _valu_roofline(kernel, launch_ms, workload):
    """VALU-issue roofline of the dominant (blend) kernel.  Inputs, both committed under profiles/ and measured on MI355X:
      * valu_microbench/r03_valu_rates.json — issue cycles one wave64 instruction of each class costs its SIMD at
        saturation (profiles/valu_microbench/valu_microbench.hip: v_mul/add 2.2, v_fma 3.6, packed fp32 / compares /
        selects / min / max / DPP 4.1-4.2, transcendentals 8.1 cycles);
      * r03_valu_mix.json (r03_cfg3n_valu_mix.json for the secondary workload) — for the default cfg3 run: the kernel's VALU wave-instructions per launch by class (rocprofv3
        PMC SQ_INSTS_VALU_* + the packed share of each class from the disassembly), made by profiles/valu_mix.py.
    required = sum over classes of instructions x issue cycles; available = SIMDs x cycles of the launch.
    `frac` = required / available: 1.0 would mean every SIMD issues a VALU instruction on every cycle of the launch."""
    rates = _profiles_json("valu_microbench/r03_valu_rates.json") or _profiles_json("valu_microbench/r02_valu_rates.json")
    mix = _profiles_json("r03_cfg3n_valu_mix.json" if workload == "cfg3n" else "r03_valu_mix.json")
    if not rates or not mix or kernel not in mix.get("kernels", {}) or mix.get("workload", "cfg3") != workload:
        return None
    k = mix["kernels"][kernel]
    clock_hz = float(rates.get("clock_hz", 2.4e9))
    simds = int(rates.get("simds", 1024))
    cyc = {name: r["cycles_per_instruction"] for name, r in rates["rates"].items()}
    required = sum(k["insts_valu"] * share * cyc.get(cls, cyc["v_fma_f32"]) for cls, share in k["class_share"].items())
    available = simds * launch_ms * 1e-3 * clock_hz
    return {"kernel": kernel, "unit": "VALU issue cycles per launch", "achieved": int(required), "peak": int(available),
            "frac": round(required / available, 4), "insts_valu_per_launch": int(k["insts_valu"]),
            "clock_hz": clock_hz, "class_share": k["class_share"],
            "issue_cycles_per_class": {c: cyc[c] for c in k["class_share"] if c in cyc},
            "note": "peak assumes the microbenchmark's clock; the blend kernels also issue LDS reads, scalar branches and "
                    "s_waitcnt on the same SIMD"}


def _cpu_baseline(workload):
    """The CPU oracle (oracle/, a plain-C port of the algorithm; the reference has no CPU rasterizer:
    SURVEY F4) timed on the host cores: ONE frame of the same workload, rasterizer forward+backward."""
    import numpy as np

    import oracle
    import scene_synth as S
    oracle.build()
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import raster_kwargs
    cfg = S.CONFIGS[workload]
    scene, cam = S.make_config(workload)
    kw = raster_kwargs(scene, cam)
    a = scene.activated()
    g = S.make_grad_image(cfg["W"], cfg["H"], cfg["seed"]).numpy()
    cores = os.cpu_count() or 1
    t0 = time.perf_counter()
    fr = oracle.rasterize(dtype=np.float32, parallel=True, **kw)
    fr.backward(g, parallel=True)
    dt = time.perf_counter() - t0
    # SURVEY 8d also asks for the reference-semantics Python SH path (the `convert_SHs_python` twin, utils/sh_utils.py
    # eval_sh + clamp) on the host cores for the same P: this repo's torch mirror of it, forward only
    from gaussian_renderer import eval_sh
    sh_view = a["shs"].transpose(1, 2).contiguous()
    dirs = torch.nn.functional.normalize(a["means3D"] - cam.camera_center[None, :], dim=1)
    t1 = time.perf_counter()
    torch.clamp_min(eval_sh(cfg["D"], sh_view, dirs) + 0.5, 0.0)
    sh_ms = 1e3 * (time.perf_counter() - t1)
    return {"value": round(1.0 / dt, 4), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"1 frame of the same workload, rasterizer forward+backward only (no loss), C oracle with "
                      f"OpenMP over {cores} host threads (sort and per-Gaussian stages serial); {dt:.1f} s",
            "python_sh_path_ms": round(sh_ms, 1), "torch_threads": torch.get_num_threads(),
            "cfg1": _cpu_baseline_cfg1()}


def _cpu_baseline_cfg1():
    """BASELINE.json configs[0] as BASELINE.md section 3 planned it: 10 k Gaussians, SH degree 0, 256x256, CPU rasterize
    FORWARD only (plumbing) — the C oracle, 3 warm-up + 10 timed frames, single thread and all threads."""
    import numpy as np

    import oracle
    import scene_synth as S
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from util import raster_kwargs
    scene, cam = S.make_config("cfg1")
    kw = raster_kwargs(scene, cam)
    out = {"workload": "cfg1: 10000 Gaussians, SH degree 0, 256x256, forward only", "kind": "port", "unit": "images/s"}
    for label, par in (("1_thread", False), ("all_threads", True)):
        ts = []
        for i in range(13):
            t0 = time.perf_counter()
            oracle.rasterize(dtype=np.float32, parallel=par, **kw)
            if i >= 3:
                ts.append(time.perf_counter() - t0)
        ts.sort()
        out[label] = {"median_ms": round(1e3 * ts[len(ts) // 2], 2), "min_ms": round(1e3 * ts[0], 2),
                      "value": round(1.0 / ts[len(ts) // 2], 2)}
    return out


if __name__ == "__main__":
    main()
