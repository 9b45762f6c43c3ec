#!/usr/bin/env python3
"""bench.py — train-step images/s (fwd+bwd) of the Gaussian-rasterizer hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg3] [--no-cpu-baseline]

A "step" is the reference's timed training window (train.py:79-108): render() through the drop-in
rasterizer + L1/D-SSIM loss + backward into the leaf parameters, on BASELINE.json configs[2]'s scene
(1e6 Gaussians, 1920x1080, SH degree 3; SURVEY Appendix B, seed 3).  Inputs are resident in HBM before
the timed region.  For N > 1 (launched by torch.distributed.run, one rank per GPU) the image is split
into tile-row slabs: the scene is fixed and each rank renders its slab of the SAME image, so
scaling is "strong".

Prints ONE JSON line on rank 0 (see the keys at the bottom).  The roofline object is for the dominant
kernel of the step, timed live with hipEvents inside libgsrast.so; cpu_baseline times the CPU oracle
(a port, test infrastructure) on rank 0 at N = 1 only.
"""
import argparse
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

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s spec


def algorithmic_bytes(P, V, Re, N, Tn, K, M, Vlive, C=3):
    """ALGORITHMIC bytes per step and kernel: SURVEY.md 8d's per-unit figures x the units each kernel
    processes in THIS design.  Re = instances actually emitted by the progressive
    binning (the reference would process R = num_rendered of them), Vlive = Gaussians with a non-zero
    screen-space gradient.  Keys are the library's profile names; values are totals per step."""
    return {
        "preprocess": 52 * P + (12 * K + 67) * V,          # K1 (+ 8 B/Gaussian depth-sort key/value)
        "depth_sort": 16 * P,                               # minimum: one read + one write of (key, value)
        "scan_tiles": 12 * P, "chunk_plan": 0, "open_count": 8 * Tn,    # the scan gathers tiles[order[r]] itself
        "count_open": 56 * V, "scan_open": 8 * V,
        "emit": 12 * Re,                                    # K3: key 4 + slot 4 + Gaussian 4 per instance
        "tile_sort": 16 * Re,                               # K4 minimum: one read + one write of (tile, slot)
        "ranges": 16 * Re + 8 * Tn,                         # K5 (+ sorted position -> Gaussian)
        "render_fwd": 40 * Re + 20 * N,                     # K6: id 4 + record 36 per instance; 20 B/px
        "render_bwd": 76 * Re + 20 * N,                     # K7: 40 read + 36 written per instance; 20 B/px
        "reduce_rows": 36 * Re + 36 * P,                    # deterministic reduction (replaces atomic RMW)
        "geom_bwd": 4 * P + (99 + 12 * K) * Vlive + (40 + 12 * M) * P,   # K8 + K9
        "loss_fwd": 20 * C * N, "loss_bwd": 24 * C * N,
        "zero_outputs": (104 + 12 * M) * P,                # early zero fill: screen-space (48 B) + parameter gradients
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg3", choices=["cfg1", "cfg2", "cfg3", "cfg5"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-sharded", action="store_true",
                    help="rehearsal on one GPU: run the N > 1 code path (RCCL process group, ShardedRenderer, slab-local loss) "
                         "with world_size 1")
    ap.add_argument("--train-loop", type=int, default=0, metavar="ITERS",
                    help="also time ITERS iterations of the full training loop (Adam, densify every 100) on the workload")
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

    import scene_synth as S
    from diff_gaussian_rasterization import _native as N
    from gaussian_params import GaussianParams, Pipe
    from gaussian_renderer import render
    from loss_utils import training_loss

    cfg = S.CONFIGS[args.workload]
    scene, cam = S.make_config(args.workload)
    scene, cam = scene.to(dev), cam.to(dev)
    model = GaussianParams(scene).to(dev)
    params = [p for p in model.parameters()]
    bg = torch.zeros(3, device=dev)
    gt = torch.rand(3, cfg["H"], cfg["W"], generator=torch.Generator().manual_seed(cfg["seed"] + 100)).to(dev)
    pipe = Pipe()

    if world > 1 or args.force_sharded:
        from diff_gaussian_rasterization.sharded import ShardedRenderer
        sharded = ShardedRenderer(dist, world, rank)
    else:
        sharded = None

    def step():
        for p in params:
            p.grad = None
        if sharded is None:
            out = render(cam, model, pipe, bg)
        else:
            out = sharded.render(cam, model, pipe, bg)
        # same loss; on N > 1 every rank evaluates the terms of its own rows (slab-local, one 8-byte all-reduce)
        loss = training_loss(out["render"], gt) if sharded is None else sharded.training_loss(out["render"], gt)
        loss.backward()
        return out

    def sync():
        torch.cuda.synchronize(dev)
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    sync()
    # (1) the timed region: EXACTLY K steps, no instrumentation inside
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    sync()
    elapsed = time.perf_counter() - t0
    # (2) the same K steps again with the library's per-kernel hipEvent pairs (roofline); the event packets
    # add ~10 us between kernels, so this pass is reported separately and never feeds `value`
    N.profile_enable(True)
    t1 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    sync()
    elapsed_profiled = time.perf_counter() - t1
    prof = N.profile_read()
    N.profile_enable(False)
    # (3) extension, reported beside `value`, never as it: the same K steps with the activations of the parameter
    # store (exp / sigmoid / normalize / cat and their backward) fused into the HIP kernels
    # (pipe.fused_activations -> GaussianRasterizer.forward_raw) instead of running as ~30 torch kernels
    fused = None
    if sharded is None:
        pipe.fused_activations = True
        for _ in range(max(args.warmup, 1)):
            step()
        sync()
        t2 = time.perf_counter()
        for _ in range(args.steps):
            step()
        sync()
        el = time.perf_counter() - t2
        pipe.fused_activations = False
        fused = {"value": round(args.steps / el, 3), "unit": "images/s", "ms_per_step": round(1e3 * el / args.steps, 4),
                 "what": "same step, activations fused into preprocess / geometry-backward kernels (extension beyond the "
                         "reference API: render(..., pipe.fused_activations=True)); same image and parameter gradients"}
    # (4) extension, also reported beside `value`: the SAME caller code as the timed pass (getters + standard forward()),
    # with the rasterizer's opt-in FUSE_GETTERS: it recognises the getters in the arguments' autograd history and renders
    # from the leaves, so the getters' backward kernels (and cat's copies) never run
    getter_fusion = None
    if sharded is None:
        import diff_gaussian_rasterization as _dgr
        _dgr.FUSE_GETTERS = True
        for _ in range(max(args.warmup, 1)):
            step()
        sync()
        t3 = time.perf_counter()
        for _ in range(args.steps):
            step()
        sync()
        el = time.perf_counter() - t3
        _dgr.FUSE_GETTERS = False
        getter_fusion = {"value": round(args.steps / el, 3), "unit": "images/s", "ms_per_step": round(1e3 * el / args.steps, 4),
                         "what": "unchanged caller (reference-style render(): getters + GaussianRasterizer.forward) with "
                                 "diff_gaussian_rasterization.FUSE_GETTERS = True (or GSR_FUSE_GETTERS=1)"}
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
    alg = algorithmic_bytes(P, V, Re, Npix, Tn, K, M, Vlive)
    per_kernel = {}
    for k, (ms, n) in prof.items():
        per_step_ms = ms / args.steps
        per_kernel[k] = dict(ms_per_step=per_step_ms, launches_per_step=n / args.steps,
                             alg_GBs=(alg.get(k, 0) / 1e9) / (per_step_ms / 1e3) if per_step_ms > 0 else None)
        keys = {"depth_sort": P, "tile_sort": Re}.get(k)             # SURVEY 8d secondary rate for K4: keys/s
        if keys is not None and per_step_ms > 0:
            per_kernel[k]["Gkeys_per_s"] = keys / 1e9 / (per_step_ms / 1e3)
    raster_ms = sum(ms for ms, _ in prof.values()) / args.steps
    dom = max(prof, key=lambda k: prof[k][0])
    launches = max(prof[dom][1], 1)
    dom_ms = prof[dom][0] / launches                              # average duration of one launch
    bytes_per_launch = alg.get(dom, 0) * args.steps / launches
    achieved = (bytes_per_launch / 1e9) / (dom_ms / 1e3)
    roofline = dict(bound="hbm", kernel=dom, achieved=round(achieved, 2), peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=round(achieved / HBM_PEAK_GBS, 5), traffic=_traffic_from_profiles(dom) if (args.workload == "cfg3" and world == 1) else None,
                    avg_launch_ms=round(dom_ms, 4), algorithmic_bytes_per_launch=int(bytes_per_launch),
                    step_algorithmic_bytes=int(sum(alg.values())),
                    step_frac=round(sum(alg.values()) / 1e9 / (elapsed / args.steps) / HBM_PEAK_GBS, 5),
                    # context only: SURVEY 8d's whole-step total evaluated as the REFERENCE's algorithm would move it, i.e.
                    # with all R = num_rendered duplicates (this design bins `instances_emitted` of them), over the same time
                    reference_formula_step_bytes=int((144 + 12 * M) * P + (182 + 24 * K) * V + 160 * R + 40 * Npix + 8 * Tn),
                    reference_formula_step_frac=round(((144 + 12 * M) * P + (182 + 24 * K) * V + 160 * R + 40 * Npix + 8 * Tn)
                                                      / 1e9 / (elapsed / args.steps) / HBM_PEAK_GBS, 5),
                    note="blend kernels are VALU-bound (SURVEY 8d caveat): secondary rate = "
                         f"{pairs / 1e9 / (per_kernel.get('render_bwd', {}).get('ms_per_step', 0) / 1e3 + 1e-12):.1f} "
                         "G (pixel,splat) pairs/s in render_bwd")

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    train_loop = None
    if world == 1 and args.train_loop > 0:
        train_loop = _train_loop(args.workload, args.train_loop, dev, False)
        train_loop["fused_activations"] = _train_loop(args.workload, args.train_loop, dev, True)

    cpu_baseline = None
    if world == 1 and not args.no_cpu_baseline:
        cpu_baseline = _cpu_baseline(scene, cam, cfg)

    line = {
        "metric": "train-step images/sec (fwd+bwd) @1080p, 1e6 Gaussians" if args.workload == "cfg3"
        else f"train-step images/sec (fwd+bwd) {args.workload}",
        "value": round(args.steps / elapsed, 3), "unit": "images/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / args.steps, 4), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": f"{args.workload}: {P} Gaussians, {cfg['W']}x{cfg['H']}, SH degree {cfg['D']} "
                               f"(SURVEY Appendix B seed {cfg['seed']}); step = render() + L1/D-SSIM loss + backward "
                               f"(train.py:79-108 window)",
                   "visible": V, "num_rendered": R, "instances_emitted": Re, "chunks_run": stats["chunks_run"],
                   "parallelism": "single" if sharded is None else f"tile-row slabs x{world}",
                   **({"collectives": collectives} if collectives else {})},
        "raster_ms_per_step": round(raster_ms, 4), "profiled_ms_per_step": round(1e3 * elapsed_profiled / args.steps, 4),
        "kernels": {k: {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items()}
                    for k, v in per_kernel.items()},
        "roofline": roofline,
        "cpu_baseline": cpu_baseline,
    }
    if fused is not None:
        line["fused_activations"] = fused
    if getter_fusion is not None:
        line["getter_fusion"] = getter_fusion
    if train_loop is not None:
        line["train_loop"] = train_loop
    print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


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
    return dict(num_rendered=fr.R, emitted=emitted, chunks_run=int(fr.plan.chunks_run), live=live,
                pairs_bwd=int(ncontrib.sum()))


def _train_loop(workload, iters, dev, fused):
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
    gm.adopt_scene(S.make_scene(cfg["P"], cfg["W"], cfg["H"], cfg["D"], cfg["seed"]), device=dev)
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
    pipe.fused_activations = bool(fused)
    train(gm, cams, targets, opt, pipe, bg, iterations=warm, scene_extent=6.0)
    with torch.no_grad():       # warm-up includes one full-size densification: the allocator has seen the grown tensors
        gm.densify_and_prune(opt.densify_grad_threshold, 0.005, 6.0, None)
    torch.cuda.synchronize(dev)
    n0 = gm._xyz.shape[0]
    t0 = time.perf_counter()
    train(gm, cams, targets, opt, pipe, bg, iterations=warm + iters, first_iter=warm + 1, scene_extent=6.0)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    return {"iterations": iters, "its_per_s": round(iters / dt, 2), "ms_per_it": round(1e3 * dt / iters, 3),
            "gaussians_start": int(n0), "gaussians_end": int(gm._xyz.shape[0]),
            "includes": "LR schedule, render, L1/D-SSIM, backward, densification stats, densify+prune every 100 it, Adam"}


def _traffic_from_profiles(kernel):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/traffic.json: FETCH_SIZE x2 +
    WRITE_SIZE, made by profiles/pmc_to_traffic.py for the default cfg3 single-GPU run), or None."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(path):
        try:
            return json.load(open(path)).get(kernel)
        except Exception:
            return None
    return None


def _cpu_baseline(scene, cam, cfg):
    """The CPU oracle (oracle/, a plain-C port of the algorithm; the reference has no CPU rasterizer:
    SURVEY F4) timed on the host cores: ONE frame of the same workload, rasterizer forward+backward."""
    import math

    import numpy as np

    import oracle
    oracle.build()
    a = scene.to("cpu").activated()
    cam = cam.to("cpu")
    kw = dict(image_height=cfg["H"], image_width=cfg["W"], tanfovx=math.tan(cam.FoVx * .5), tanfovy=math.tan(cam.FoVy * .5),
              bg=np.zeros(3), scale_modifier=1.0, viewmatrix=cam.world_view_transform.numpy(),
              projmatrix=cam.full_proj_transform.numpy(), sh_degree=cfg["D"], campos=cam.camera_center.numpy(),
              means3D=a["means3D"].numpy(), opacities=a["opacities"].numpy(), shs=a["shs"].numpy(),
              scales=a["scales"].numpy(), rotations=a["rotations"].numpy())
    import scene_synth as S
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
