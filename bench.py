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


def algorithmic_bytes(P, V, R, N, Tn, K, M):
    """Per-kernel ALGORITHMIC bytes of one step (each logical array counted once per
    read and once per write).  Keys are the library's profile names."""
    return {
        "preprocess": 44 * P + 12 * K * V + 8 * P + 67 * V,
        "scan": 8 * P,
        "duplicate": 4 * P + 16 * V + 12 * R,
        "radix_sort": 24 * R,                       # algorithmic minimum: one read + one write of (key, value)
        "ranges": 8 * R + 8 * Tn,
        "render_fwd": 40 * R + 12 + 20 * N,
        "render_bwd": 36 * P + 40 * R + 20 * N + 36 * R,
        "reduce_rows": 36 * R + 36 * P,            # this design's deterministic reduction (replaces atomics RMW)
        "geom_bwd": 4 * P + (99 + 12 * K) * V + (40 + 12 * M) * P,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="cfg3", choices=["cfg2", "cfg3", "cfg5"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with --nproc-per-node {args.gpus} "
                         f"(WORLD_SIZE={world})")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

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

    if world > 1:
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
        loss = training_loss(out["render"], gt)
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
    N.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    sync()
    elapsed = time.perf_counter() - t0
    prof = N.profile_read()
    N.profile_enable(False)
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- workload statistics for the roofline (measured, not assumed)
    radii = out["radii"]
    P = scene.P
    V = int((radii > 0).sum())
    Npix, Tn = cfg["W"] * cfg["H"], ((cfg["W"] + 15) // 16) * ((cfg["H"] + 15) // 16)
    K = M = (cfg["D"] + 1) ** 2
    R = int(getattr(render, "last_num_rendered", 0)) or _num_rendered(model, cam, bg, pipe)
    alg = algorithmic_bytes(P, V, R, Npix, Tn, K, M)
    per_kernel = {k: dict(ms=ms / max(n, 1), launches_per_step=n / args.steps,
                          alg_GBs=(alg.get(k, 0) / 1e9) / (ms / max(n, 1) / 1e3) if ms > 0 else None)
                  for k, (ms, n) in prof.items()}
    raster_ms = sum(ms for ms, _ in prof.values()) / args.steps
    dom = max(prof, key=lambda k: prof[k][0])
    dom_ms = prof[dom][0] / max(prof[dom][1], 1)
    achieved = (alg[dom] / 1e9) / (dom_ms / 1e3)
    roofline = dict(bound="hbm", kernel=dom, achieved=round(achieved, 2), peak=HBM_PEAK_GBS, unit="GB/s",
                    frac=round(achieved / HBM_PEAK_GBS, 5), traffic=_traffic_from_profiles(dom),
                    avg_launch_ms=round(dom_ms, 4), algorithmic_bytes_per_launch=int(alg[dom]),
                    step_algorithmic_bytes=int(sum(alg.values())),
                    step_frac=round(sum(alg.values()) / 1e9 / (elapsed / args.steps) / HBM_PEAK_GBS, 5))

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

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
                   "visible": V, "num_rendered": R, "parallelism": "single" if world == 1 else f"tile-row slabs x{world}"},
        "raster_ms_per_step": round(raster_ms, 4),
        "kernels": {k: {kk: (round(vv, 4) if isinstance(vv, float) else vv) for kk, vv in v.items()}
                    for k, v in per_kernel.items()},
        "roofline": roofline,
        "cpu_baseline": cpu_baseline,
    }
    print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


def _num_rendered(model, cam, bg, pipe):
    """R of the benchmark frame (one extra un-timed forward through the functional API)."""
    import math

    import diff_gaussian_rasterization as dgr
    with torch.no_grad():
        rs = dgr.GaussianRasterizationSettings(cam.image_height, cam.image_width, math.tan(cam.FoVx * .5),
                                               math.tan(cam.FoVy * .5), bg, 1.0, cam.world_view_transform,
                                               cam.full_proj_transform, model.active_sh_degree, cam.camera_center,
                                               False, False)
        _, _, fr = dgr.rasterize_forward(model.get_xyz, model.get_features, None, model.get_opacity, model.get_scaling,
                                         model.get_rotation, None, rs)
    return fr.R


def _traffic_from_profiles(kernel):
    """HBM bytes per launch from the committed rocprofv3 PMC passes (profiles/traffic.json), or None."""
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
    return {"value": round(1.0 / dt, 4), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"1 frame of the same workload, rasterizer forward+backward only (no loss), C oracle with "
                      f"OpenMP over {cores} host threads (sort and per-Gaussian stages serial); {dt:.1f} s"}


if __name__ == "__main__":
    main()
