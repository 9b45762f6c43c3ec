"""Host-side timeline of one training step (cfg3, GaussianModel): wall-clock stamps at the library's entry points, averaged
over steady-state steps, no profiler attached.  Shows how long the host needs between the forward's last readback and the
launch of the blend backward (the stretch in which the stream can run dry)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "structured-gaussian-splatting_amd")]
import torch
import scene_synth as S
from gaussian_params import Pipe
import gaussian_renderer
from scene import GaussianModel
import loss_utils
import diff_gaussian_rasterization as dgr
from diff_gaussian_rasterization import _native as N

dev = "cuda:0"
wl = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
cfg = S.CONFIGS[wl]
scene, cam = S.make_config(wl); scene, cam = scene.to(dev), cam.to(dev)
gm = GaussianModel(scene.sh_degree); gm.adopt_scene(scene, device=dev)
bg = torch.zeros(3, device=dev); gt = torch.rand(3, cfg["H"], cfg["W"], device=dev); pipe = Pipe()
ps = list(gm._t.values())
T = []
now = time.perf_counter_ns


def wrap(mod, name, tag):
    f = getattr(mod, name)
    def g(*a, **k):
        T.append((tag + ">", now()))
        r = f(*a, **k)
        T.append((tag + "<", now()))
        return r
    setattr(mod, name, g)


wrap(N, "forward_both", "native forward (both stages, 2 readbacks)")
wrap(N, "loss_forward", "native loss fwd")
wrap(N, "loss_backward", "native loss bwd")
wrap(N, "backward_render", "native backward_render")
wrap(N, "backward_geom", "native backward_geom")
wrap(N, "activations_forward", "native act fwd")
wrap(N, "activations_backward", "native act bwd")
if os.environ.get("FINE"):
    wrap(dgr, "_alloc_grads", "  _alloc_grads")
    wrap(dgr, "_workspace", "  _workspace")
    wrap(dgr, "rasterize_forward", " rasterize_forward")
    wrap(dgr, "prepare_backward", " prepare_backward")
    wrap(dgr, "rasterize_backward_screen", " rasterize_backward_screen")
    wrap(dgr, "rasterize_backward_geom", " rasterize_backward_geom")
    wrap(N, "Grads", "   N.Grads()")


if os.environ.get("AG"):
    def _alloc_grads_instr(fr, needs, alloc):
        P, M, dev_ = fr.desc.P, fr.M, fr.device
        T.append(("   ag: enter", now()))
        (_, means3D, sh, colors_precomp, opacities, scales, rotations, cov3D_precomp, sh_rest) = fr.keep
        T.append(("   ag: unpacked keep", now()))
        def mk(flag, present, *shape):
            return alloc(*shape, dtype=torch.float32, device=dev_) if (flag and present) else None
        t0_ = mk(needs[0], True, P, 3); T.append(("   ag: t0", now()))
        t1_ = mk(needs[1], True, P, 3); T.append(("   ag: t1", now()))
        t2_ = mk(needs[2], sh is not None, P, 1 if fr.raw else M, 3); T.append(("   ag: t2 (sh)", now()))
        t3_ = mk(needs[3], colors_precomp is not None, P, 3)
        t4_ = mk(needs[4], True, P, 1); T.append(("   ag: t4", now()))
        t5_ = mk(needs[5], scales is not None, P, 3)
        t6_ = mk(needs[6], rotations is not None, P, 4); T.append(("   ag: t6", now()))
        t7_ = mk(needs[7], cov3D_precomp is not None, P, 6)
        t8_ = mk(fr.raw and len(needs) > 8 and needs[8], sh_rest is not None, P, M - 1, 3); T.append(("   ag: t8", now()))
        t = (t0_, t1_, t2_, t3_, t4_, t5_, t6_, t7_, t8_)
        ptrs = []
        for i, x in enumerate(t):
            ptrs.append(N._ptr(x)); T.append(("   ag: ptr%d" % i, now()))
        grads = N.Grads(*ptrs, 0); T.append(("   ag: Grads", now()))
        return t, grads
    dgr._alloc_grads = _alloc_grads_instr


def step():
    T.append(("step>", now()))
    for p in ps: p.grad = None
    out = gaussian_renderer.render(cam, gm, pipe, bg)
    T.append(("render() returned", now()))
    loss = loss_utils.training_loss(out["render"], gt)
    T.append(("training_loss returned", now()))
    loss.backward()
    T.append(("backward() returned", now()))


for _ in range(30): step()
torch.cuda.synchronize()
T.clear()
def _stat():
    s = torch.cuda.memory_stats(dev)
    return {k: s.get(k, 0) for k in ("num_device_alloc", "num_device_free", "num_alloc_retries", "num_sync_all_streams", "allocation.all.allocated")}
_s0 = _stat()
t0 = time.perf_counter()
for _ in range(200): step()
torch.cuda.synchronize()
print("ms/step %.4f (with stamps)" % (1e3 * (time.perf_counter() - t0) / 200))
_s1 = _stat(); print("allocator per step:", {k: (_s1[k] - _s0[k]) / 200 for k in _s0})
# average offset of each tag from its step's start
import collections
acc, cnt = collections.OrderedDict(), collections.Counter()
start = None
for tag, t in T:
    if tag == "step>":
        start = t; seen = collections.Counter()
    seen[tag] += 1
    key = tag if seen[tag] == 1 else "%s #%d" % (tag, seen[tag])
    acc[key] = acc.get(key, 0) + (t - start); cnt[key] += 1
prev = 0.0
for k, v in acc.items():
    off = v / cnt[k] / 1e3
    print("%9.1f us  (+%6.1f)  %s" % (off, off - prev, k))
    prev = off

