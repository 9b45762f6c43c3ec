"""Times the library's radix sort through its test hook (gsr_debug_sort_pairs): the depth sort's shape (1e6 32-bit keys)
and the tile sort's (2.2e6 13-bit keys), device-side count as in the pipeline."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "structured-gaussian-splatting_amd")]
import torch
from diff_gaussian_rasterization import _native as N
dev = "cuda:0"
def bench(n, end_bit, dev_count, reps=50):
    g = torch.Generator().manual_seed(n)
    hi = (1 << end_bit) - 1 if end_bit < 31 else (1 << 31) - 1
    keys = torch.randint(0, hi + 1, (n,), generator=g, dtype=torch.int64).to(torch.int32).to(dev)
    vals = torch.arange(n, dtype=torch.int32, device=dev)
    for _ in range(5):
        ks, vs = N.debug_sort_pairs(keys, vals, end_bit, dev_count)
    torch.cuda.synchronize()
    N.profile_enable(True)
    for _ in range(reps):
        N.debug_sort_pairs(keys, vals, end_bit, dev_count)
    torch.cuda.synchronize()
    p = N.profile_read(); N.profile_enable(False)
    ms, cnt = p["debug_sort"]
    want, order = torch.sort(keys.long(), stable=True)
    ok = bool(torch.equal(ks.long(), want) and torch.equal(vs.long(), order))
    print(f"n={n} bits={end_bit} dev_count={dev_count}: {1e3*ms/cnt:.1f} us per sort ({n/1e3/(ms/cnt):.1f} Mkeys/ms) correct={ok}", flush=True)
for n, b, d in ((1_000_000, 32, False), (2_170_000, 13, True), (1_630_000, 13, True), (5_000_000, 32, False), (8_000_000, 15, True)):
    bench(n, b, d)
