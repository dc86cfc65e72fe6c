"""dg_head_post_bwd / dg_head_post_fwd_sum alone at the bench shape (64x1024, batch 32), with and without the bias sums:
HIP-event time per launch over a rotation of 4 operand sets (304 MB > the 256 MB infinity cache).
usage: python scripts/bench_head_post.py [arch 0|1|2]"""
import sys
import torch
sys.path.insert(0, ".")
from dusty_gan_amd import _lib as L

lib = L.lib()
arch = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B, H, W = 32, 64, 1024
HW = H * W
dev = "cuda"
cp = 2 if arch <= 1 else 4
sets = []
for i in range(4):
    g = torch.randn(B, 1 + arch, H, W, device=dev)
    sets.append(dict(g=g, npx=torch.randn(B, 1, H, W, device=dev), nim=torch.randn(B, device=dev),
                     mask=(torch.rand(B, max(arch, 1), H, W, device=dev) > 0.5).float(), go=torch.randn(B, 1, H, W, device=dev),
                     pm=torch.empty(B, H, W, cp, device=dev, dtype=torch.bfloat16), draw=torch.empty_like(g),
                     depth=torch.empty(B, 1, H, W, device=dev)))
dbias = torch.zeros(3, device=dev)
dsum = torch.zeros(B, device=dev)


ws = torch.zeros(B * 1024, device=dev)


def bwd(s, bias, planar, staged=False):
    L.check(lib.dg_head_post_bwd(s["g"].data_ptr(), s["npx"].data_ptr(), s["nim"].data_ptr(), s["mask"].data_ptr(),
                                 s["go"].data_ptr(), arch, 1.0, -1.0, B, HW, 0.25, 0.125,
                                 s["draw"].data_ptr() if planar else None, dbias.data_ptr() if bias else None,
                                 s["pm"].data_ptr(), cp, ws.data_ptr() if staged else None, L.stream_ptr()))


def fwd(s):
    L.check(lib.dg_head_post_fwd_sum(s["g"].data_ptr(), s["npx"].data_ptr(), s["nim"].data_ptr(), arch, 1, 1.0, -1.0, B, HW,
                                     s["mask"].data_ptr(), s["depth"].data_ptr(), dsum.data_ptr(), L.stream_ptr()))


def timeit(f, n=40):
    for i in range(8):
        f(sets[i % 4])
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n):
        f(sets[i % 4])
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


print(f"arch {arch}: bwd pm+staged bias {timeit(lambda s: bwd(s, True, False, True)):.1f} us, pm+bias {timeit(lambda s: bwd(s, True, False)):.1f} us, pm no bias {timeit(lambda s: bwd(s, False, False)):.1f} us, "
      f"planar+pm+bias {timeit(lambda s: bwd(s, True, True)):.1f} us, fwd_sum {timeit(fwd):.1f} us")
