"""The step's image-sized pointwise launches alone at the bench shape (64x1024, B = 32 / 2B = 64), HIP-event time per launch over
a rotation of operand sets, each in the forms that tell WHAT it is bound by: with its cross-block sums in the registered
accumulator arena (fixed-point adds + ticket: common.h dg_acc_add), with the sums as plain float atomics (no arena), and
without sums where an entry point allows it.   usage: python scripts/bench_pointwise.py"""
import math
import sys

import torch

sys.path.insert(0, ".")
from dusty_gan_amd import _lib as L  # noqa: E402

lib = L.lib()
B, H, W = 32, 64, 1024
HW = H * W
dev = "cuda"
NSET = 6
torch.manual_seed(0)


def timeit(f, n=60):
    for i in range(2 * NSET):
        f(i % NSET)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n):
        f(i % NSET)
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


L.AccArena.begin(torch.device(dev))
arena = [L.AccArena.take(64, torch.device(dev)) for _ in range(NSET)]
plain = [torch.zeros(64, device=dev) for _ in range(NSET)]
macc_a = L.AccArena.take(16, torch.device(dev))
macc_p = torch.zeros(16, device=dev)

# ---- head post-processing forward (arch none): gout [B,1,H,W] -> depth (+ per-sample sums)
g = [torch.randn(B, 1, H, W, device=dev) for _ in range(NSET)]
depth = [torch.empty(B, 1, H, W, device=dev) for _ in range(NSET)]
mask = torch.empty(B, 1, H, W, device=dev)


def hp(i, sums):
    if sums is None:
        L.check(lib.dg_head_post_fwd(g[i].data_ptr(), None, None, 0, 1, 1.0, -1.0, B, HW, mask.data_ptr(), depth[i].data_ptr(), None))
    else:
        L.check(lib.dg_head_post_fwd_sum(g[i].data_ptr(), None, None, 0, 1, 1.0, -1.0, B, HW, mask.data_ptr(), depth[i].data_ptr(),
                                         sums[i].data_ptr(), None))


print(f"head_post_fwd (17 MB):   sums in arena {timeit(lambda i: hp(i, arena)):6.1f} us   float atomics {timeit(lambda i: hp(i, plain)):6.1f} us   "
      f"no sums (scalar kernel) {timeit(lambda i: hp(i, None)):6.1f} us")

# ---- final conv forward: h4 [2B, 131072] bf16 -> logits
nf = 4 * 64 * 512
h4 = [torch.randn(2 * B, nf, device=dev).bfloat16() for _ in range(NSET)]
wf = torch.randn(nf, device=dev)
fb = torch.zeros(1, device=dev)
for nb in (2 * B, B):
    def ff(i, ys):
        L.check(lib.dg_final_fwd_acc(h4[i].data_ptr(), L.DG_BF16, wf.data_ptr(), fb.data_ptr(), 1.0 / math.sqrt(nf), nb, nf,
                                     ys[i].data_ptr(), None))
    print(f"final_fwd {nb} samples ({nb * nf * 2 / 1e6:.0f} MB): logits in arena {timeit(lambda i: ff(i, arena)):6.1f} us   float atomics {timeit(lambda i: ff(i, plain)):6.1f} us")

# ---- final conv loss step + backward
e4 = [torch.empty(2 * B, nf, device=dev, dtype=torch.bfloat16) for _ in range(NSET)]
y = torch.randn(2 * B, device=dev)
dy, up, rs, acc = (torch.zeros(2 * B, device=dev) for _ in range(4))
dwf, dbp, db4, dfb = torch.zeros(nf, device=dev), torch.zeros(nf, device=dev), torch.zeros(512, device=dev), torch.zeros(1, device=dev)
for mode_g, nb in ((0, B), (1, B)):
    def fg(i, part=True):
        L.check(lib.dg_final_gan_bwd(0, mode_g, 1.0, y.data_ptr(), y.data_ptr() + 4 * B, nb, 1.0, 0 if mode_g else 1, dy.data_ptr(),
                                     None if mode_g else up.data_ptr(), None if mode_g else rs.data_ptr(), acc.data_ptr(),
                                     None if mode_g else dfb.data_ptr(), h4[i].data_ptr(), L.DG_BF16, wf.data_ptr(), 1.0 / math.sqrt(nf),
                                     nf, 512, e4[i].data_ptr(), None if mode_g else db4.data_ptr(), None if mode_g else dwf.data_ptr(),
                                     None if (mode_g or not part) else dbp.data_ptr(), None))
    print(f"final_gan_bwd {'G' if mode_g else 'D'} ({(1 if mode_g else 2) * nb * nf * 4 / 1e6:.0f} MB): {timeit(fg):6.1f} us"
          + ("" if mode_g else f"   bias gradient by atomics {timeit(lambda i: fg(i, False)):6.1f} us"))

# ---- BlurVH adjoint forms: e0 [B,H,W,2] bf16 -> image
e0 = [torch.randn(B, H, W, 2, device=dev).bfloat16() for _ in range(NSET)]
dx = [torch.empty(B, 1, H, W, device=dev) for _ in range(NSET)]
h0 = [torch.empty(B, H, W, 2, device=dev, dtype=torch.bfloat16) for _ in range(NSET)]


def bb(i, ssq):
    if ssq is None:
        L.check(lib.dg_blur_bwd(e0[i].data_ptr(), L.DG_BF16, dx[i].data_ptr(), B, H, W, 1, None))
    else:
        L.check(lib.dg_blur_bwd_r1(e0[i].data_ptr(), L.DG_BF16, dx[i].data_ptr(), 0.03, ssq[i].data_ptr(), B, H, W, 1, None))


def bf(i):
    L.check(lib.dg_blur_fwd(dx[i].data_ptr(), h0[i].data_ptr(), L.DG_BF16, B, H, W, 1, None))


def rt(i, ssq, macc):
    L.check(lib.dg_blur_r1_tangent(e0[i].data_ptr(), L.DG_BF16, h0[i].data_ptr(), 0.03, ssq[i].data_ptr(),
                                   None if macc is None else macc.data_ptr(), B, B, H, W, 1, None))


print(f"blur_bwd (17 MB): r1 sums in arena {timeit(lambda i: bb(i, arena)):6.1f} us   float atomics {timeit(lambda i: bb(i, plain)):6.1f} us   "
      f"no sums {timeit(lambda i: bb(i, None)):6.1f} us;   blur_fwd (17 MB) {timeit(bf):6.1f} us")
print(f"blur_r1_tangent (17 MB): arena sums + mean {timeit(lambda i: rt(i, arena, macc_a)):6.1f} us   arena sums, no mean "
      f"{timeit(lambda i: rt(i, arena, None)):6.1f} us   float atomics, no mean {timeit(lambda i: rt(i, plain, None)):6.1f} us   "
      f"float atomics + mean {timeit(lambda i: rt(i, plain, macc_p)):6.1f} us")

# ---- DiffAugment + BlurVH over real | fake
from dusty_gan_amd.utils.diff_augment import DiffAugment  # noqa: E402
A = DiffAugment()
xs = [torch.randn(2, B, 1, H, W, device=dev) for _ in range(NSET)]
sums = torch.randn(2, B, device=dev)
rps = [A.draw(B, H, W, torch.device(dev)) for _ in range(2)]
h0b = [torch.empty(2 * B, H, W, 2, device=dev, dtype=torch.bfloat16) for _ in range(NSET)]
keep = []


def da(i, nsets):
    arr = (L.DgAugSet * nsets)()
    for k in range(nsets):
        args, kp = A._args(rps[k], B, torch.device(dev))
        keep.append(kp)
        arr[k].x, arr[k].xsum, arr[k].xsum_parts = xs[i][k].data_ptr(), sums[k].data_ptr(), 1
        arr[k].u_b, arr[k].u_c, arr[k].t_h, arr[k].t_w, arr[k].o_x, arr[k].o_y = args
    L.check(lib.dg_diffaug_blur_fwd(arr, nsets, A.mask, B, H, W, 1, h0b[i].data_ptr(), L.DG_BF16, None))


print(f"diffaug_blur_fwd: 2 sets (34 MB) {timeit(lambda i: da(i, 2)):6.1f} us   1 set (17 MB) {timeit(lambda i: da(i, 1)):6.1f} us")

# ---- the step's first launch: zero-fill 23 MB + fetch_reals of a pooled batch
pool_p = torch.rand(8, B, 1, H, W, device=dev)
pool_m = (torch.rand(8, B, 1, H, W, device=dev) < 0.85).float()
ctr = torch.zeros(1, dtype=torch.int64, device=dev)
zbuf = [torch.empty(5767168, device=dev) for _ in range(NSET)]
outs = [torch.empty(B, 1, H, W, device=dev) for _ in range(NSET)]
parts = torch.empty(B, L.XSUM_PARTS, device=dev)


def pro(i, fetch, zero):
    f = None
    if fetch:
        f = L.DgFetch()
        f.pol, f.mask, f.pool_ctr, f.npool = pool_p.data_ptr(), pool_m.data_ptr(), ctr.data_ptr(), 8
        f.min_depth, f.max_depth, f.drop_const, f.B, f.HW = 0.9, 120.0, -1.0, B, HW
        f.out, f.parts = outs[i].data_ptr(), parts.data_ptr()
    L.step_prologue([zbuf[i]] if zero else [], [], fetch=f)


print(f"step prologue: zero 23 MB + fetch (25 MB) {timeit(lambda i: pro(i, True, True)):6.1f} us   fetch alone {timeit(lambda i: pro(i, True, False)):6.1f} us   "
      f"zero alone {timeit(lambda i: pro(i, False, True)):6.1f} us")
