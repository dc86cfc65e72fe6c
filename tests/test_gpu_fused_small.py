"""GPU parity of the round-2 forms of the small kernels (through the C ABI): each fused / pre-zeroed / vector form against
the plain entry point it replaces in the training step, and the scalar fall-backs of the 16-byte kernels against the
oracle.  Reference call sites: trainers/dcgan_amp.py:154-160 (fetch_reals), :218-235 (R1), utils/diff_augment.py:29-33
(contrast mean), models/ops/common.py:74-88 (BlurVH), models/gans/dcgan_eqlr.py:95 (final conv)."""
import ctypes as C
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import dusty_oracle as O
from tests.golden_util import rel_l2
from tests.test_gpu_ops import from_nhwc, nhwc

pytestmark = pytest.mark.gpu
DEV = "cuda"


@pytest.fixture(scope="module")
def L():
    from dusty_gan_amd import _lib
    _lib.lib()
    return _lib


@pytest.mark.parametrize("H,W", [(32, 64), (64, 1024), (8, 48)])   # 8 x 48: H W % 1024 != 0 -> refused
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_blur_adjoint_r1_form(L, H, W, dtype):
    """dg_blur_bwd_r1 == dg_blur_bwd, then |g_b|^2 per sample and the scaled copy (the three launches it replaces)"""
    lib = L.lib()
    g = torch.Generator().manual_seed(H + W)
    B = 5
    d = torch.randn(B, H, W, 2, generator=g).to(DEV, dtype)
    ref = torch.empty(B, 1, H, W, device=DEV)
    L.check(lib.dg_blur_bwd(d.data_ptr(), L.dtype_code(dtype), ref.data_ptr(), B, H, W, 1, None))
    out = torch.full((B, 1, H, W), 7.0, device=DEV)
    ssq = torch.zeros(B, device=DEV)
    rc = lib.dg_blur_bwd_r1(d.data_ptr(), L.dtype_code(dtype), out.data_ptr(), 0.375, ssq.data_ptr(), B, H, W, 1, None)
    torch.cuda.synchronize()
    if (H * W) % 1024 != 0:
        assert rc == L.DG_EUNSUPPORTED and float(out.min()) == 7.0   # refused before anything was written
        return
    L.check(rc)
    assert torch.equal(out, 0.375 * ref)
    assert rel_l2(ssq.cpu(), ref.pow(2).sum(dim=[1, 2, 3]).cpu()) < 1e-6


@pytest.mark.parametrize("H,W", [(8, 32), (64, 1024), (5, 7)])   # 5 x 7: H W % 256 != 0 -> refused
def test_fetch_reals_and_head_emit_per_sample_sums(L, H, W):
    lib = L.lib()
    g = torch.Generator().manual_seed(7 * H + W)
    B, HW = 3, H * W
    pol = torch.rand(B, 1, H, W, generator=g).to(DEV)
    m = (torch.rand(B, 1, H, W, generator=g) < 0.8).float().to(DEV)
    ref = torch.empty_like(pol)
    L.check(lib.dg_fetch_reals(pol.data_ptr(), m.data_ptr(), 0.9, 120.0, -1.0, pol.numel(), ref.data_ptr(), None))
    out, sums = torch.empty_like(pol), torch.zeros(B, device=DEV)
    rc = lib.dg_fetch_reals_sum(pol.data_ptr(), m.data_ptr(), 0.9, 120.0, -1.0, B, HW, out.data_ptr(), sums.data_ptr(), None)
    if HW % 256 != 0:
        assert rc == L.DG_EUNSUPPORTED
    else:
        L.check(rc)
        assert torch.equal(out, ref)
        assert rel_l2(sums.cpu(), ref.sum(dim=[1, 2, 3]).cpu()) < 1e-6
    for k in (0, 1, 2):  # none / dusty1 / dusty2 heads
        raw = torch.randn(B, 1 + k, H, W, generator=g).to(DEV)
        npx = torch.randn(B, H, W, generator=g).to(DEV)
        nim = torch.randn(B, generator=g).to(DEV)
        g1, g2 = raw.clone(), raw.clone()
        m1, m2 = torch.zeros(B, max(k, 1), H, W, device=DEV), torch.zeros(B, max(k, 1), H, W, device=DEV)
        d1, d2 = torch.empty(B, 1, H, W, device=DEV), torch.empty(B, 1, H, W, device=DEV)
        L.check(lib.dg_head_post_fwd(g1.data_ptr(), npx.data_ptr(), nim.data_ptr(), k, 1, 1.0, -1.0, B, HW, m1.data_ptr(),
                                     d1.data_ptr(), None))
        ds = torch.zeros(B, device=DEV)
        rc = lib.dg_head_post_fwd_sum(g2.data_ptr(), npx.data_ptr(), nim.data_ptr(), k, 1, 1.0, -1.0, B, HW, m2.data_ptr(),
                                      d2.data_ptr(), ds.data_ptr(), None)
        if HW % 256 != 0:
            assert rc == L.DG_EUNSUPPORTED
            continue
        L.check(rc)
        assert torch.equal(d1, d2) and torch.equal(g1, g2) and torch.equal(m1, m2)
        assert rel_l2(ds.cpu(), d1.sum(dim=[1, 2, 3]).cpu()) < 1e-6


def test_diffaug_takes_the_producers_sums_and_the_arena(L):
    """DiffAugment.apply with (a) its own pass, (b) a pre-zeroed arena slice (dg_diffaug_fwd_acc), (c) the sums its input's
    producer tagged on the tensor (dg_diffaug_fwd_pre): the same image; a stale tag (older arena epoch) is ignored"""
    from dusty_gan_amd.utils.diff_augment import DiffAugment
    torch.manual_seed(5)
    B, H, W = 4, 32, 64
    A = DiffAugment(["brightness", "saturation", "contrast", "translation", "cutout"])
    x = torch.rand(B, 1, H, W, device=DEV) * 2 - 1
    rp = A.draw(B, H, W, DEV)
    L.AccArena.buf = None                                   # (a) no arena
    ya = A.apply(x, rp)
    L.AccArena.begin(DEV)                                   # (b) arena slice
    pos = L.AccArena.pos
    yb = A.apply(x, rp)
    assert L.AccArena.pos > pos and torch.equal(ya, yb)
    sums = L.AccArena.take(B, DEV)                          # (c) tagged sums
    sums.copy_(x.sum(dim=[1, 2, 3]))
    L.tag_sums(x, sums)
    pos = L.AccArena.pos
    yc = A.apply(x, rp)
    assert L.AccArena.pos == pos                            # no slice of its own
    assert rel_l2(yc.cpu(), ya.cpu()) < 1e-6
    sums.fill_(1e6)                                         # a wrong tag would show
    L.AccArena.begin(DEV)                                   # new epoch: the tag is stale
    assert L.tagged_sums(x) is None
    assert torch.equal(A.apply(x, rp), ya)
    x2 = x.clone()
    L.tag_sums(x2, L.AccArena.take(B, DEV))
    x2.add_(1.0)                                            # rewritten by a torch op: tag void
    assert L.tagged_sums(x2) is None


def test_arena_slices_are_zero_once_and_exhaustion_falls_back(L):
    L.AccArena.begin(DEV)
    a = L.AccArena.take(40, DEV)
    b = L.AccArena.take(40, DEV)
    assert a.data_ptr() % 64 == 0 and b.data_ptr() - a.data_ptr() == 48 * 4       # 64-byte slices, handed out once
    assert float(a.abs().max()) == 0.0 and float(b.abs().max()) == 0.0
    a.fill_(3.0)
    L.AccArena.begin(DEV)
    assert float(L.AccArena.take(40, DEV).abs().max()) == 0.0                       # re-zeroed by the next step
    assert L.AccArena.take(L.AccArena.SIZE, DEV) is None                            # exhausted -> caller's own buffer
    # the self-zeroing and the _acc forms agree
    lib = L.lib()
    x = torch.randn(6, 4096, device=DEV)
    o1, o2 = torch.full((6,), 9.0, device=DEV), torch.zeros(6, device=DEV)
    L.check(lib.dg_sample_sum(x.data_ptr(), 6, 4096, 1, o1.data_ptr(), None))
    L.check(lib.dg_sample_sum_acc(x.data_ptr(), 6, 4096, 1, o2.data_ptr(), None))
    assert rel_l2(o1.cpu(), o2.cpu()) < 1e-6 and rel_l2(o1.cpu(), x.pow(2).sum(1).cpu()) < 1e-6


def test_counter_add_multi_snap_into_pinned_host_ring(L):
    """dg_counter_add_multi_snap: the advancing launch files src[0..n) in slot (OLD counter) % ring - of a device buffer and
    of mapped pinned host memory (what Trainer.step's scalars use); argument errors."""
    lib = L.lib()
    ring_n, n = 4, 8
    for host in (False, True):
        c = [torch.full((1,), v, dtype=torch.int64, device=DEV) for v in (100, 6)]
        ring = torch.zeros(ring_n, n).pin_memory() if host else torch.zeros(ring_n, n, device=DEV)
        ptrs = (C.c_void_p * 2)(*[t.data_ptr() for t in c])
        for k in range(6):
            src = torch.arange(n, dtype=torch.float32, device=DEV) + 10.0 * k
            L.check(lib.dg_counter_add_multi_snap(ptrs, (C.c_uint64 * 2)(5, 1), 2, 1, src.data_ptr(), n, ring.data_ptr(),
                                                  ring_n, None))
            torch.cuda.synchronize()
            assert [int(t) for t in c] == [100 + 5 * (k + 1), 6 + k + 1]
            assert ring[(6 + k) % ring_n].tolist() == [float(i) + 10.0 * k for i in range(n)], (host, k)
        assert lib.dg_counter_add_multi_snap(ptrs, (C.c_uint64 * 2)(5, 1), 2, 2, src.data_ptr(), n, ring.data_ptr(), ring_n,
                                             None) == L.DG_EINVAL
        assert lib.dg_counter_add_multi_snap(ptrs, (C.c_uint64 * 2)(5, 1), 2, 0, None, n, ring.data_ptr(), ring_n,
                                             None) == L.DG_EINVAL
        assert lib.dg_counter_add_multi_snap(ptrs, (C.c_uint64 * 2)(5, 1), 2, 0, src.data_ptr(), 65, ring.data_ptr(), ring_n,
                                             None) == L.DG_EINVAL
    # the host-side queue: a snapshot needs its counter's advance in the same flush
    L.Counters.flush()
    ctr = torch.zeros(1, dtype=torch.int64, device=DEV)
    L.Counters.snapshot(ctr, src.data_ptr(), n, ring.data_ptr(), ring_n)
    with pytest.raises(RuntimeError):
        L.Counters.flush()
    assert L.Counters.snap is None


def test_counters_queue_and_multi_add(L):
    from dusty_gan_amd.utils.rng import Philox
    lib = L.lib()
    c = [torch.full((1,), 10 * i, dtype=torch.int64, device=DEV) for i in range(3)]
    ptrs = (C.c_void_p * 3)(*[t.data_ptr() for t in c])
    L.check(lib.dg_counter_add_multi(ptrs, (C.c_uint64 * 3)(1, 2, 3), 3, None))
    torch.cuda.synchronize()
    assert [int(t) for t in c] == [1, 12, 23]
    dup = (C.c_void_p * 2)(c[0].data_ptr(), c[0].data_ptr())
    assert lib.dg_counter_add_multi(dup, (C.c_uint64 * 2)(1, 1), 2, None) == L.DG_EINVAL
    assert lib.dg_counter_add_multi(ptrs, (C.c_uint64 * 3)(1, 2, 3), 9, None) == L.DG_EINVAL
    # queued advances: two draws from one generator see consecutive offsets, .offset is current, numbers == one long draw
    L.Counters.flush()
    r1, r2 = Philox(77, DEV, stream_id=3), Philox(77, DEV, stream_id=3)
    a = r1.normal(1000)
    assert r1.ctr.data_ptr() in L.Counters.pending           # the advance is waiting for a flush
    b = r1.normal(1000)                                      # ... which the second draw triggers
    assert r1.offset == 500 and not L.Counters.pending
    ab = r2.normal(2000)
    assert torch.equal(torch.cat([a, b]), ab)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_final_conv_kernels_scalar_fallbacks(L, dtype):
    """n = 3 * 5 * 7 (not a multiple of the 16-byte vector) and C = 7: the scalar kernels behind dg_final_fwd /
    dg_final_bwd_data / dg_batch_wsum, against the oracle ops (the vector forms: tests/test_gpu_ops.py)"""
    lib = L.lib()
    g = torch.Generator().manual_seed(9)
    B, C3, h0, w0 = 5, 7, 3, 5
    n = h0 * w0 * C3
    d4 = torch.randn(B, C3, h0, w0, generator=g)
    if dtype == torch.bfloat16:
        d4 = d4.bfloat16().float()
    d4r = d4.clone().requires_grad_()
    wf = torch.randn(1, C3, h0, w0, generator=g).requires_grad_()
    bf = torch.randn(1, generator=g)
    yf = F.conv2d(d4r * O.equal_lr_scale(wf), wf, bf).view(B)
    d4d = nhwc(d4).to(DEV, dtype)
    wfd = wf.detach().permute(0, 2, 3, 1).contiguous().view(-1).to(DEV)
    yd, bfd = torch.empty(B, device=DEV), bf.to(DEV)
    L.check(lib.dg_final_fwd(d4d.data_ptr(), L.dtype_code(dtype), wfd.data_ptr(), bfd.data_ptr(), 1.0 / math.sqrt(n), B, n,
                             yd.data_ptr(), None))
    assert rel_l2(yd.cpu(), yf) < 1e-5
    up = torch.randn(B, generator=g)
    upd = up.to(DEV)
    gd4, gwf = torch.autograd.grad(yf, [d4r, wf], up)
    ref = gd4 * torch.where(d4 > 0, 1.0, 0.2) * math.sqrt(2.0)
    dd4, db = torch.empty_like(d4d), torch.zeros(C3, device=DEV)
    L.check(lib.dg_final_bwd_data(d4d.data_ptr(), L.dtype_code(dtype), wfd.data_ptr(), upd.data_ptr(), None,
                                  1.0 / math.sqrt(n), B, n, C3, dd4.data_ptr(), db.data_ptr(), None))
    assert rel_l2(from_nhwc(dd4.float().cpu(), B, C3, h0, w0), ref) < (1e-6 if dtype == torch.float32 else 1e-2)
    assert rel_l2(db.cpu(), ref.sum(dim=[0, 2, 3])) < (1e-5 if dtype == torch.float32 else 2e-2)
    dwf = torch.zeros(n, device=DEV)
    L.check(lib.dg_batch_wsum(d4d.data_ptr(), L.dtype_code(dtype), upd.data_ptr(), 1.0 / math.sqrt(n), B, n,
                              dwf.data_ptr(), None))
    assert rel_l2(dwf.cpu().view(h0, w0, C3).permute(2, 0, 1), gwf[0]) < 1e-5


@pytest.mark.parametrize("ring", [True, False])
def test_blur_scalar_fallback_width_not_a_multiple_of_four(L, ring):
    lib = L.lib()
    g = torch.Generator().manual_seed(4)
    B, H, W = 2, 6, 10
    x = torch.randn(B, 1, H, W, generator=g).requires_grad_()
    y = O.blur_vh(x, ring)
    out = torch.empty(B * H * W * 2, device=DEV)
    xd = x.detach().to(DEV)
    L.check(lib.dg_blur_fwd(xd.data_ptr(), out.data_ptr(), L.DG_F32, B, H, W, int(ring), None))
    assert rel_l2(from_nhwc(out.cpu(), B, 2, H, W), y) < 1e-6
    gy = torch.randn(y.shape, generator=g)
    (gx,) = torch.autograd.grad(y, x, gy)
    dx = torch.empty(B, 1, H, W, device=DEV)
    gyd = nhwc(gy).to(DEV)
    L.check(lib.dg_blur_bwd(gyd.data_ptr(), L.DG_F32, dx.data_ptr(), B, H, W, int(ring), None))
    assert rel_l2(dx.cpu(), gx) < 1e-6


def _aug_params(B, H, W, seed, extreme=False):
    g = torch.Generator().manual_seed(seed)
    rp = O.draw_augment_params(B, H, W, g)
    if extreme:  # shifts / boxes at their limits: rows shifted out, wrap at W - 1, boxes clipped by the border
        sh, sw = O.translation_shift(H, W)
        rp["t_h"][0], rp["t_w"][0] = -sh, sw
        rp["t_h"][1], rp["t_w"][1] = sh, -sw
        rp["o_x"][0], rp["o_y"][0] = 0, 0
        rp["o_x"][1], rp["o_y"][1] = H - 1, W - 1
    return rp


@pytest.mark.parametrize("H,W,ring", [(32, 64, 1), (64, 1024, 1), (16, 40, 0)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("nsets", [1, 2])
def test_diffaug_blurvh_one_pass(L, H, W, ring, dtype, nsets):
    """dg_diffaug_blur_fwd (DiffAugment + BlurVH, the augmented image never written; one or two source sets per launch)
    == dg_diffaug_fwd_pre into a buffer, then dg_blur_fwd - the launches it replaces in D(A(x)); and the augmented
    image itself against the oracle's DiffAugment (utils/diff_augment.py:114-132)"""
    from dusty_gan_amd.utils.diff_augment import DiffAugment
    lib = L.lib()
    B = 3
    A = DiffAugment()
    g = torch.Generator().manual_seed(H * 3 + W + nsets)
    xs = [torch.randn(B, 1, H, W, generator=g) for _ in range(nsets)]
    rps = [_aug_params(B, H, W, 11 + k, extreme=True) for k in range(nsets)]
    ref = torch.empty(nsets * B, H, W, 2, device=DEV, dtype=dtype)
    sets, keep = [], []
    for k in range(nsets):
        xd = xs[k].to(DEV)
        rp = DiffAugment.params_to_device(rps[k], DEV)
        sums = xd.sum(dim=[1, 2, 3]).contiguous()
        args, kp = A._args(rp, B, xd.device)
        aug = torch.empty_like(xd)
        L.check(lib.dg_diffaug_fwd_pre(xd.data_ptr(), *args, A.mask, B, H, W, sums.data_ptr(), aug.data_ptr(), None))
        assert rel_l2(aug.cpu(), O.diff_augment(xs[k], rps[k])) < 1e-6
        L.check(lib.dg_blur_fwd(aug.data_ptr(), ref[k * B:].data_ptr(), L.dtype_code(dtype), B, H, W, ring, None))
        q = L.DgAugSet()
        q.x, q.xsum = xd.data_ptr(), sums.data_ptr()
        q.u_b, q.u_c, q.t_h, q.t_w, q.o_x, q.o_y = args
        sets.append(q)
        keep += [xd, rp, sums, kp]
    out = torch.empty_like(ref)
    arr = (L.DgAugSet * nsets)(*sets)
    L.check(lib.dg_diffaug_blur_fwd(arr, nsets, A.mask, B, H, W, ring, out.data_ptr(), L.dtype_code(dtype), None))
    torch.cuda.synchronize()
    # same expressions; the compiler may contract a multiply-add differently in the two kernels
    assert float((out.float() - ref.float()).abs().max()) <= (1e-6 if dtype == torch.float32 else 2e-2)
    assert rel_l2(out.float().cpu(), ref.float().cpu()) < (1e-6 if dtype == torch.float32 else 2e-3)


@pytest.mark.parametrize("H,W", [(32, 64), (64, 1024)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_blurvh_adjoint_with_augment_sum(L, H, W, dtype):
    """dg_blur_bwd_augsum + dg_diffaug_bwd_pre == dg_blur_bwd + dg_diffaug_bwd (adjoint, masked sum, gather: the three
    launches of the G phase's way back through BlurVH and DiffAugment)"""
    from dusty_gan_amd.utils.diff_augment import DiffAugment
    lib = L.lib()
    B = 4
    A = DiffAugment()
    g = torch.Generator().manual_seed(H + 5 * W)
    d = torch.randn(B, H, W, 2, generator=g).to(DEV, dtype)
    rp = DiffAugment.params_to_device(_aug_params(B, H, W, 23, extreme=True), DEV)
    args, kp = A._args(rp, B, d.device)
    dx = torch.empty(B, 1, H, W, device=DEV)
    L.check(lib.dg_blur_bwd(d.data_ptr(), L.dtype_code(dtype), dx.data_ptr(), B, H, W, 1, None))
    ws, ref = torch.empty(B, device=DEV), torch.empty_like(dx)
    L.check(lib.dg_diffaug_bwd(dx.data_ptr(), *args, A.mask, B, H, W, ws.data_ptr(), ref.data_ptr(), None))
    dx2, gsum, got = torch.empty_like(dx), torch.zeros(B, device=DEV), torch.empty_like(dx)
    L.check(lib.dg_blur_bwd_augsum(d.data_ptr(), L.dtype_code(dtype), dx2.data_ptr(), args[2], args[4], args[5], A.mask,
                                   gsum.data_ptr(), B, H, W, 1, None))
    L.check(lib.dg_diffaug_bwd_pre(dx2.data_ptr(), *args, A.mask, B, H, W, gsum.data_ptr(), got.data_ptr(), None))
    torch.cuda.synchronize()
    assert torch.equal(dx2, dx)
    assert rel_l2(gsum.cpu(), ws.cpu()) < 1e-5
    assert rel_l2(got.cpu(), ref.cpu()) < 1e-5


@pytest.mark.parametrize("H,W,ring", [(32, 64, 1), (64, 1024, 1), (16, 48, 0), (8, 2048, 1), (6, 64, 1)])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_r1_turnaround_in_one_launch(L, H, W, ring, dtype):
    """dg_blur_r1_tangent (round 6: BlurVH^T, |g|^2, the scaling and the tangent's BlurVH in ONE launch, g never written;
    trainers/dcgan_amp.py:218-235 through models/ops/common.py:74-88) == dg_blur_bwd_r1 + dg_blur_fwd: the tangent map bit for
    bit, the per-sample sums and their mean to rounding; against the oracle's blur_vh; refused shapes launch nothing."""
    lib = L.lib()
    g = torch.Generator().manual_seed(3 * H + W)
    B = 3
    dt = L.dtype_code(dtype)
    d = torch.randn(B, H, W, 2, generator=g).to(DEV, dtype)
    out = torch.full((B, H, W, 2), 7.0, device=DEV, dtype=dtype)
    ssq, macc = torch.zeros(B, device=DEV), torch.zeros(1, device=DEV)
    rc = lib.dg_blur_r1_tangent(d.data_ptr(), dt, out.data_ptr(), 0.375, ssq.data_ptr(), macc.data_ptr(), B, B, H, W, ring, None)
    torch.cuda.synchronize()
    if H % 4 or 6 * W * 4 > 60 * 1024:
        assert rc == L.DG_EUNSUPPORTED and float((out.float() - 7.0).abs().max()) == 0.0
        return
    L.check(rc)
    vg, ssq2 = torch.empty(B, 1, H, W, device=DEV), torch.zeros(B, device=DEV)
    ok = lib.dg_blur_bwd_r1(d.data_ptr(), dt, vg.data_ptr(), 0.375, ssq2.data_ptr(), B, H, W, ring, None)
    if ok == L.DG_EUNSUPPORTED:                         # (H W % 1024 != 0: the plain adjoint, scaled here)
        L.check(lib.dg_blur_bwd(d.data_ptr(), dt, vg.data_ptr(), B, H, W, ring, None))
        ssq2 = (vg.double() ** 2).sum(dim=[1, 2, 3]).float()
        vg = vg * 0.375
    ref = torch.empty(B, H, W, 2, device=DEV, dtype=dtype)
    L.check(lib.dg_blur_fwd(vg.data_ptr(), ref.data_ptr(), dt, B, H, W, ring, None))
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    assert rel_l2(ssq.cpu(), ssq2.cpu()) < 1e-6
    assert abs(float(macc) - float(ssq2.sum()) / B) <= 1e-6 * float(ssq2.sum()) / B
    if ring:   # the oracle: BlurVH of the scaled adjoint image (autograd of the reference composition)
        x = torch.zeros(B, 1, H, W, requires_grad=True)
        y = O.blur_vh(x, ring=True)
        (gx,) = torch.autograd.grad(y, x, d.float().cpu().permute(0, 3, 1, 2))
        want = O.blur_vh(0.375 * gx, ring=True).permute(0, 2, 3, 1)
        assert rel_l2(out.float().cpu(), want) < (1e-2 if dtype == torch.bfloat16 else 1e-5)
    assert lib.dg_blur_r1_tangent(d.data_ptr(), dt, out.data_ptr(), 1.0, None, None, B, B, H, W, ring, None) == L.DG_EINVAL


@pytest.mark.parametrize("mag", [1.0, 1e-3, 3e-5], ids=["unit", "1e-3", "3e-5"])
def test_r1_sums_in_the_arena_keep_their_documented_window(L, mag):
    """The accumulator arena's sums are 32.32 FIXED POINT (csrc/common.h dg_acc_add): order-independent, with an ABSOLUTE
    resolution of 2^-32 per contribution (round-5 advice: say what that means for small magnitudes, and test it).  R1's per-sample
    |g|^2 through dg_blur_r1_tangent with its sums in the arena, for gradient maps of magnitude 1, 1e-3 and 3e-5 (|g|^2 per
    sample ~ 2e4, 2e-2, 2e-5): two launches agree bit for bit at every magnitude, and the sums are within
    contributors x 2^-33 (the rounding of each contribution) + 1e-6 relative (the float partials) of a float64 reference - i.e.
    exact to float precision at ordinary magnitudes and ~1e-4 relative at |g|^2 ~ 2e-5, the regime of a collapsed run, where
    only the LOGGED penalty reads it (the tangent is formed from g itself, asserted here: scaled inputs give the scaled map)."""
    lib = L.lib()
    B, H, W = 4, 64, 256
    g = torch.Generator().manual_seed(17)
    d1 = torch.randn(B, H, W, 2, generator=g)
    d = (d1 * mag).to(DEV)
    outs, sums = [], []
    for _ in range(2):
        L.AccArena.begin(DEV)
        ssq, macc = L.AccArena.take(B, DEV), L.AccArena.take(1, DEV)
        out = torch.empty(B, H, W, 2, device=DEV)
        L.check(lib.dg_blur_r1_tangent(d.data_ptr(), L.DG_F32, out.data_ptr(), 0.5, ssq.data_ptr(), macc.data_ptr(), B, B, H, W, 1, None))
        torch.cuda.synchronize()
        outs.append(out.cpu()); sums.append((ssq.cpu().clone(), macc.cpu().clone()))
    assert torch.equal(outs[0], outs[1]) and torch.equal(sums[0][0], sums[1][0]) and torch.equal(sums[0][1], sums[1][1])
    vg = torch.empty(B, 1, H, W, device=DEV)
    L.check(lib.dg_blur_bwd(d.data_ptr(), L.DG_F32, vg.data_ptr(), B, H, W, 1, None))
    want = (vg.double() ** 2).sum(dim=[1, 2, 3]).cpu()
    err = (sums[0][0].double() - want).abs()
    bound = 1024 * 2.0 ** -33 + 1e-6 * want
    assert (err <= bound).all(), (mag, err.tolist(), want.tolist())
    assert abs(float(sums[0][1]) - float(want.mean())) <= 1024 * 2.0 ** -33 + 1e-6 * float(want.mean())
    if mag != 1.0:   # the tangent map does not go through the sums: it scales with its input
        L.AccArena.begin(DEV)
        s1, m1 = L.AccArena.take(B, DEV), L.AccArena.take(1, DEV)
        o1 = torch.empty(B, H, W, 2, device=DEV)
        L.check(lib.dg_blur_r1_tangent(d1.to(DEV).data_ptr(), L.DG_F32, o1.data_ptr(), 0.5, s1.data_ptr(), m1.data_ptr(), B, B, H, W, 1, None))
        torch.cuda.synchronize()
        assert rel_l2(outs[0], o1.cpu() * mag) < 1e-6


def test_fetch_reals_from_the_device_resident_pool(L):
    """dg_fetch_reals_pool_sum picks batch (*counter % pool) on the device: == dg_fetch_reals_sum of that batch"""
    lib = L.lib()
    P, B, H, W = 3, 2, 8, 32
    g = torch.Generator().manual_seed(9)
    pol = torch.rand(P, B, 1, H, W, generator=g).to(DEV)
    m = (torch.rand(P, B, 1, H, W, generator=g) < 0.8).float().to(DEV)
    for k in (0, 1, 2, 4, 8):
        ctr = torch.full((1,), k, dtype=torch.int64, device=DEV)
        ref, rs = torch.empty(B, 1, H, W, device=DEV), torch.zeros(B, device=DEV)
        L.check(lib.dg_fetch_reals_sum(pol[k % P].data_ptr(), m[k % P].data_ptr(), 0.9, 120.0, -1.0, B, H * W, ref.data_ptr(),
                                       rs.data_ptr(), None))
        out, s = torch.empty_like(ref), torch.zeros(B, device=DEV)
        L.check(lib.dg_fetch_reals_pool_sum(pol.data_ptr(), m.data_ptr(), ctr.data_ptr(), P, 0.9, 120.0, -1.0, B, H * W,
                                            out.data_ptr(), s.data_ptr(), None))
        torch.cuda.synchronize()
        assert torch.equal(out, ref) and rel_l2(s.cpu(), rs.cpu()) < 1e-6


def test_fetch_reals_rides_on_the_step_prologue(L):
    """dg_step_prologue_fetch (round 6): fetch_reals of the pooled batch as more workgroups of the step's first launch -
    the image bit for bit what dg_fetch_reals_pool_sum writes (and the oracle's fetch_reals, trainers/dcgan_amp.py:154-160), the
    per-sample sums as DG_XSUM_PARTS stored partials whose in-order sum is the sample's sum; the zero-fill of the same launch
    still happens; two launches agree bit for bit; and dg_diffaug_blur_fwd reads the partial form like the single sums."""
    lib = L.lib()
    P, B, H, W = 3, 4, 16, 512                       # HW = 8192 = 1024 x DG_XSUM_PARTS
    g = torch.Generator().manual_seed(19)
    pol = torch.rand(P, B, 1, H, W, generator=g).to(DEV)
    m = (torch.rand(P, B, 1, H, W, generator=g) < 0.8).float().to(DEV)
    for k in (0, 2, 7):
        ctr = torch.full((1,), k, dtype=torch.int64, device=DEV)
        ref, rs = torch.empty(B, 1, H, W, device=DEV), torch.zeros(B, device=DEV)
        L.check(lib.dg_fetch_reals_pool_sum(pol.data_ptr(), m.data_ptr(), ctr.data_ptr(), P, 0.9, 120.0, -1.0, B, H * W,
                                            ref.data_ptr(), rs.data_ptr(), None))
        res = []
        for _ in range(2):
            out = torch.full((B, 1, H, W), float("nan"), device=DEV)
            parts = torch.full((B, L.XSUM_PARTS), float("nan"), device=DEV)
            dirty = torch.ones(1024, device=DEV)
            f = L.DgFetch()
            f.pol, f.mask, f.pool_ctr, f.npool = pol.data_ptr(), m.data_ptr(), ctr.data_ptr(), P
            f.min_depth, f.max_depth, f.drop_const, f.B, f.HW = 0.9, 120.0, -1.0, B, H * W
            f.out, f.parts = out.data_ptr(), parts.data_ptr()
            L.step_prologue([dirty], [], fetch=f)
            torch.cuda.synchronize()
            assert float(dirty.abs().max()) == 0.0
            res.append((out, parts))
        out, parts = res[0]
        assert torch.equal(out, ref) and torch.equal(res[1][0], out) and torch.equal(res[1][1], parts)
        want = O.fetch_reals(pol[k % P].cpu(), m[k % P].cpu())[0]
        assert float((out.cpu() - want).abs().max()) < 1e-5
        assert rel_l2(parts.sum(1).cpu(), rs.cpu()) < 1e-6
        assert rel_l2(parts.double().cpu(), out.double().view(B, L.XSUM_PARTS, -1).sum(2).cpu()) < 1e-6
    # the reader: DiffAugment + BlurVH with the partial sums == with their in-order total as the single sum
    from dusty_gan_amd.utils.diff_augment import DiffAugment
    A = DiffAugment()
    rp = A.draw(B, H, W, torch.device(DEV))
    args, keep = A._args(rp, B, torch.device(DEV))
    tot = torch.zeros(B, device=DEV)
    for j in range(L.XSUM_PARTS):                    # (the kernel's own order of additions)
        tot = tot + parts[:, j]
    outs = []
    for xs, np_ in ((parts, L.XSUM_PARTS), (tot, 1)):
        q = L.DgAugSet()
        q.x, q.xsum, q.xsum_parts = out.data_ptr(), xs.data_ptr(), np_
        q.u_b, q.u_c, q.t_h, q.t_w, q.o_x, q.o_y = args
        h0 = torch.empty(B, H, W, 2, device=DEV)
        L.check(lib.dg_diffaug_blur_fwd((L.DgAugSet * 1)(q), 1, A.mask, B, H, W, 1, h0.data_ptr(), L.DG_F32, None))
        outs.append(h0)
    torch.cuda.synchronize()
    assert torch.equal(outs[0], outs[1]) and float(outs[0].abs().mean()) > 0
    # refusals
    f.HW = 4096
    assert lib.dg_step_prologue_fetch(None, None, 0, None, 0, C.byref(f), None) == L.DG_EUNSUPPORTED
    assert lib.dg_step_prologue_fetch(None, None, 0, None, 0, None, None) == L.DG_EINVAL


@pytest.mark.parametrize("shape", [(3, 1, 8, 32), (2, 1, 1, 1), (5, 1, 7, 9)])
def test_logistic_noise_in_one_launch(L, shape):
    """Philox.logistic_noise (dg_philox_logistic_dev: U1, U2 drawn and combined in one launch) == two uniform fills of the
    same stream followed by dg_logistic_noise (GumbelSigmoid.logistic_noise, models/dusty.py:30-36), and the same counter
    position afterwards - also for element counts that are not multiples of 4"""
    from dusty_gan_amd.utils.rng import Philox
    n = 1
    for s in shape:
        n *= s
    a, b = Philox(1234, torch.device(DEV), stream_id=3), Philox(1234, torch.device(DEV), stream_id=3)
    a.uniform(5)
    b.uniform(5)                                   # (a non-zero starting offset)
    got = a.logistic_noise(shape)
    u1, u2 = b.uniform(n), b.uniform(n)
    want = torch.empty(n, device=DEV)
    L.check(L.lib().dg_logistic_noise(u1.data_ptr(), u2.data_ptr(), 1e-10, n, want.data_ptr(), None))
    torch.cuda.synchronize()
    assert got.shape == tuple(shape) and torch.equal(got.flatten(), want)
    assert a.offset == b.offset
    assert rel_l2(got.flatten().cpu(), O.logistic_noise(u1.cpu(), u2.cpu())) < 1e-5


def test_step_prologue_equals_the_single_launches(L):
    """dg_step_prologue: zero-fill + latents (fp32 and the bfloat16 copy) + two logistic-noise tensors + DiffAugment
    parameters in one launch are bit for bit what dg_zero_multi and the single draws (Philox.normal / logistic_noise /
    DiffAugment.draw_sets) produce from the same counters, and the counters end where the single draws leave them."""
    from dusty_gan_amd.utils.diff_augment import DiffAugment
    from dusty_gan_amd.utils.rng import Philox
    B, nz, H, W = 6, 20, 16, 64            # (nz B not a multiple of 4 x 256: partial last block)
    def gens():
        r, A = Philox(4242, DEV, stream_id=1), DiffAugment(seed=99)
        r.normal(7)                        # a non-zero starting offset
        A.draw(3, H, W, torch.device(DEV))
        return r, A
    r1, A1 = gens()
    z1 = r1.normal(B * nz)
    px1 = r1.logistic_noise((B, 1, H, W))
    im1 = r1.logistic_noise((B, 1, 1, 1))
    sets1 = A1.draw_sets(4, B, H, W, torch.device(DEV))
    r2, A2 = gens()
    ra = A2.rng(torch.device(DEV))
    r2.sync(); ra.sync()
    z2 = torch.empty(B * nz, device=DEV)
    zb = torch.empty(B * nz, device=DEV, dtype=torch.bfloat16)
    px2, im2 = torch.empty(B, 1, H, W, device=DEV), torch.empty(B, 1, 1, 1, device=DEV)
    uf, qi = torch.empty(3, 4 * B, device=DEV), torch.empty(4, 4 * B, device=DEV, dtype=torch.int32)
    base = 0
    jobs = [r2.job(0, base, fill_kind=1, n=B * nz, out=z2.data_ptr(), out_bf16=zb.data_ptr())]
    base += (B * nz + 3) // 4
    jobs.append(r2.job(1, base, eps=1e-10, n=B * H * W, out=px2.data_ptr()))
    base += 2 * ((B * H * W + 3) // 4)
    jobs.append(r2.job(1, base, eps=1e-10, n=B, out=im2.data_ptr()))
    base += 2 * ((B + 3) // 4)
    jobs.append(ra.job(2, 0, B=4 * B, H=H, W=W, uf=uf.data_ptr(), qi=qi.data_ptr()))
    a, b = torch.full((4096,), 3.0, device=DEV), torch.full((260,), 5.0, device=DEV)
    L.step_prologue([a, b], jobs)
    r2.advance(base); ra.advance(2 * 4 * B)
    torch.cuda.synchronize()
    assert float(a.abs().max()) == 0.0 and float(b.abs().max()) == 0.0
    assert torch.equal(z1, z2) and torch.equal(zb, z1.bfloat16())
    assert torch.equal(px1, px2) and torch.equal(im1, im2)
    for s1, s2 in zip(sets1, DiffAugment.sets_of(uf, qi, 4, B)):
        for k in s1:
            assert torch.equal(s1[k], s2[k]), k
    assert r1.offset == r2.offset and A1.rng(torch.device(DEV)).offset == ra.offset
    # draws only / zero only / argument errors
    L.step_prologue([], [r2.job(0, 0, fill_kind=0, n=5, out=z2.data_ptr())])
    torch.cuda.synchronize()
    assert float(z2[:5].min()) >= 0.0 and float(z2[:5].max()) < 1.0 and torch.equal(z2[5:], z1[5:])
    a.fill_(1.0)
    L.step_prologue([a], [])
    assert float(a.abs().max()) == 0.0
    bad = r2.job(3, 0, n=4, out=z2.data_ptr())
    assert L.lib().dg_step_prologue(None, None, 0, (L.DgDraw * 1)(bad), 1, None) == L.DG_EINVAL
    assert L.lib().dg_step_prologue(None, None, 0, (L.DgDraw * 1)(bad), 7, None) == L.DG_EINVAL


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("metric,mode_g,r1", [(0, 0, 1), (0, 0, 0), (2, 0, 1), (4, 0, 0), (0, 1, 0), (5, 1, 0)])
def test_final_gan_bwd_equals_the_separate_launches(L, dtype, metric, mode_g, r1):
    """dg_final_gan_bwd (loss step + final conv backward-data + its weight gradient in one launch, every block evaluating
    the loss step for itself) against dg_gan_d_step / dg_gan_g_step + dg_final_bwd_data + dg_batch_wsum: the per-sample
    vectors, scalars and the data gradient bit for bit, the atomically summed bias gradient to rounding."""
    lib = L.lib()
    g = torch.Generator().manual_seed(metric * 7 + mode_g)
    B, C, n = 12, 32, 32 * 4 * 16                       # n = C h0 w0
    ns = B if mode_g else 2 * B
    dt = L.dtype_code(dtype)
    y = torch.randn(2 * B, generator=g).to(DEV)
    d4 = torch.randn(ns, n, generator=g).to(DEV, dtype)
    wf = torch.randn(n, generator=g).to(DEV)
    scale, w_gan, smoothing = 1.0 / math.sqrt(n), 0.7, 0.9
    rel = metric >= 4

    def bufs():
        return dict(dy=torch.zeros(ns, device=DEV), up=torch.zeros(ns, device=DEV), rs=torch.zeros(ns, device=DEV),
                    acc=torch.zeros(4, device=DEV), fb=torch.zeros(1, device=DEV), dd4=torch.zeros(ns, n, device=DEV, dtype=dtype),
                    db=torch.zeros(C, device=DEV), dwf=torch.zeros(n, device=DEV))
    a, b = bufs(), bufs()
    y_real, y_fake = (y.data_ptr() + 4 * B if (mode_g and rel) else (None if mode_g else y.data_ptr()),
                      y.data_ptr() if mode_g else y.data_ptr() + 4 * B)
    # separate launches
    if mode_g:
        L.check(lib.dg_gan_g_step(metric, y_real, y_fake, B, w_gan, a["dy"].data_ptr(), a["acc"].data_ptr(), None))
        L.check(lib.dg_final_bwd_data(d4.data_ptr(), dt, wf.data_ptr(), a["dy"].data_ptr(), None, scale, ns, n, C,
                                      a["dd4"].data_ptr(), None, None))
    else:
        L.check(lib.dg_gan_d_step(metric, smoothing, y_real, y_fake, B, w_gan, a["dy"].data_ptr(),
                                  a["up"].data_ptr() if r1 else None, a["rs"].data_ptr() if r1 else None,
                                  a["acc"].data_ptr(), a["fb"].data_ptr(), None))
        L.check(lib.dg_final_bwd_data(d4.data_ptr(), dt, wf.data_ptr(), (a["up"] if r1 else a["dy"]).data_ptr(),
                                      a["rs"].data_ptr() if r1 else None, scale, ns, n, C, a["dd4"].data_ptr(),
                                      a["db"].data_ptr(), None))
        L.check(lib.dg_batch_wsum(d4.data_ptr(), dt, a["dy"].data_ptr(), scale, ns, n, a["dwf"].data_ptr(), None))
    # one launch
    L.check(lib.dg_final_gan_bwd(metric, mode_g, smoothing, y_real, y_fake, B, w_gan, r1, b["dy"].data_ptr(),
                                 b["up"].data_ptr() if r1 else None, b["rs"].data_ptr() if r1 else None, b["acc"].data_ptr(),
                                 None if mode_g else b["fb"].data_ptr(), d4.data_ptr(), dt, wf.data_ptr(), scale, n, C,
                                 b["dd4"].data_ptr(), None if mode_g else b["db"].data_ptr(),
                                 None if mode_g else b["dwf"].data_ptr(), None, None))
    torch.cuda.synchronize()
    for k in ("dy", "up", "rs", "acc", "fb", "dd4"):
        assert torch.equal(a[k], b[k]), k
    # (round 6: eight waves split the samples instead of four - the weight gradient's partial sums meet in another order)
    assert rel_l2(b["dwf"].cpu(), a["dwf"].cpu()) < 1e-6 or mode_g
    assert float(a["dd4"].float().abs().mean()) > 0 and (mode_g or float(a["dwf"].abs().mean()) > 0)
    assert rel_l2(b["db"].cpu(), a["db"].cpu()) < 1e-5 or mode_g
    if not mode_g:
        # round 5, dbias_part: one partial per element of the map instead of atomics onto dbias, summed per channel by
        # dg_wgrad_reduce - the same bias gradient, and two launches of it agree bit for bit
        res = []
        for _ in range(2):
            c = bufs()
            part = torch.full((n,), float("nan"), device=DEV)
            L.check(lib.dg_final_gan_bwd(metric, mode_g, smoothing, y_real, y_fake, B, w_gan, r1, c["dy"].data_ptr(),
                                         c["up"].data_ptr() if r1 else None, c["rs"].data_ptr() if r1 else None,
                                         c["acc"].data_ptr(), c["fb"].data_ptr(), d4.data_ptr(), dt, wf.data_ptr(), scale, n, C,
                                         c["dd4"].data_ptr(), c["db"].data_ptr(), c["dwf"].data_ptr(), part.data_ptr(), None))
            item = (L.DgWgradReduce * 1)()
            item[0].ws, item[0].dw, item[0].numel, item[0].splits, item[0].accumulate = part.data_ptr(), c["db"].data_ptr(), C, n // C, 1
            L.check(lib.dg_wgrad_reduce(item, 1, None))
            torch.cuda.synchronize()
            assert torch.equal(c["dd4"], b["dd4"]) and torch.equal(c["dwf"], b["dwf"])
            res.append(c["db"].clone())
        assert rel_l2(res[0].cpu(), a["db"].cpu()) < 1e-5
        assert torch.equal(res[0], res[1])
    # shapes the kernel refuses: nothing launched
    big = 200
    yb = torch.randn(2 * big, device=DEV)
    rc = lib.dg_final_gan_bwd(0, 0, 1.0, yb.data_ptr(), yb.data_ptr() + 4 * big, big, 1.0, 0, b["dy"].data_ptr(), None, None,
                              b["acc"].data_ptr(), None, d4.data_ptr(), dt, wf.data_ptr(), scale, n, C, b["dd4"].data_ptr(),
                              None, None, None, None)
    assert rc == L.DG_EUNSUPPORTED
    assert lib.dg_final_gan_bwd(0, 1, 1.0, None, y_fake, B, 1.0, 1, b["dy"].data_ptr(), None, None, b["acc"].data_ptr(), None,
                                d4.data_ptr(), dt, wf.data_ptr(), scale, n, C, b["dd4"].data_ptr(), None, None,
                                None, None) == L.DG_EINVAL


@pytest.mark.parametrize("with_ema,co", [(False, 128), (True, 64), (True, 256)])
@pytest.mark.parametrize("sdt", [torch.bfloat16, torch.float32])
def test_one_launch_optimizer_is_the_four_launches(L, with_ema, co, sdt):
    """dg_adam_fused (round 6) against the launches it replaces - dg_batch_wsum, dg_wgrad_reduce, dg_adam_ema_step_dev,
    dg_transpose_shadow_multi - on a parameter buffer with a fat conv segment (split-K partial tiles), a bias with more partial
    rows than the wide path's threshold, a plain run and a segment that takes an extra batch term (the final conv's R1 term):
    master, exp_avg_sq, EMA, gradient, shadow and transposed shadow bit for bit (the sums are taken in the same order), the
    extra-term segment to rounding; the reference arithmetic: torch.optim.Adam at beta1 = 0 + ema_inplace,
    trainers/dcgan_amp.py:30-35,116-125."""
    lib = L.lib()
    g = torch.Generator().manual_seed(77)
    ci = 48 if co != 64 else 32
    n_conv, n_b, n_plain, n_ws = 16 * ci * co, 96, 320, 2048
    offs = [0, n_conv, n_conv + 128, n_conv + 128 + n_plain]
    n = offs[3] + n_ws
    sp_conv, sp_b, nb = 5, 100, 6

    def fresh():
        gg = torch.Generator().manual_seed(5)
        d = dict(p=torch.randn(n, generator=gg), v=torch.rand(n, generator=gg) * 1e-2, grad=torch.randn(n, generator=gg) * 0.1,
                 ema=torch.randn(n, generator=gg))
        d = {k: t.to(DEV) for k, t in d.items()}
        d["shadow"] = torch.zeros(n, dtype=sdt, device=DEV)
        d["coci"] = torch.zeros(n_conv, dtype=sdt, device=DEV)
        return d
    part_conv = (torch.randn(sp_conv, n_conv, generator=g) * 0.1).to(DEV)
    part_b = (torch.randn(sp_b, n_b, generator=g) * 0.1).to(DEV)
    src = torch.randn(nb, n_ws, generator=g).to(DEV, torch.bfloat16)
    coef = torch.randn(nb, generator=g).to(DEV)
    step = torch.full((1,), 3, dtype=torch.int64, device=DEV)
    lr, b2, eps, decay, gscale = 2e-3, 0.99, 1e-8, 0.998, 0.5
    code = L.dtype_code(sdt)
    # the four launches
    a = fresh()
    L.check(lib.dg_batch_wsum(src.data_ptr(), L.DG_BF16, coef.data_ptr(), 0.25, nb, n_ws, a["grad"].data_ptr() + 4 * offs[3], None))
    items = (L.DgWgradReduce * 2)()
    items[0].ws, items[0].dw, items[0].numel, items[0].splits, items[0].accumulate = part_conv.data_ptr(), a["grad"].data_ptr(), n_conv, sp_conv, 1
    items[1].ws, items[1].dw, items[1].numel, items[1].splits, items[1].accumulate = part_b.data_ptr(), a["grad"].data_ptr() + 4 * offs[1], n_b, sp_b, 0
    L.check(lib.dg_wgrad_reduce(items, 2, None))
    L.check(lib.dg_adam_ema_step_dev(a["p"].data_ptr(), a["grad"].data_ptr(), None, a["v"].data_ptr(),
                                     a["ema"].data_ptr() if with_ema else None, a["shadow"].data_ptr(), code, n, gscale, lr, 0.0,
                                     b2, eps, step.data_ptr(), decay, None))
    desc = torch.tensor([0, a["coci"].data_ptr(), ci, co, 0], dtype=torch.int64).to(DEV)
    L.check(lib.dg_transpose_shadow_multi(a["p"].data_ptr(), desc.data_ptr(), 1, 16 * ((ci + 31) // 32) * ((co + 31) // 32), code, None))
    # one launch
    b = fresh()
    segs = (L.DgOptSeg * 4)()
    segs[0].off, segs[0].numel, segs[0].part, segs[0].splits, segs[0].accumulate = 0, n_conv, part_conv.data_ptr(), sp_conv, 1
    segs[0].kind, segs[0].ci, segs[0].co, segs[0].shadow_t = 1, ci, co, b["coci"].data_ptr()
    segs[1].off, segs[1].numel, segs[1].part, segs[1].splits, segs[1].accumulate = offs[1], n_b, part_b.data_ptr(), sp_b, 0
    segs[2].off, segs[2].numel, segs[2].accumulate = offs[1] + n_b, offs[3] - offs[1] - n_b, 1      # the plain run (+ padding)
    segs[3].off, segs[3].numel, segs[3].accumulate = offs[3], n_ws, 1
    segs[3].ws_src, segs[3].ws_bf16, segs[3].ws_coef, segs[3].ws_n, segs[3].ws_stride, segs[3].ws_scale = src.data_ptr(), 1, coef.data_ptr(), nb, n_ws, 0.25
    for _ in range(1):
        L.check(lib.dg_adam_fused(b["p"].data_ptr(), b["grad"].data_ptr(), b["v"].data_ptr(), b["ema"].data_ptr() if with_ema else None,
                                  b["shadow"].data_ptr(), code, segs, 4, gscale, lr, b2, eps, step.data_ptr(), decay, None))
    torch.cuda.synchronize()
    lo = offs[3]
    assert torch.equal(a["grad"][:lo], b["grad"][:lo])       # the sums: same additions in the same order
    for k in ("p", "v", "grad", "ema", "shadow"):
        # (Adam's arithmetic may be contracted into fused multiply-adds differently in the two kernels: last-bit differences)
        tol = 1e-2 if (k == "shadow" and sdt == torch.bfloat16) else 1e-6
        assert rel_l2(b[k][:lo].float().cpu(), a[k][:lo].float().cpu()) < tol, k
        assert rel_l2(b[k][lo:].float().cpu(), a[k][lo:].float().cpu()) < max(tol, 1e-5), k
    assert torch.equal(b["shadow"], b["p"].to(sdt))          # the shadows ARE the updated master, rounded
    assert torch.equal(b["coci"].view(16, co, ci), b["p"][:n_conv].view(16, ci, co).to(sdt).permute(0, 2, 1).contiguous())
    # against the formulas (fp64)
    f = fresh()
    gsum = f["grad"].double().cpu()
    gsum[:n_conv] += part_conv.double().cpu().sum(0)
    gsum[offs[1]:offs[1] + n_b] = part_b.double().cpu().sum(0)
    gsum[lo:] += 0.25 * (coef.double().cpu()[:, None] * src.double().cpu()).sum(0)
    gg = gsum * gscale
    v = b2 * f["v"].double().cpu() + (1 - b2) * gg * gg
    pnew = f["p"].double().cpu() - lr * gg / (v.sqrt() / math.sqrt(1 - b2 ** 4) + eps)
    assert rel_l2(b["grad"].double().cpu(), gsum) < 1e-6 and rel_l2(b["v"].double().cpu(), v) < 1e-6
    assert rel_l2(b["p"].double().cpu(), pnew) < 1e-6
    if with_ema:
        assert rel_l2(b["ema"].double().cpu(), decay * f["ema"].double().cpu() + (1 - decay) * pnew) < 1e-6
    else:
        assert torch.equal(b["ema"], f["ema"])
    # argument errors
    segs[0].co = 40
    assert lib.dg_adam_fused(b["p"].data_ptr(), b["grad"].data_ptr(), b["v"].data_ptr(), None, b["shadow"].data_ptr(), code, segs, 4,
                             gscale, lr, b2, eps, step.data_ptr(), decay, None) == L.DG_EINVAL
    assert lib.dg_adam_fused(b["p"].data_ptr(), b["grad"].data_ptr(), b["v"].data_ptr(), None, b["shadow"].data_ptr(), code, segs, 0,
                             gscale, lr, b2, eps, step.data_ptr(), decay, None) == L.DG_EINVAL


def test_transpose_shadow_multi_tail_carries_the_counters(L):
    """dg_transpose_shadow_multi_tail: the shadow-refresh launch with one more block that advances counters and files a
    snapshot exactly as dg_counter_add_multi_snap does; and the host-side queue: a flagged ride is taken by the refresh
    behind an optimizer step, a mid-step flush (a consumer syncing its counter) leaves the riding snapshot queued."""
    lib = L.lib()
    ci, co = 48, 40
    master = torch.randn(16, ci, co, device=DEV)
    coci = torch.empty(16 * ci * co, dtype=torch.bfloat16, device=DEV)
    desc = torch.tensor([0, coci.data_ptr(), ci, co, 0], dtype=torch.int64).to(DEV)
    tiles = 16 * ((ci + 31) // 32) * ((co + 31) // 32)
    c = [torch.full((1,), v, dtype=torch.int64, device=DEV) for v in (7, 100)]
    ptrs = (C.c_void_p * 2)(*[t.data_ptr() for t in c])
    ring = torch.zeros(4, 8).pin_memory()
    src = torch.arange(8, dtype=torch.float32, device=DEV) + 1.0
    L.check(lib.dg_transpose_shadow_multi_tail(master.data_ptr(), desc.data_ptr(), 1, tiles, L.DG_BF16, None, 0, ptrs,
                                               (C.c_uint64 * 2)(1, 5), 2, 0, src.data_ptr(), 8, ring.data_ptr(), 4, None))
    torch.cuda.synchronize()
    assert torch.equal(coci.view(16, co, ci), master.bfloat16().permute(0, 2, 1).contiguous())
    assert [int(t) for t in c] == [8, 105] and ring[7 % 4].tolist() == [float(i + 1) for i in range(8)]
    L.check(lib.dg_transpose_shadow_multi_tail(master.data_ptr(), desc.data_ptr(), 1, tiles, L.DG_BF16, None, 0, ptrs,
                                               (C.c_uint64 * 2)(1, 5), 2, -1, None, 0, None, 1, None))      # no snapshot
    torch.cuda.synchronize()
    assert [int(t) for t in c] == [9, 110]
    assert lib.dg_transpose_shadow_multi_tail(master.data_ptr(), desc.data_ptr(), 1, tiles, L.DG_BF16, None, 0, ptrs,
                                              (C.c_uint64 * 2)(1, 5), 9, -1, None, 0, None, 1, None) == L.DG_EINVAL
    # host-side queue
    from dusty_gan_amd import engine as E
    L.Counters.flush()
    st = E.ParamStore(E.d_segments(1, [64, 128, 256, 512], (64, 256)))
    st.apply(lambda t: t.to(DEV))
    st.flat.normal_()
    st.refresh_shadows(torch.bfloat16)
    other, snapc = torch.zeros(1, dtype=torch.int64, device=DEV), torch.full((1,), 2, dtype=torch.int64, device=DEV)
    L.Counters.add(other, 3)
    L.Counters.add(snapc, 1)
    L.Counters.snapshot(snapc, src.data_ptr(), 8, ring.data_ptr(), 4)
    L.Counters.ride = True
    ring.zero_()
    L.Counters.flush_if(other)                    # a consumer of `other` in the middle of the step
    torch.cuda.synchronize()
    assert int(other) == 3 and int(snapc) == 2 and float(ring.abs().max()) == 0.0 and L.Counters.ride
    L.Counters.add(other, 1)
    st.refresh_transposed()                       # not the optimizer's refresh: nothing rides
    torch.cuda.synchronize()
    assert int(snapc) == 2 and L.Counters.ride
    st.refresh_transposed(tail=True)
    torch.cuda.synchronize()
    assert int(other) == 4 and int(snapc) == 3 and ring[2].tolist() == [float(i + 1) for i in range(8)]
    assert not L.Counters.pending and L.Counters.snap is None and not L.Counters.ride


@pytest.mark.parametrize("policy", [None, [], ["translation"], ["brightness", "contrast", "cutout"]],
                         ids=["default", "none", "translation", "no-translation"])
@pytest.mark.parametrize("arch", [0, 1, 2])
def test_head_post_bwd_aug_equals_gather_then_head_post_bwd(L, arch, policy):
    """dg_head_post_bwd_aug (DiffAugment's adjoint gather evaluated inside the head post-processing's backward) against
    dg_diffaug_bwd_pre followed by dg_head_post_bwd, incl. extreme draws (clamped cut-out boxes, maximal shifts, rows
    shifted out of the image) and the column the reference's `% (W - 1)` reads twice."""
    from dusty_gan_amd.utils.diff_augment import DiffAugment
    lib = L.lib()
    g = torch.Generator().manual_seed(5 + arch)
    B, H, W = 4, 16, 64
    A = DiffAugment() if policy is None else DiffAugment(policy=policy)
    gy = torch.randn(B, 1, H, W, generator=g).to(DEV)
    gsum = torch.randn(B, generator=g).to(DEV)
    gout = torch.randn(B, 1 + arch, H, W, generator=g).to(DEV)
    gout[:, 0] = torch.tanh(gout[:, 0])
    npx, nim = torch.randn(B, 1, H, W, generator=g).to(DEV), torch.randn(B, generator=g).to(DEV)
    mask = (torch.rand(B, max(arch, 1), H, W, generator=g) > 0.4).float().to(DEV)
    cp = 2 if arch <= 1 else 4
    for trial in range(3):
        rp = O.draw_augment_params(B, H, W, g)
        if trial == 0:
            sh, sw = O.translation_shift(H, W)
            rp["o_x"][:] = torch.tensor([0, H, 0, H])
            rp["o_y"][:] = torch.tensor([0, W, W, 0])
            rp["t_h"][:] = torch.tensor([-sh, sh, 0, 1])
            rp["t_w"][:] = torch.tensor([-sw, sw, 1, 0])
        rpd = DiffAugment.params_to_device(rp, DEV)
        ddepth = A.backward_pre(gy, rpd, gsum)
        args, keep = A._args(rpd, B, gy.device)
        outs = []
        for fused in (False, True):
            pm = torch.full((B, H, W, cp), 3.0, device=DEV, dtype=torch.bfloat16)
            draw = torch.full((B, 1 + arch, H, W), 3.0, device=DEV)
            db = torch.zeros(3, device=DEV)
            ws = torch.zeros(B * 1024, device=DEV)
            head = (gout.data_ptr(), npx.data_ptr(), nim.data_ptr(), mask.data_ptr())
            tail = (0.25, 0.125, draw.data_ptr(), db.data_ptr(), pm.data_ptr(), cp, ws.data_ptr(), None)
            if fused:
                L.check(lib.dg_head_post_bwd_aug(*head, gy.data_ptr(), *args, A.mask, gsum.data_ptr(), arch, 1.0, -1.0, B, H, W,
                                                 *tail))
            else:
                L.check(lib.dg_head_post_bwd(*head, ddepth.data_ptr(), arch, 1.0, -1.0, B, H * W, *tail))
            torch.cuda.synchronize()
            outs.append((pm.float().cpu(), draw.cpu(), db.cpu()))
        assert float(outs[0][1].abs().mean()) > 0
        assert rel_l2(outs[1][1], outs[0][1]) < 1e-6 and rel_l2(outs[1][0], outs[0][0]) < 1e-3
        assert rel_l2(outs[1][2][:1 + arch], outs[0][2][:1 + arch]) < 1e-5
    # the scalar form's shapes are refused, nothing launched
    assert lib.dg_head_post_bwd_aug(gout.data_ptr(), npx.data_ptr(), nim.data_ptr(), mask.data_ptr(), gy.data_ptr(), *args,
                                    A.mask, gsum.data_ptr(), arch, 1.0, -1.0, B, H, W - 2, 0.25, 0.125, draw.data_ptr(),
                                    db.data_ptr(), None, 0, None, None) == L.DG_EUNSUPPORTED
