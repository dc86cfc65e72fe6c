"""GPU parity of the thin (<= 4-channel side) LDS/VALU kernels of csrc/conv_thin.hip against the CPU oracle:
Down1 forward / backward-data / weight gradient (Cin = 2) and Head forward / backward-data / weight gradient
(Cout = 1..3), in fp32 and bf16 feature-map storage, forced through the thin path (force = 3)."""
import math

import pytest
import torch

from oracle import dusty_oracle as O
from tests.golden_util import rel_l2
from tests.test_gpu_ops import DEV, from_nhwc, nhwc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def L():
    from dusty_gan_amd import _lib
    _lib.lib()
    return _lib


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("Hc,Wc,B", [(4, 64, 3), (16, 128, 2), (4, 1024, 1),   # 1024: 2048-wide images (> 64 KiB LDS in the wgrad)
                                     (32, 512, 24)])  # the bench geometry: 5-6 tiles per wave, i.e. the steady state of
def test_down1_thin(L, dtype, Hc, Wc, B):            # thin_s2_mfma's two-tiles-ahead prefetch (counted vmcnt waits)
    from dusty_gan_amd.engine import Ops
    if B > 8 and dtype == torch.float32:
        pytest.skip("the large case is there for the bf16 matrix-core kernel")
    g = torch.Generator().manual_seed(Hc + Wc)
    Ci, Co = 2, 64
    tol = 1e-4 if dtype == torch.float32 else 2e-2
    x = torch.randn(B, Ci, 2 * Hc, 2 * Wc, generator=g)
    w = torch.randn(Co, Ci, 4, 4, generator=g)
    b = torch.randn(Co, generator=g)
    if dtype == torch.bfloat16:
        x = x.bfloat16().float()
    xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    wq = w.bfloat16().float() if dtype == torch.bfloat16 else w  # forward reads the T shadow
    y = O.down(xr, wq.clone().requires_grad_(), br, True)
    s = 1.0 / math.sqrt(Ci * 16)
    o = Ops(dtype)
    o.force = 3
    xd = nhwc(x).to(DEV, dtype)
    master = w.permute(2, 3, 1, 0).contiguous().to(DEV)  # [ky][kx][ci][co] fp32
    coci = w.permute(2, 3, 0, 1).contiguous().to(DEV, dtype)  # [tap][co][ci] shadow
    out = torch.empty(B * Hc * Wc * Co, device=DEV, dtype=dtype)
    bd = b.to(DEV)
    from dusty_gan_amd.engine import MaskBits
    from tests.test_gpu_ops import pack_bits
    obits = MaskBits.register(out)               # bf16: the launch also leaves the saved 1-bit leaky-relu mask (DgConv.mask_out)
    o.conv(L.MODE_S2, 0, True, B, Hc, Wc, Ci, Co, xd, (4 * Hc * Wc * Ci, Ci, 1), out, (Hc * Wc * Co, Co, 1),
           coci.data_ptr(), s, L.EPI_LRELU, bias=bd.data_ptr(), bias_mod=Co)
    torch.cuda.synchronize()
    if obits is not None:
        assert torch.equal(obits, pack_bits(out)), "mask_out differs from (out > 0)"
    assert rel_l2(from_nhwc(out.float().cpu(), B, Co, Hc, Wc), y) < tol
    # backward-data (MODE_UP adjoint, N = 2) straight from the fp32 master weights
    y32 = O.down(xr, wr, br, True)
    gy = torch.randn(y32.shape, generator=g)
    lr = torch.where(y32 > 0, 1.0, 0.2) * math.sqrt(2.0)
    e = gy * lr
    if dtype == torch.bfloat16:
        e = e.bfloat16().float()
    gx, gw = torch.autograd.grad(y32, [xr, wr], e / lr)
    ed = nhwc(e).to(DEV, dtype)
    dx = torch.empty(B * 4 * Hc * Wc * Ci, device=DEV, dtype=dtype)
    cico = master.to(dtype)  # backward-data reads the T shadow [tap][n = ci][k = co]
    o.conv(L.MODE_UP, 1, True, B, Hc, Wc, Co, Ci, ed, (Hc * Wc * Co, Co, 1), dx, (4 * Hc * Wc * Ci, Ci, 1),
           cico.data_ptr(), s, L.EPI_LINEAR)
    torch.cuda.synchronize()
    if dtype == torch.bfloat16:
        # the kernel applies the LINEAR adjoint to e with the bf16 shadow weights; differentiate the linear part
        # only (going through O.down would re-derive the leaky-relu slopes from the bf16-weight forward, which flips
        # a few of them relative to `lr`)
        y_lin = torch.nn.functional.conv2d(O.pad_ring(xr, True) * s, wq, None, 2, 0)
        (gx,) = torch.autograd.grad(y_lin, xr, e)
    assert rel_l2(from_nhwc(dx.float().cpu(), B, Ci, 2 * Hc, 2 * Wc), gx) < (1e-4 if dtype == torch.float32 else 1e-2)
    # weight gradient with per-sample weights
    rs = torch.rand(B, generator=g) + 0.5
    dw = torch.zeros(16, Ci, Co, device=DEV)
    rsd = rs.to(DEV)
    o.wgrad(0, True, B, Hc, Wc, Ci, Co, xd, (4 * Hc * Wc * Ci, Ci, 1), ed, (Hc * Wc * Co, Co, 1), dw.data_ptr(), s,
            rowscale=rsd)
    torch.cuda.synchronize()
    (gw2,) = torch.autograd.grad(O.down(xr, wr, br, True), wr, (e * rs.view(B, 1, 1, 1)) / lr)
    assert rel_l2(dw.cpu().view(4, 4, Ci, Co).permute(3, 2, 0, 1), gw2) < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("nh,Hc,Wc", [(1, 8, 64), (2, 8, 64), (3, 8, 64), (2, 4, 256), (1, 3, 192)])  # 256: two column
def test_head_thin(L, dtype, nh, Hc, Wc):                                 # parts of two pipelined tiles; 192: one of three
    from dusty_gan_amd.engine import Ops
    g = torch.Generator().manual_seed(nh)
    B, C0 = 2, 64
    tol = 1e-4 if dtype == torch.float32 else 1e-2
    x = torch.randn(B, C0, Hc, Wc, generator=g)
    if dtype == torch.bfloat16:
        x = x.bfloat16().float()
    ws = [torch.randn(C0, 1, 4, 4, generator=g)] + ([torch.randn(C0, nh - 1, 4, 4, generator=g)] if nh > 1 else [])
    bs = [torch.randn(1, generator=g)] + ([torch.randn(nh - 1, generator=g)] if nh > 1 else [])
    xr = x.clone().requires_grad_()
    wr = [w.clone().requires_grad_() for w in ws]
    ys = [O.head(xr, w, b, True) for w, b in zip(wr, bs)]
    y = torch.cat(ys, dim=1)  # [B,nh,2Hc,2Wc]
    y_graph = y
    master = torch.cat(ws, dim=1).permute(2, 3, 0, 1).contiguous().to(DEV)  # [ky][kx][ci][co]
    bias = torch.cat(bs).to(DEV)
    scales = [1.0 / math.sqrt(16)] + [1.0 / math.sqrt(max(nh - 1, 1) * 16)] * (nh - 1)
    nscale = torch.tensor(scales, device=DEV)
    o = Ops(dtype)
    o.force = 3
    xd = nhwc(x).to(DEV, dtype)
    HW = 4 * Hc * Wc
    out = torch.empty(B, nh, 2 * Hc, 2 * Wc, device=DEV)
    coci = torch.cat(ws, dim=1).permute(2, 3, 1, 0).contiguous().to(DEV, dtype)  # forward shadow [tap][n = co][k = ci]
    o.conv(L.MODE_UP, 0, True, B, Hc, Wc, C0, nh, xd, (Hc * Wc * C0, C0, 1), out, (nh * HW, 1, HW), coci.data_ptr(),
           1.0, L.EPI_LINEAR, bias=bias.data_ptr(), bias_mod=nh, out_dt=L.DG_F32, nscale=nscale)
    torch.cuda.synchronize()
    if dtype == torch.bfloat16:
        y = torch.cat([O.head(x, w.bfloat16().float(), b, True) for w, b in zip(ws, bs)], dim=1)
    assert rel_l2(out.cpu(), y) < tol
    # backward: draw = s_n * dL/dy (planar fp32) -> gradient w.r.t. the previous pre-activation, masked, + bias sums
    gy = torch.randn(y.shape, generator=g)
    grads = torch.autograd.grad(y_graph, [xr] + wr, gy)
    gx, gws = grads[0], grads[1:]
    draw = (gy * torch.tensor(scales).view(1, nh, 1, 1)).to(DEV).contiguous()
    prev = torch.randn(x.shape, generator=g)
    prevd = nhwc(prev).to(DEV, dtype)
    shadow = master.to(dtype)  # cico shadow [tap][ci][co]: n = ci, k = co
    dp = torch.empty(B * Hc * Wc * C0, device=DEV, dtype=dtype)
    db = torch.zeros(C0, device=DEV)
    o.conv(L.MODE_S2, 1, True, B, Hc, Wc, nh, C0, draw, (nh * HW, 1, HW), dp, (Hc * Wc * C0, C0, 1), shadow.data_ptr(),
           1.0, L.EPI_MASK, aux=prevd, dbias=db.data_ptr(), bias_mod=C0, in_dt=L.DG_F32)
    torch.cuda.synchronize()
    if dtype == torch.bfloat16:  # the kernel reads bf16-rounded weights
        wq = [w.bfloat16().float().requires_grad_() for w in ws]
        yq = torch.cat([O.head(xr, w, b, True) for w, b in zip(wq, bs)], dim=1)
        (gx,) = torch.autograd.grad(yq, xr, gy)
    ref = gx * torch.where(prev > 0, 1.0, 0.2) * math.sqrt(2.0)
    assert rel_l2(from_nhwc(dp.float().cpu(), B, C0, Hc, Wc), ref) < tol
    assert rel_l2(db.cpu(), ref.sum(dim=[0, 2, 3])) < (1e-4 if dtype == torch.float32 else 2e-2)
    # weight gradient (scale 1: draw is pre-scaled)
    dw = torch.zeros(16, C0, nh, device=DEV)
    o.wgrad(1, True, B, Hc, Wc, C0, nh, xd, (Hc * Wc * C0, C0, 1), draw, (nh * HW, 1, HW), dw.data_ptr(), 1.0,
            g_dt=L.DG_F32)
    torch.cuda.synchronize()
    got = dw.cpu().view(4, 4, C0, nh).permute(2, 3, 0, 1)
    assert rel_l2(got, torch.cat(list(gws), dim=1)) < tol


@pytest.mark.parametrize("Hc,Wc,B", [(8, 64, 2), (32, 256, 2),  # 8 blocks: direct bias-gradient rows; 128: staged per slot
                                     (32, 512, 24)])            # the bench geometry: 4 tiles per wave (prefetch steady state)
@pytest.mark.parametrize("nh", [1, 2, 3])
def test_head_bwd_data_pixel_major_mfma(L, nh, Hc, Wc, B):
    """Head backward-data through the direct-fragment MFMA kernel (thin_s2_mfma, adjoint boundary incl. the
    reflect-adjoint extra taps at rows 1 and H-2) from the pixel-major bf16 copy of the head gradient; the bias-gradient
    rows added directly (few blocks) and staged through DgConv.dbias_ws, which every launch leaves zero."""
    from dusty_gan_amd.engine import Ops
    g = torch.Generator().manual_seed(40 + nh)
    C0 = 64
    dtype = torch.bfloat16
    x = torch.randn(B, C0, Hc, Wc, generator=g).requires_grad_()
    ws = [torch.randn(C0, 1, 4, 4, generator=g).bfloat16().float()] + \
         ([torch.randn(C0, nh - 1, 4, 4, generator=g).bfloat16().float()] if nh > 1 else [])
    y = torch.cat([O.head(x, w, None, True) * math.sqrt(w.shape[1] * 16) for w in ws], dim=1)  # unscaled linear map
    gy = torch.randn(y.shape, generator=g).bfloat16().float()
    (gx,) = torch.autograd.grad(y, x, gy)
    prev = torch.randn(x.shape, generator=g)
    ref = gx * torch.where(prev > 0, 1.0, 0.2) * math.sqrt(2.0)
    o = Ops(dtype)
    o.force = 3
    HW = 4 * Hc * Wc
    cp = 2 if nh <= 2 else 4                     # channel padding of the pixel-major gradient
    draw_pm = torch.zeros(B, 2 * Hc, 2 * Wc, cp)
    draw_pm[..., :nh] = gy.permute(0, 2, 3, 1)
    draw_pm = draw_pm.to(DEV, dtype).contiguous()
    shadow = torch.cat(ws, dim=1).permute(2, 3, 0, 1).contiguous().to(DEV, dtype)  # [tap][n = ci][k = co]
    prevd = nhwc(prev).to(DEV, dtype)
    dp = torch.empty(B * Hc * Wc * C0, device=DEV, dtype=dtype)
    db = torch.zeros(C0, device=DEV)
    o.conv(L.MODE_S2, 1, True, B, Hc, Wc, nh, C0, draw_pm, (HW * cp, cp, 1), dp, (Hc * Wc * C0, C0, 1),
           shadow.data_ptr(), 1.0, L.EPI_MASK, aux=prevd, dbias=db.data_ptr(), bias_mod=C0)
    torch.cuda.synchronize()
    assert rel_l2(from_nhwc(dp.float().cpu(), B, C0, Hc, Wc), ref) < 1e-2
    assert rel_l2(db.cpu(), ref.sum(dim=[0, 2, 3])) < 2e-2
    if dtype == torch.bfloat16:   # the same launch from the saved 1-bit mask of `prev` (DgConv.mask_in): identical bytes
        from tests.test_gpu_ops import pack_bits
        prevd._dg_bits = pack_bits(prevd)
        dp_b, db_b = torch.empty_like(dp), torch.zeros_like(db)
        o.conv(L.MODE_S2, 1, True, B, Hc, Wc, nh, C0, draw_pm, (HW * cp, cp, 1), dp_b, (Hc * Wc * C0, C0, 1),
               shadow.data_ptr(), 1.0, L.EPI_MASK, aux=prevd, dbias=db_b.data_ptr(), bias_mod=C0)
        torch.cuda.synchronize()
        del prevd._dg_bits
        assert torch.equal(dp.view(torch.int16), dp_b.view(torch.int16))
        assert rel_l2(db.cpu(), db_b.cpu()) < 1e-4   # (atomic adds: the order of the block sums differs run to run)
    scratch = Ops._dbias_ws[str(draw_pm.device)]
    assert float(scratch.abs().max()) == 0.0
    # a second launch adds onto db again (accumulating entry point): twice the sums, scratch zero again
    o.conv(L.MODE_S2, 1, True, B, Hc, Wc, nh, C0, draw_pm, (HW * cp, cp, 1), dp, (Hc * Wc * C0, C0, 1),
           shadow.data_ptr(), 1.0, L.EPI_MASK, aux=prevd, dbias=db.data_ptr(), bias_mod=C0)
    torch.cuda.synchronize()
    assert rel_l2(db.cpu(), 2 * ref.sum(dim=[0, 2, 3])) < 2e-2 and float(scratch.abs().max()) == 0.0


@pytest.mark.parametrize("nh,Hc,Wc,B", [(1, 8, 64, 2), (2, 4, 128, 3), (2, 2, 64, 2), (3, 4, 64, 2),
                                         (3, 32, 512, 2),   # the dusty2 head of the benchmark: both channel pairs, one pass
                                         (3, 2, 128, 3), (3, 2, 2048, 1),   # reflected rows only; too wide for one pass: two
                                         (2, 2, 1024, 1)])  # 2048-wide images: > 64 KiB of dynamic LDS
def test_head_wgrad_pixel_major_mfma(L, nh, Hc, Wc, B):
    """Head weight gradient through thin_wgrad_up_mfma (input-pixel-indexed im2col of the pixel-major bf16 head
    gradient, incl. the mirror terms of the two reflected rows) against autograd of the reference op, with per-sample
    weights."""
    from dusty_gan_amd.engine import Ops
    g = torch.Generator().manual_seed(70 + nh + Hc)
    C0 = 64
    dtype = torch.bfloat16
    x = torch.randn(B, C0, Hc, Wc, generator=g).bfloat16().float()
    ws = [torch.randn(C0, 1, 4, 4, generator=g).requires_grad_()] + \
         ([torch.randn(C0, nh - 1, 4, 4, generator=g).requires_grad_()] if nh > 1 else [])
    y = torch.cat([O.head(x, w, None, True) * math.sqrt(w.shape[1] * 16) for w in ws], dim=1)  # unscaled linear map
    gy = torch.randn(y.shape, generator=g).bfloat16().float()
    rs = torch.rand(B, generator=g) + 0.5
    gws = torch.autograd.grad(y, ws, gy * rs.view(B, 1, 1, 1))
    ref = torch.cat(list(gws), dim=1)  # [ci][co][ky][kx]
    o = Ops(dtype)
    o.force = 3
    HW = 4 * Hc * Wc
    cp = 2 if nh <= 2 else 4
    draw_pm = torch.full((B, 2 * Hc, 2 * Wc, cp), 7.0)  # unused padding channels must not leak into the real ones
    draw_pm[..., :nh] = gy.permute(0, 2, 3, 1)
    draw_pm = draw_pm.to(DEV, dtype).contiguous()
    xd = nhwc(x).to(DEV, dtype)
    dw = torch.zeros(16, C0, nh, device=DEV)
    from dusty_gan_amd import engine as E
    E.TRACE = []
    o.wgrad(1, True, B, Hc, Wc, C0, nh, xd, (Hc * Wc * C0, C0, 1), draw_pm, (HW * cp, cp, 1), dw.data_ptr(), 1.0,
            rowscale=rs.to(DEV))
    torch.cuda.synchronize()
    trace, E.TRACE = E.TRACE, None
    got = dw.cpu().view(4, 4, C0, nh).permute(2, 3, 0, 1)
    assert rel_l2(got, ref) < 1e-2
    # the launch above stored one partial tile per block in the workspace (summed by dg_wgrad_reduce's wide form when there
    # are more than 64 of them) unless the map is too wide for a single pass; with fp32 atomics onto dW instead: the same
    one_pass = not (nh == 3 and Wc == 2048)
    assert trace[0][1] == 7 and trace[0][5] == one_pass and (trace[0][3] == B * Hc // 2 if one_pass else True), trace
    o.use_ws = False
    dw2 = torch.zeros(16, C0, nh, device=DEV)
    o.wgrad(1, True, B, Hc, Wc, C0, nh, xd, (Hc * Wc * C0, C0, 1), draw_pm, (HW * cp, cp, 1), dw2.data_ptr(), 1.0,
            rowscale=rs.to(DEV))
    torch.cuda.synchronize()
    assert rel_l2(dw2.cpu(), dw.cpu()) < 1e-5
    # accumulate = 0 overwrites whatever dW held (the reduce launch does it in the workspace form)
    o.use_ws = True
    dw3 = torch.full((16, C0, nh), 5.0, device=DEV)
    o.wgrad(1, True, B, Hc, Wc, C0, nh, xd, (Hc * Wc * C0, C0, 1), draw_pm, (HW * cp, cp, 1), dw3.data_ptr(), 1.0,
            rowscale=rs.to(DEV), accumulate=0)
    torch.cuda.synchronize()
    assert rel_l2(dw3.cpu(), dw.cpu()) < 1e-5


@pytest.mark.parametrize("n,Hc", [(2, 8), (12, 32)])       # 24 partial tiles per launch; 576: dg_wgrad_reduce's wide form
def test_down1_wgrad_one_launch_over_real_fake_tangent(L, n, Hc):
    """thin_wgrad_down_mfma with the gradient-sample map (DgWgrad.g_mod): one launch over 3n input samples real | fake |
    tangent against the 2n-sample gradient chain (g sample = b % 2n, per-sample weights) == the two launches it replaces in
    the D phase (ordinary + R1 weight gradient of Down1, trainers/dcgan_amp.py:229-235)"""
    from dusty_gan_amd import engine as E
    g = torch.Generator().manual_seed(3)
    Wc, Ci, Co = 64, 2, 64
    a = torch.randn(3 * n * 4 * Hc * Wc * Ci, generator=g).to(DEV, torch.bfloat16)
    e = torch.randn(2 * n * Hc * Wc * Co, generator=g).to(DEV, torch.bfloat16)
    rs = (torch.rand(3 * n, generator=g) + 0.5).to(DEV)
    rs[2 * n:] = 1.0
    sa, sg = (4 * Hc * Wc * Ci, Ci, 1), (Hc * Wc * Co, Co, 1)
    o = E.Ops(torch.bfloat16)
    E.TRACE = []
    try:
        two = torch.zeros(16, Ci, Co, device=DEV)
        o.wgrad(0, True, 2 * n, Hc, Wc, Ci, Co, a, sa, e, sg, two.data_ptr(), 0.1, rowscale=rs[:2 * n].contiguous())
        o.wgrad(0, True, n, Hc, Wc, Ci, Co, a, sa, e, sg, two.data_ptr(), 0.1, a_off=2 * n * sa[0])
        assert o.wgrad_takes_map(0, True, 3 * n, Hc, Wc, Ci, Co, a, sa, e, sg, two.data_ptr())
        one = torch.zeros(16, Ci, Co, device=DEV)
        o.wgrad(0, True, 3 * n, Hc, Wc, Ci, Co, a, sa, e, sg, one.data_ptr(), 0.1, rowscale=rs, g_mod=2 * n)
        assert all(t[1] == 7 and t[5] for t in E.TRACE if t[0] == "wgrad"), E.TRACE   # the thin matrix-core kernel, partial
        assert E.TRACE[-1][3] == 3 * n * Hc // 2                                       # tiles in the workspace: one per block
        o.use_ws = False                                                               # fp32 atomics onto dW: the same sums
        atom = torch.zeros(16, Ci, Co, device=DEV)
        o.wgrad(0, True, 3 * n, Hc, Wc, Ci, Co, a, sa, e, sg, atom.data_ptr(), 0.1, rowscale=rs, g_mod=2 * n)
        o.use_ws = True
    finally:
        E.TRACE = None
    torch.cuda.synchronize()
    assert rel_l2(one.cpu(), two.cpu()) < 1e-5 and rel_l2(atom.cpu(), one.cpu()) < 1e-5
    ref = torch.zeros(16, Ci, Co)                  # and against the definition, sample by sample on the host
    a_h = a.float().cpu().view(3 * n, 2 * Hc, 2 * Wc, Ci)
    e_h = e.float().cpu().view(2 * n, Hc, Wc, Co)
    ap = O.pad_ring(a_h.permute(0, 3, 1, 2), True)                                    # [3n, Ci, 2Hc + 2, 2Wc + 2]
    for ky in range(4):
        for kx in range(4):
            win = ap[:, :, ky:ky + 2 * Hc:2, kx:kx + 2 * Wc:2]                         # [3n, Ci, Hc, Wc]
            for b in range(3 * n):
                ref[ky * 4 + kx] += 0.1 * float(rs[b]) * torch.einsum("chw,hwo->co", win[b], e_h[b % (2 * n)])
    assert rel_l2(one.cpu(), ref) < 1e-2
    o.force = 1                                    # the direct kernel has no map: refused, not ignored
    with pytest.raises(L.DgError):
        o.wgrad(0, True, 3 * n, Hc, Wc, Ci, Co, a, sa, e, sg, one.data_ptr(), 0.1, g_mod=2 * n)


@pytest.mark.parametrize("case", ["head1", "head2", "head3", "down1"])
@pytest.mark.parametrize("Hc", [2, 3, 8])
def test_thin_up_fragments_kept_with_the_shadows(L, case, Hc):
    """DgConv.up_frag: the thin matrix-core MODE_UP kernel (Head forward, Down1 backward-data) with weight fragments built
    from the fp32 MASTER by the shadow-refresh launch (dg_transpose_shadow_multi_frags) gives bit for bit what the same
    kernel gives with the fragments its own preparation launch builds from the bf16 shadow; the ParamStore keeps them
    current across weight updates."""
    import ctypes as C
    from dusty_gan_amd.engine import Ops
    lib = L.lib()
    g = torch.Generator().manual_seed(Hc)
    B, Wc, K = 2, 64, 64
    N = {"head1": 1, "head2": 2, "head3": 3, "down1": 2}[case]
    adj = 1 if case == "down1" else 0
    # fp32 master [tap][ci][co]: head ci = 64, co = N (n = co, k = ci); down1 ci = N, co = 64 (n = ci, k = co)
    ci, co = (N, K) if adj else (K, N)
    master = torch.randn(16, ci, co, generator=g).to(DEV)
    coci = torch.empty(16 * ci * co, dtype=torch.bfloat16, device=DEV)
    desc = torch.tensor([0, coci.data_ptr(), ci, co, 0], dtype=torch.int64).to(DEV)
    tiles = 16 * ((ci + 31) // 32) * ((co + 31) // 32)
    frag = torch.zeros(L.UP_FRAG_BYTES, dtype=torch.uint8, device=DEV)
    d = L.DgUpFrag()
    d.off, d.frag = 0, frag.data_ptr()
    d.m_st, d.m_sn, d.m_sk = (ci * co, co, 1) if adj else (ci * co, 1, co)
    d.N, d.Hc, d.adj = N, Hc, adj
    L.check(lib.dg_transpose_shadow_multi_frags(master.data_ptr(), desc.data_ptr(), 1, tiles, L.DG_BF16,
                                                (L.DgUpFrag * 1)(d), 1, None))
    assert torch.equal(coci.view(16, co, ci), master.bfloat16().permute(0, 2, 1).contiguous())   # the transposes still run
    w = master.bfloat16() if adj else coci                                                       # [tap][n][k] shadow
    x = torch.randn(B, Hc, Wc, K, generator=g).to(DEV, torch.bfloat16)
    o = Ops(torch.bfloat16)
    o.force = 3
    outs = []
    from dusty_gan_amd import engine as E
    E.TRACE = []
    for use in (False, True):
        out = torch.full((B, 2 * Hc, 2 * Wc, N), 7.0, device=DEV, dtype=torch.bfloat16)
        o.conv(L.MODE_UP, adj, True, B, Hc, Wc, K, N, x, (Hc * Wc * K, K, 1), out, (4 * Hc * Wc * N, N, 1), w.data_ptr(),
               0.125, L.EPI_LINEAR, up_frag=frag.data_ptr() if use else None)
        torch.cuda.synchronize()
        outs.append(out.clone())
    trace, E.TRACE = E.TRACE, None
    assert all(t[1] == 3 and t[8] == (2 if Hc >= 2 else 0) for t in trace), trace     # thin_up_mfma (needs two rows)
    assert float(outs[0].float().abs().mean()) > 0.01
    assert torch.equal(outs[0], outs[1])
    # argument checks
    assert lib.dg_transpose_shadow_multi_frags(master.data_ptr(), desc.data_ptr(), 1, tiles, L.DG_F32,
                                               (L.DgUpFrag * 1)(d), 1, None) == L.DG_EUNSUPPORTED
    assert lib.dg_transpose_shadow_multi_frags(master.data_ptr(), desc.data_ptr(), 1, tiles, L.DG_BF16,
                                               (L.DgUpFrag * 1)(d), 5, None) == L.DG_EINVAL
    d.N = 5
    assert lib.dg_transpose_shadow_multi_frags(master.data_ptr(), desc.data_ptr(), 1, tiles, L.DG_BF16,
                                               (L.DgUpFrag * 1)(d), 1, None) == L.DG_EINVAL


@pytest.mark.parametrize("Hc,Wc,B", [(8, 64, 3), (32, 512, 2), (20, 128, 2)])
def test_depth_head_applies_tanh_and_stores_the_image_sums(L, Hc, Wc, B):
    """DgConv.tanh_sum_parts (round 6): the thin matrix-core MODE_UP kernel as the baseline generator's depth head writes
    tanh(out) (Generator.forward, models/gans/dcgan_eqlr.py:69-72) and every workgroup stores the sum of what it wrote - against
    the same launch without the field followed by tanh: the image within fp32 rounding of the 17-instruction tanh (1.3e-7
    absolute), the per-sample sums of DgConvPlan.sum_parts partials in index order equal to the image's sums; two launches
    agree bit for bit; a two-channel head (N = 2) does not take the field (sum_parts = 0, plain output)."""
    from dusty_gan_amd.engine import Ops
    g = torch.Generator().manual_seed(Hc + Wc)
    K = 64
    o = Ops(torch.bfloat16)
    o.force = 3
    x = torch.randn(B, Hc, Wc, K, generator=g).to(DEV, torch.bfloat16)
    for N in (1, 2):
        w = (torch.randn(16, N, K, generator=g) * 0.5).to(DEV, torch.bfloat16)
        bias = torch.randn(N, generator=g).to(DEV)
        nscale = torch.full((N,), 0.25, device=DEV)
        HW = 4 * Hc * Wc

        def run(parts_buf):
            out = torch.full((B, N, 2 * Hc, 2 * Wc), 7.0, device=DEV)
            n = o.conv(L.MODE_UP, 0, True, B, Hc, Wc, K, N, x, (Hc * Wc * K, K, 1), out, (N * HW, 1, HW), w.data_ptr(), 1.0,
                       L.EPI_LINEAR, bias=bias.data_ptr(), bias_mod=N, out_dt=L.DG_F32, nscale=nscale, tanh_sums=parts_buf)
            torch.cuda.synchronize()
            return out, n
        plain, n0 = run(None)
        assert n0 == 0 and float(plain.abs().mean()) > 0.05
        buf = torch.full((B * 256,), float("nan"), device=DEV)
        got, parts = run(buf)
        if N == 2:
            assert parts == 0 and torch.equal(got, plain) and bool(torch.isnan(buf).all())
            continue
        assert parts == (Wc // 64) * ((Hc + 7) // 8)
        want = torch.tanh(plain.double())
        assert float((got.double() - want).abs().max()) < 4e-7
        sums = torch.zeros(B, device=DEV)
        for j in range(parts):                       # the reader's order of additions (dg_diffaug_blur_fwd)
            sums = sums + buf[:B * parts].view(B, parts)[:, j]
        ref = got.double().sum(dim=[1, 2, 3])
        assert float((sums.double() - ref).abs().max()) <= 1e-5 * max(1.0, float(ref.abs().max()))
        assert bool(torch.isnan(buf[B * parts:]).all())
        buf2 = torch.zeros(B * 256, device=DEV)
        got2, _ = run(buf2)
        assert torch.equal(got, got2) and torch.equal(buf[:B * parts], buf2[:B * parts])


def test_param_store_rebuilds_up_fragments_with_every_refresh(L):
    """engine.ParamStore.up_frag: registered once, rebuilt by every refresh_transposed (the launch behind each optimizer
    step) - after a weight change the fragments equal those of a freshly registered store."""
    from dusty_gan_amd import engine as E
    segs = E.d_segments(1, [64, 128, 256, 512], (64, 256))     # Down1: 2 (BlurVH) -> 64 channels
    st = E.ParamStore(segs)
    st.apply(lambda t: t.to(DEV))
    st.flat.normal_()
    st.refresh_shadows(torch.bfloat16)
    p = st.up_frag("d1_w", (2 * 64, 64, 1), 2, 32, 1)
    assert p is not None and p == st.up_frag("d1_w", (2 * 64, 64, 1), 2, 32, 1)     # registered once
    before = st.up_frags[("d1_w", 32, 1)][0].clone()
    st.flat.mul_(-0.5)                                                               # "an optimizer step"
    st.refresh_shadows(torch.bfloat16)
    after = st.up_frags[("d1_w", 32, 1)][0].clone()
    assert not torch.equal(before, after)
    st2 = E.ParamStore(segs)
    st2.apply(lambda t: t.to(DEV))
    st2.flat.copy_(st.flat)
    st2.refresh_shadows(torch.bfloat16)
    st2.up_frag("d1_w", (2 * 64, 64, 1), 2, 32, 1)
    assert torch.equal(after, st2.up_frags[("d1_w", 32, 1)][0])
    st3 = E.ParamStore(segs)
    st3.apply(lambda t: t.to(DEV))
    st3.refresh_shadows(torch.float32)
    assert st3.up_frag("d1_w", (2 * 64, 64, 1), 2, 32, 1) is None                    # fp32: the kernel prepares its own
