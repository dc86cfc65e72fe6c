"""DG_BF16X2 - the split-bf16 storage form of the fp32x3 precision mode (include/dusty_gan_hip.h) - through the C ABI:
every element is the pair hi = bf16(x), lo = bf16(x - hi), laid out per 64 channels as 128 bytes of hi and 128 bytes of lo,
and the bf16 matrix-core kernels contract x . w as x_hi w_hi + x_lo w_hi + x_hi w_lo (three K steps per real one).

Checked here against the CPU oracle at the FP32 tolerance (rel-L2 <= 1e-4; the dropped x_lo w_lo term is ~2^-16 of a
product): the layout itself (dg_cast / dg_uncast), the ping-pong conv (Down / Up forward, both backward-data passes with mask
source and bias-gradient rows; reference models/gans/dcgan_eqlr.py:19-26,75-82), the LDS-DMA weight-gradient kernel (single
launches, per-sample weights, the 3n-sample map, grouped launches), the thin VALU kernels at the two ends of the networks
(Head forward from a split-bf16 feature map, Down1 forward into one) and the direct kernels as a second opinion on the
same buffers.
"""
import contextlib
import math

import pytest
import torch

from oracle import dusty_oracle as O
from tests.golden_util import rel_l2
from tests.test_gpu_ops import from_nhwc, nhwc, pack_down, pack_up

pytestmark = pytest.mark.gpu

DEV = "cuda"
TOL = 1e-4


@pytest.fixture(scope="module")
def L():
    from dusty_gan_amd import _lib
    _lib.lib()
    return _lib


def to_x2(t):
    """fp32 tensor (channel-minor, channels a multiple of 64) -> a tagged DG_BF16X2 device tensor"""
    from dusty_gan_amd import engine as E
    return E.x2_pack(t.contiguous().view(-1).to(DEV, torch.float32))


def from_x2(t):
    from dusty_gan_amd import engine as E
    return E.x2_unpack(t)


def test_layout_and_round_trip(L):
    """hi at bf16 index 2 i - i % 64, lo 64 further; hi + lo keeps 16 mantissa bits of x"""
    g = torch.Generator().manual_seed(5)
    x = torch.randn(7 * 192, generator=g) * torch.logspace(-6, 6, 7 * 192)
    xd = to_x2(x)
    raw = xd.view(torch.bfloat16).cpu().float()          # the bytes as bf16
    i = torch.arange(x.numel())
    hi_idx = 2 * i - (i % 64)
    hi, lo = raw[hi_idx], raw[hi_idx + 64]
    assert torch.equal(hi, x.bfloat16().float())
    assert torch.equal(lo, (x - hi).bfloat16().float())
    back = from_x2(xd).cpu()
    assert torch.equal(back, hi + lo)
    assert float(((back - x).abs() / x.abs()).max()) < 2.0 ** -16


def weights_x2(E, w_nk):
    """fp32 [16][n][k] shadow + its registered split twin"""
    wd = w_nk.contiguous().to(DEV, torch.float32)
    tw = E.x2_pack(wd.view(-1))
    E.X2_TWIN[wd.data_ptr()] = tw.data_ptr()
    return wd, tw


def run_conv_x2(L, mode, adj, x, w_nk, N, scale, epi, force, bias=None, aux=None, want_db=False, rowscale=None):
    from dusty_gan_amd import engine as E
    o = E.Ops(torch.float32)
    o.force = force
    B, K, Hin, Win = x.shape
    if mode == L.MODE_S2:
        Hc, Wc, Ho, Wo = Hin // 2, Win // 2, Hin // 2, Win // 2
    else:
        Hc, Wc, Ho, Wo = Hin, Win, 2 * Hin, 2 * Win
    xd = to_x2(nhwc(x))
    wd, tw = weights_x2(E, w_nk)
    out = E.tag_x2(torch.full((B * Ho * Wo * N,), 3.0, device=DEV))
    auxd = None if aux is None else to_x2(nhwc(aux))
    biasd = None if bias is None else bias.to(DEV)
    db = torch.zeros(N, device=DEV) if want_db else None
    rs = None if rowscale is None else rowscale.to(DEV)
    E.TRACE = []
    try:
        if force == 1:   # the direct kernel reads the split weights through dg_ld like everything else
            o.conv(mode, adj, True, B, Hc, Wc, K, N, xd, (Hin * Win * K, K, 1), out, (Ho * Wo * N, N, 1), tw.data_ptr(), scale, epi,
                   bias=None if biasd is None else biasd.data_ptr(), bias_mod=N, aux=auxd,
                   dbias=None if db is None else db.data_ptr(), rowscale=rs, w_dt=L.DG_BF16X2, w_strides=(N * K, 1, K))
        else:
            o.conv(mode, adj, True, B, Hc, Wc, K, N, xd, (Hin * Win * K, K, 1), out, (Ho * Wo * N, N, 1), wd.data_ptr(), scale, epi,
                   bias=None if biasd is None else biasd.data_ptr(), bias_mod=N, aux=auxd,
                   dbias=None if db is None else db.data_ptr(), rowscale=rs)
        fam = [t for t in E.TRACE if t[0] == "conv"][0][1]
    finally:
        E.TRACE = None
        E.X2_TWIN.pop(wd.data_ptr(), None)
    torch.cuda.synchronize()
    assert fam == (1 if force == 1 else 5), fam
    res = from_nhwc(from_x2(out).cpu(), B, N, Ho, Wo)
    return (res, db.cpu()) if want_db else res


X2_CASES = [  # (Ci, Co, H, W, B): the ping-pong kernel's tile flavours (tests/test_gpu_ops.py CASES, force 5)
    (256, 128, 4, 128, 2),
    (128, 128, 2, 512, 1),
    (128, 256, 4, 64, 8),
    (64, 128, 2, 256, 1),    # backward-data into 64 channels: the both-parities tile
    (64, 128, 4, 64, 8),
    (512, 128, 2, 128, 2),   # four N tiles in the backward-data pass
]


@pytest.mark.parametrize("Ci,Co,H,W,B", X2_CASES)
def test_down_layer(L, Ci, Co, H, W, B):
    from dusty_gan_amd import engine as E
    g = torch.Generator().manual_seed(Ci * 1000 + Co + H)
    x = torch.randn(B, Ci, 2 * H, 2 * W, generator=g)
    w = torch.randn(Co, Ci, 4, 4, generator=g)
    b = torch.randn(Co, generator=g)
    xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    y = O.down(xr, wr, br, True)
    fwd, bwd = pack_down(w)
    s = 1.0 / math.sqrt(Ci * 16)
    out = run_conv_x2(L, L.MODE_S2, 0, x, fwd, Co, s, L.EPI_LRELU, 5, bias=b)
    assert rel_l2(out, y) < TOL
    gy = torch.randn(y.shape, generator=g)
    e = gy * torch.where(y > 0, 1.0, 0.2) * math.sqrt(2.0)
    gx, gw = torch.autograd.grad(y, [xr, wr], gy, retain_graph=True)
    prev = torch.randn(x.shape, generator=g)
    rs = torch.rand(B, generator=g) + 0.5
    dx, db = run_conv_x2(L, L.MODE_UP, 1, e, bwd, Ci, s, L.EPI_MASK, 5, aux=prev, want_db=True, rowscale=rs)
    ref_dx = gx * torch.where(prev > 0, 1.0, 0.2) * math.sqrt(2.0)
    assert rel_l2(dx, ref_dx) < TOL
    assert rel_l2(db, (ref_dx * rs.view(B, 1, 1, 1)).sum(dim=[0, 2, 3])) < TOL
    # ... and the direct kernel on the same split buffers (dg_ld / dg_st understand the form): the layout's second opinion
    if Ci * Co <= 128 * 128:
        out1 = run_conv_x2(L, L.MODE_S2, 0, x, fwd, Co, s, L.EPI_LRELU, 1, bias=b)
        assert rel_l2(out1, y) < TOL
    # weight gradient on the LDS-DMA kernel, with per-sample weights
    o = E.Ops(torch.float32)
    o.force = 2
    xd, ed = to_x2(nhwc(x)), to_x2(nhwc(e))
    for rsd, ref in ((None, gw), (rs, None)):
        dw = torch.zeros(16, Ci, Co, device=DEV)
        E.TRACE = []
        try:
            o.wgrad(0, True, B, H, W, Ci, Co, xd, (4 * H * W * Ci, Ci, 1), ed, (H * W * Co, Co, 1), dw.data_ptr(), s,
                    rowscale=None if rsd is None else rsd.to(DEV))
            var = [t for t in E.TRACE if t[0] == "wgrad"][0][1]
        finally:
            E.TRACE = None
        torch.cuda.synchronize()
        assert var == 5, var
        if ref is None:
            ew = e * rs.view(B, 1, 1, 1)
            ref = torch.autograd.grad(O.down(xr, wr, br, True), wr, ew / (torch.where(y > 0, 1.0, 0.2) * math.sqrt(2.0)))[0]
        assert rel_l2(dw.cpu().view(4, 4, Ci, Co).permute(3, 2, 0, 1), ref) < TOL


@pytest.mark.parametrize("Ci,Co,H,W,B", X2_CASES + [(128, 64, 4, 128, 2)])   # + Up forward into 64 channels (Up3)
def test_up_layer(L, Ci, Co, H, W, B):
    from dusty_gan_amd import engine as E
    g = torch.Generator().manual_seed(Ci * 77 + Co + W)
    x = torch.randn(B, Ci, H, W, generator=g)
    w = torch.randn(Ci, Co, 4, 4, generator=g)
    b = torch.randn(Co, generator=g)
    xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    y = O.up(xr, wr, br, True)
    fwd, bwd = pack_up(w)
    s = 1.0 / math.sqrt(Co * 16)
    out = run_conv_x2(L, L.MODE_UP, 0, x, fwd, Co, s, L.EPI_LRELU, 5, bias=b)
    assert rel_l2(out, y) < TOL
    gy = torch.randn(y.shape, generator=g)
    lr = torch.where(y > 0, 1.0, 0.2) * math.sqrt(2.0)
    e = gy * lr
    gx, gw = torch.autograd.grad(y, [xr, wr], gy)
    if Co >= 128:            # (the adjoint MODE_UP... the MODE_S2 adjoint pass contracts over Co: the kernel wants >= 128)
        prev = torch.randn(x.shape, generator=g)
        dx, db = run_conv_x2(L, L.MODE_S2, 1, e, bwd, Ci, s, L.EPI_MASK, 5, aux=prev, want_db=True)
        ref_dx = gx * torch.where(prev > 0, 1.0, 0.2) * math.sqrt(2.0)
        assert rel_l2(dx, ref_dx) < TOL
        assert rel_l2(db, ref_dx.sum(dim=[0, 2, 3])) < TOL
    o = E.Ops(torch.float32)
    o.force = 2
    xd, ed = to_x2(nhwc(x)), to_x2(nhwc(e))
    dw = torch.zeros(16, Ci, Co, device=DEV)
    o.wgrad(1, True, B, H, W, Ci, Co, xd, (H * W * Ci, Ci, 1), ed, (4 * H * W * Co, Co, 1), dw.data_ptr(), s)
    torch.cuda.synchronize()
    assert rel_l2(dw.cpu().view(4, 4, Ci, Co).permute(2, 3, 0, 1), gw) < TOL


def test_wgrad_group_and_sample_map(L):
    """three layers' split-bf16 weight gradients as ONE launch (dg_wgrad_group) = the single launches bit for bit; the 3n-sample
    map (g sample = b % 2n: trainers/dcgan_amp.py:229-235 in one launch) against the two launches it replaces"""
    from dusty_gan_amd import engine as E
    g = torch.Generator().manual_seed(99)
    n = 2
    shapes = [(64, 128, 8, 128), (128, 256, 4, 64), (256, 128, 2, 64)]   # (Ci, Co, H, W) of a Down chain
    ops = E.Ops(torch.float32)
    ops.force = 2
    data = []
    for Ci, Co, H, W in shapes:
        a = to_x2(torch.randn(3 * n * 4 * H * W * Ci, generator=g))
        e = to_x2(torch.randn(2 * n * H * W * Co, generator=g))
        rs = (torch.rand(3 * n, generator=g) + 0.5).to(DEV)
        data.append((Ci, Co, H, W, a, e, rs))

    def run(grouped):
        outs = []
        with (ops.grouped() if grouped else contextlib.nullcontext()):
            for Ci, Co, H, W, a, e, rs in data:
                dw = torch.zeros(16, Ci, Co, device=DEV)
                ops.wgrad(0, True, 3 * n, H, W, Ci, Co, a, (4 * H * W * Ci, Ci, 1), e, (H * W * Co, Co, 1), dw.data_ptr(), 0.1,
                          rowscale=rs, g_mod=2 * n, defer=True)
                outs.append(dw)
        E.WGRAD_WS.flush()
        torch.cuda.synchronize()
        return outs
    single, group = run(False), run(True)
    for s_, g_ in zip(single, group):
        assert torch.equal(s_, g_)
    # the map against two launches on fp32 copies of the same operands (exact fp32 kernels)
    Ci, Co, H, W, a, e, rs = data[0]
    af, ef = from_x2(a), from_x2(e)
    o32 = E.Ops(torch.float32)
    ref = torch.zeros(16, Ci, Co, device=DEV)
    o32.wgrad(0, True, 2 * n, H, W, Ci, Co, af, (4 * H * W * Ci, Ci, 1), ef, (H * W * Co, Co, 1), ref.data_ptr(), 0.1, rowscale=rs)
    o32.wgrad(0, True, n, H, W, Ci, Co, af, (4 * H * W * Ci, Ci, 1), ef, (H * W * Co, Co, 1), ref.data_ptr(), 0.1,
              rowscale=rs[2 * n:].contiguous(), a_off=2 * n * 4 * H * W * Ci)
    torch.cuda.synchronize()
    assert rel_l2(single[0].cpu(), ref.cpu()) < TOL


@pytest.mark.parametrize("nheads", [1, 2, 3])
def test_network_ends_on_the_thin_kernels(L, nheads):
    """Head forward / its weight gradient FROM a split-bf16 feature map (models/gans/dcgan_eqlr.py:29-46), Head backward-data and
    Down1 forward INTO one (dcgan_eqlr.py:90), Down1 backward-data and weight gradient from one: fp32 on the other side,
    fp32 weights - the VALU kernels of the fp32 mode with dg_ld / dg_st (thin_smalln: typed reads of the staged rows)."""
    from dusty_gan_amd import engine as E
    g = torch.Generator().manual_seed(17 + nheads)
    B, C, H, W = 2, 64, 8, 64
    o = E.Ops(torch.float32)
    # Head forward: K = 64 -> N = nheads, MODE_UP, linear + bias, planar fp32 output
    x = torch.randn(B, C, H, W, generator=g)
    w = torch.randn(C, nheads, 4, 4, generator=g)
    b = torch.randn(nheads, generator=g)
    fwd, bwd = pack_up(w)
    xd = to_x2(nhwc(x))
    wd = fwd.contiguous().to(DEV)
    HW = 4 * H * W
    out = torch.empty(B, nheads, 2 * H, 2 * W, device=DEV)
    E.TRACE = []
    try:
        o.conv(L.MODE_UP, 0, True, B, H, W, C, nheads, xd, (H * W * C, C, 1), out, (nheads * HW, 1, HW), wd.data_ptr(), 0.25,
               L.EPI_LINEAR, bias=b.to(DEV).data_ptr(), bias_mod=nheads, out_dt=L.DG_F32)
        fam = [t for t in E.TRACE if t[0] == "conv"][0][1]
    finally:
        E.TRACE = None
    assert fam == 3, fam
    # the same launch from the fp32 copy of the split input: the reference of this comparison (hi + lo is what the kernel sees)
    xf = from_x2(xd)
    ref = torch.empty_like(out)
    o.conv(L.MODE_UP, 0, True, B, H, W, C, nheads, xf, (H * W * C, C, 1), ref, (nheads * HW, 1, HW), wd.data_ptr(), 0.25,
           L.EPI_LINEAR, bias=b.to(DEV).data_ptr(), bias_mod=nheads, out_dt=L.DG_F32)
    torch.cuda.synchronize()
    assert rel_l2(out.cpu(), ref.cpu()) < TOL      # (the split input on thin_up_mfma<X2>: three bf16 products per product)
    # Head backward-data: K = nheads (fp32, planar) -> N = 64 split-bf16, EPI_MASK with a split mask source + bias-gradient sums
    gy = torch.randn(B, nheads, 2 * H, 2 * W, generator=g).to(DEV)
    auxf = torch.randn(B * H * W * C, generator=g)
    auxd = to_x2(auxf)
    wb = bwd.contiguous().to(DEV)   # [16][n = 64][k = nheads]
    res = {}
    for form in ("x2", "f32"):
        dst = E.tag_x2(torch.empty(B * H * W * C, device=DEV)) if form == "x2" else torch.empty(B * H * W * C, device=DEV)
        db = torch.zeros(C, device=DEV)
        o.conv(L.MODE_S2, 1, True, B, H, W, nheads, C, gy, (nheads * HW, 1, HW), dst, (H * W * C, C, 1), wb.data_ptr(), 0.25,
               L.EPI_MASK, aux=auxd if form == "x2" else from_x2(auxd), dbias=db.data_ptr(), bias_mod=C, in_dt=L.DG_F32)
        torch.cuda.synchronize()
        res[form] = ((from_x2(dst) if form == "x2" else dst).cpu(), db.cpu())
    assert float(((res["x2"][0] - res["f32"][0]).abs() / (res["f32"][0].abs() + 1e-20)).max()) < 2.0 ** -15
    assert rel_l2(res["x2"][1], res["f32"][1]) < 1e-5
    # Head weight gradient: a = split feature map, g = planar fp32
    dws = {}
    for form in ("x2", "f32"):
        dw = torch.zeros(16, C, nheads, device=DEV)
        o.wgrad(1, True, B, H, W, C, nheads, xd if form == "x2" else xf, (H * W * C, C, 1), gy, (nheads * HW, 1, HW),
                dw.data_ptr(), 1.0, g_dt=L.DG_F32)
        torch.cuda.synchronize()
        dws[form] = dw.cpu()
    assert rel_l2(dws["x2"], dws["f32"]) < 1e-6
    # Down1: 2 -> 64 forward into a split map; its backward-data and weight gradient from one
    img = torch.randn(B * 2 * H * 2 * W * 2, generator=g).to(DEV)          # [B][2H][2W][2] pixel-major
    w1 = torch.randn(16, 64, 2, generator=g).to(DEV)                          # [tap][n][k]
    b1 = torch.randn(64, generator=g).to(DEV)
    h = {}
    for form in ("x2", "f32"):
        dst = E.tag_x2(torch.empty(B * H * W * 64, device=DEV)) if form == "x2" else torch.empty(B * H * W * 64, device=DEV)
        o.conv(L.MODE_S2, 0, True, B, H, W, 2, 64, img, (4 * H * W * 2, 2, 1), dst, (H * W * 64, 64, 1), w1.data_ptr(), 0.2,
               L.EPI_LRELU, bias=b1.data_ptr(), bias_mod=64)
        torch.cuda.synchronize()
        h[form] = dst
    hx = from_x2(h["x2"]).cpu()
    assert float(((hx - h["f32"].cpu()).abs() / (h["f32"].cpu().abs() + 1e-20)).max()) < 2.0 ** -15
    w1b = torch.randn(16, 2, 64, generator=g).to(DEV)                         # backward: [tap][n = 2][k = 64]
    e1 = to_x2(torch.randn(B * H * W * 64, generator=g))
    outs = {}
    for form in ("x2", "f32"):
        dst = torch.empty(B * 4 * H * W * 2, device=DEV)
        o.conv(L.MODE_UP, 1, True, B, H, W, 64, 2, e1 if form == "x2" else from_x2(e1), (H * W * 64, 64, 1), dst,
               (4 * H * W * 2, 2, 1), w1b.data_ptr(), 0.2, L.EPI_LINEAR)
        dw = torch.zeros(16, 2, 64, device=DEV)
        o.wgrad(0, True, B, H, W, 2, 64, img, (4 * H * W * 2, 2, 1), e1 if form == "x2" else from_x2(e1), (H * W * 64, 64, 1),
                dw.data_ptr(), 0.2)
        torch.cuda.synchronize()
        outs[form] = (dst.cpu(), dw.cpu())
    assert rel_l2(outs["x2"][0], outs["f32"][0]) < TOL and rel_l2(outs["x2"][1], outs["f32"][1]) < 1e-6


def test_proj_gemms_go_through_fp32_copies(L):
    """Proj's GEMMs (forward into a split activation, tangent pass with a split mask source, weight gradient from a split gradient)
    run on the fp32 kernels behind fp32 copies of the split operands (Ops.conv / Ops.wgrad, MODE_GEMM / wmode 2)"""
    from dusty_gan_amd import engine as E
    g = torch.Generator().manual_seed(3)
    B, K, N, C = 4, 64, 4096, 64
    z = torch.randn(B, K, generator=g).to(DEV)
    w = torch.randn(N, K, generator=g).to(DEV)
    bias = torch.randn(C, generator=g).to(DEV)
    o = E.Ops(torch.float32)
    s = 1.0 / math.sqrt(N)
    out_f = torch.empty(B * N, device=DEV)
    out_x = E.tag_x2(torch.empty(B * N, device=DEV))
    for dst in (out_f, out_x):
        o.conv(L.MODE_GEMM, 0, 1, B, 1, 1, K, N, z, (K, 0, 1), dst, (N, 0, 1), w.data_ptr(), s, L.EPI_LRELU,
               bias=bias.data_ptr(), bias_mod=C)
    torch.cuda.synchronize()
    back = from_x2(out_x)
    assert float(((back - out_f).abs() / (out_f.abs() + 1e-20)).max()) < 2.0 ** -15
    # weight gradient dW[n][k] = s sum_b dp[b][n] z[b][k] from the split dp
    dw_f, dw_x = torch.zeros(N, K, device=DEV), torch.zeros(N, K, device=DEV)
    dpf = from_x2(out_x)
    o.wgrad(2, 1, 1, 1, B, N, K, dpf, (0, N, 1), z.view(-1), (0, K, 1), dw_f.data_ptr(), s, accumulate=0)
    o.wgrad(2, 1, 1, 1, B, N, K, out_x, (0, N, 1), z.view(-1), (0, K, 1), dw_x.data_ptr(), s, accumulate=0)
    torch.cuda.synchronize()
    assert torch.equal(dw_f, dw_x)


def test_whole_step_split_storage_equals_register_split(monkeypatch):
    """One whole training step (dusty2, 64x1024, 512 / 64..512 channels, R1 + DiffAugment + path-length regulariser: every pass
    of both engines incl. the tangent and forward-over-reverse walks, trainers/dcgan_amp.py:162-325) in the fp32x3 mode with
    the fat feature maps stored as split-bf16 pairs against the same mode with fp32 storage and the split made in registers
    (round 4's kernels), same parameters, batches and randomness - and both against the fp32 oracle at the fp32 mode's bounds.
    B = 4: the 64-column maps tile into the ping-pong conv's 256-row tiles (four samples' segments); the path-length walk
    runs at B / 2 = 2, where those layers fall back to the direct kernels on the same split buffers."""
    from dusty_gan_amd.trainers.dcgan_amp import Trainer
    from tests.test_gpu_step import FP32_GRAD_COS, FP32_GRAD_TOL, _cos, run_both
    monkeypatch.setenv("DUSTY_GAN_FP32_SPLIT", "1")
    out = {}
    for pairs in (True, False):
        monkeypatch.setattr(Trainer, "fp32_pairs_default", pairs)
        tr, state, res = run_both("dusty2", (64, 1024), 512, 64, 512, 4, amp=False, pl=2.0, sync=True)
        assert tr.fp32_split and tr.fp32_pairs == pairs and tr.D.engine().x2 == pairs and tr._g_engines()[0].x2 == pairs
        out[pairs] = res[0]
        del tr
    (ref_a, ex_a, synth_a, gD_a, gG_a, scal_a), (_, _, synth_b, gD_b, gG_b, scal_b) = out[True], out[False]
    for va, vb in zip(scal_a, scal_b):
        assert abs(va - vb) <= 1e-4 * max(1.0, abs(vb)), (scal_a, scal_b)
    for k in synth_a:
        if k != "mask":
            assert rel_l2(synth_a[k], synth_b[k]) < 1e-4, k
    for ga, gb, tag in ((gD_a, gD_b, "D"), (gG_a, gG_b, "G")):
        for k in ga:
            if ga[k].numel() <= 4:     # (one-element head bias gradients: sums that cancel to 1e-3 of their absolute sum)
                assert rel_l2(ga[k], gb[k]) < 6e-2, (tag, k)
                continue
            assert rel_l2(ga[k], gb[k]) < FP32_GRAD_TOL and _cos(ga[k], gb[k]) > FP32_GRAD_COS, (tag, k, rel_l2(ga[k], gb[k]))
    # ... and the split-storage step against the oracle's gradients
    for k, v in gD_a.items():
        assert rel_l2(v, ex_a["grad_D"][k]) < FP32_GRAD_TOL, ("oracle D", k, rel_l2(v, ex_a["grad_D"][k]))


def test_split_storage_training_tracks_fp32_over_30_steps(monkeypatch):
    """The fp32x3 mode's TRAINING with split storage, through `Trainer.step` (the captured hipGraph from the third step on, the
    weight-shadow twins rebuilt behind every optimizer step, device RNG): 30 steps of a 64x1024 net (128 latent, 64..256
    channels, dusty2, R1 + DiffAugment, B = 8) against the exact fp32 mode from the same seeds.  GAN training amplifies the 2^-16
    operand rounding chaotically: Adam's first steps move every parameter by about +-lr whatever its gradient's size, so a gradient
    entry near zero that lands on the other side flips a whole update (scripts/probes/x2_step_gap.py, three seeds, worst logged
    scalar against max(1, |fp32|): step 1 2e-6..3e-5, step 2 1.3e-3..1.4e-2, step 3 5.7e-3..1.1e-2, step 4 7e-3..1.4e-2, step 5
    6e-3..2.2e-2, step 8 5e-3..4.6e-2).  So the first step is held to 1e-4 and steps 2-5 - the second eager step, the capture
    and the first two replays - to 5e-2 on every logged scalar, which is what the bug below missed by a factor of five; after
    that the runs are compared as curves, like the bf16 mode's 50-step test (tests/test_gpu_large_batch.py): window means
    within 0.1 (measured 0.004-0.054 over several boxes at this batch of 8).
    (This test found the one bug of the form's bring-up that no single-step test could: the twins were re-allocated on every
    refresh, harmless in eager steps, but inside the capture the D phase had already been recorded with the old twins'
    addresses - replays trained the discriminator's fat layers on stale weights: 0.25 off at the second replay.)"""
    from tests.test_gpu_step import make_trainer

    def run(x3):
        monkeypatch.setenv("DUSTY_GAN_FP32_SPLIT", "1" if x3 else "0")
        torch.manual_seed(31)
        tr = make_trainer("dusty2", True, (64, 1024), 128, 64, 256, 8, amp=False)
        assert tr.fp32_pairs == x3 and tr.D.engine().x2 == x3
        out = [dict(tr.step(i).items()) for i in range(30)]
        assert not x3 or "hipGraph" in tr.launch_mode()
        return out
    a, b = run(False), run(True)
    for i in range(5):
        for k in a[0]:
            tol = 1e-4 if i == 0 else 5e-2
            assert abs(a[i][k] - b[i][k]) <= tol * max(1.0, abs(a[i][k])), (i, k, a[i][k], b[i][k])
    worst = {}
    for lo, hi in ((0, 10), (10, 20), (20, 30)):
        for k in a[0]:
            ma = sum(s[k] for s in a[lo:hi]) / (hi - lo)
            mb_ = sum(s[k] for s in b[lo:hi]) / (hi - lo)
            assert ma == ma and mb_ == mb_, (k, lo, hi)
            dev = abs(ma - mb_) / max(1.0, abs(ma))
            worst[k] = max(worst.get(k, 0.0), dev)
            assert dev < 0.1, (k, lo, hi, ma, mb_)
    print("fp32x3 (split storage) vs fp32 over 30 steps, worst window deviation per scalar:", {k: round(v, 5) for k, v in worst.items()})
