"""GPU parity of the individual HIP kernels (through the C ABI) against the CPU oracle, same seeded inputs.

fp32 kernels: rel-L2 <= 1e-4 (expected ~1e-6); bf16 MFMA kernels: rel-L2 <= 2e-2 (bf16 inputs, fp32 accumulate:
the tolerance the reference's own AMP path needs).
"""
import ctypes as C
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import dusty_oracle as O
from tests.golden_util import load, rel_l2, sub

pytestmark = pytest.mark.gpu

DEV = "cuda"
TOL32 = 1e-4
TOLBF = 2e-2


@pytest.fixture(scope="module")
def L():
    from dusty_gan_amd import _lib
    _lib.lib()
    return _lib


def nhwc(x):  # [B,C,H,W] -> flat pixel-major/channel-minor
    return x.permute(0, 2, 3, 1).contiguous()


def from_nhwc(flat, B, C_, H, W):
    return flat.view(B, H, W, C_).permute(0, 3, 1, 2).contiguous()


def pack_bits(t):
    """1 bit per element, bit e % 8 of byte e / 8 = (element e > 0): the layout of DgConv.mask_out / mask_in"""
    b = (t.float().reshape(-1, 8) > 0).to(torch.int32)
    w = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.int32, device=t.device)
    return (b * w).sum(dim=1).to(torch.uint8)


def run_conv(L, mode, adj, ring, x, wpacked_nk, N, scale, epi, dtype, force, bias=None, aux=None, want_db=False,
             rowscale=None):
    """x [B,K,Hin,Win] torch cpu; wpacked_nk [16][N][K]; returns out [B,N,Ho,Wo] float cpu (+ dbias).
    bf16 feature maps with >= 16 channels also go through the saved 1-bit masks (DgConv.mask_out / mask_in), whatever kernel
    `force` puts them on: an EPI_LRELU launch must leave exactly the bits of (out > 0), and an EPI_MASK launch must give the
    SAME bytes from the bits of `aux` as from `aux` itself."""
    from dusty_gan_amd.engine import MaskBits, Ops
    o = Ops(dtype)
    o.force = force
    B, K, Hin, Win = x.shape
    if mode == L.MODE_S2:
        Hc, Wc, Ho, Wo = Hin // 2, Win // 2, Hin // 2, Win // 2
    else:
        Hc, Wc, Ho, Wo = Hin, Win, 2 * Hin, 2 * Win
    xd = nhwc(x).to(DEV, dtype)
    wd = wpacked_nk.contiguous().to(DEV, dtype)
    out = torch.empty(B * Ho * Wo * N, device=DEV, dtype=dtype)
    auxd = None if aux is None else nhwc(aux).to(DEV, dtype)
    biasd = None if bias is None else bias.to(DEV, torch.float32)
    db = torch.zeros(N, device=DEV) if want_db else None
    rs = None if rowscale is None else rowscale.to(DEV, torch.float32)
    use_bits = dtype == torch.bfloat16 and N % 16 == 0 and MaskBits.enabled
    obits = None
    if use_bits and epi == L.EPI_LRELU:
        obits = MaskBits.register(out)
        obits.fill_(0xA5)
    if use_bits and epi == L.EPI_MASK:
        auxd._dg_bits = pack_bits(auxd)

    def launch(dst, dbp):
        o.conv(mode, adj, ring, B, Hc, Wc, K, N, xd, (Hin * Win * K, K, 1), dst, (Ho * Wo * N, N, 1), wd.data_ptr(), scale,
               epi, bias=None if biasd is None else biasd.data_ptr(), bias_mod=N, aux=auxd,
               dbias=None if dbp is None else dbp.data_ptr(), rowscale=rs)
        torch.cuda.synchronize()
    launch(out, db)
    if obits is not None:
        assert torch.equal(obits, pack_bits(out)), "mask_out differs from (out > 0)"
    if use_bits and epi == L.EPI_MASK:   # the same launch from the saved activation itself: bit-identical
        del auxd._dg_bits
        out2 = torch.empty_like(out)
        db2 = None if db is None else torch.zeros_like(db)
        launch(out2, db2)
        assert torch.equal(out.view(torch.int16), out2.view(torch.int16)), "mask_in and aux give different outputs"
        if db is not None:
            from dusty_gan_amd import engine as E
            if E.DETERMINISTIC and force in (5, 9):
                # round 5: the ping-pong conv's bias-gradient sums leave as per-workgroup rows (DgConv.dbias_part), summed in a
                # fixed order - two launches of the same data agree bit for bit
                assert torch.equal(db, db2), rel_l2(db.cpu(), db2.cpu())
            else:
                assert rel_l2(db.cpu(), db2.cpu()) < 1e-5   # (atomic adds: the order of the partial sums differs run to run)
    res = from_nhwc(out.float().cpu(), B, N, Ho, Wo)
    return (res, db.cpu()) if want_db else res


@pytest.mark.parametrize("which", ["down", "up"])
@pytest.mark.parametrize("Ci,Co,H,W,B", [(64, 128, 4, 128, 2), (128, 64, 8, 64, 2), (256, 128, 4, 64, 3)])
def test_fp32x3_matrix_core_kernels_match_the_fp32_reference(L, which, Ci, Co, H, W, B):
    """DG_FORCE_FP32X3 (Ops.x3): fp32 operands through split-bf16 matrix instructions (a = a_hi + a_lo in bf16, three of the four
    partial products, fp32 accumulation) on the one-tile-per-workgroup conv and the register-staged weight-gradient
    kernel - forward, backward-data and weight gradient of a Down / Up layer against the fp32 oracle at the FP32 tolerance
    (1e-4; the dropped a_lo b_lo term is ~2^-16 of a product)."""
    from dusty_gan_amd.engine import Ops
    init = Ops.__init__

    def init_x3(self, dtype, x3=False):
        init(self, dtype, x3=True)
    try:
        Ops.__init__ = init_x3      # every Ops the op tests build asks for the split form
        fn = test_down_fwd_bwd_wgrad if which == "down" else test_up_fwd_bwd_wgrad
        fn(L, Ci, Co, H, W, B, True, torch.float32, 2)
        # ... and the flag really selects other kernels: same launch with and without it differ in the last bits only
        o0, o1 = Ops(torch.float32), Ops(torch.float32)
        assert o0.x3 and o1._f == (o1.force | L.DG_FORCE_FP32X3)
    finally:
        Ops.__init__ = init


def pack_down(w):  # Conv2d weight (Co,Ci,4,4) -> fwd [16][n=co][k=ci], bwd [16][n=ci][k=co]
    fwd = w.permute(2, 3, 0, 1).reshape(16, w.shape[0], w.shape[1])
    bwd = w.permute(2, 3, 1, 0).reshape(16, w.shape[1], w.shape[0])
    return fwd, bwd


def pack_up(w):  # ConvTranspose2d weight (Ci,Co,4,4) -> fwd [16][co][ci], bwd [16][ci][co]
    fwd = w.permute(2, 3, 1, 0).reshape(16, w.shape[1], w.shape[0])
    bwd = w.permute(2, 3, 0, 1).reshape(16, w.shape[0], w.shape[1])
    return fwd, bwd


CASES = [  # (Ci, Co, H, W, B, ring, dtype, force)   force 1 = direct, 2 = MFMA, 4 / 5 = lock-step / ping-pong persistent kernel
    (6, 4, 8, 16, 2, True, torch.float32, 1),
    (6, 4, 8, 16, 2, False, torch.float32, 1),
    (5, 3, 4, 6, 3, True, torch.float32, 1),
    (64, 64, 8, 64, 2, True, torch.float32, 2),
    (64, 128, 4, 128, 2, True, torch.float32, 2),
    (128, 64, 8, 64, 1, True, torch.bfloat16, 2),
    (128, 128, 4, 128, 2, True, torch.bfloat16, 2),
    # large tiles: 256-row M tiles built from 4 / 2 / 1 samples' row segments, 256- and 128-channel N tiles
    (128, 256, 2, 64, 4, True, torch.float32, 4),
    (256, 128, 4, 128, 2, True, torch.bfloat16, 4),
    (128, 128, 2, 512, 1, True, torch.bfloat16, 4),
    (128, 256, 4, 64, 8, True, torch.bfloat16, 4),
    (64, 128, 2, 256, 1, True, torch.bfloat16, 4),   # backward-data: 256 x 64 tiles (8 waves of 32 x 64)
    # the ping-pong kernel (bf16; what the benchmark's fat layers run) on the same geometries
    (256, 128, 4, 128, 2, True, torch.bfloat16, 5),
    (128, 128, 2, 512, 1, True, torch.bfloat16, 5),
    (128, 256, 4, 64, 8, True, torch.bfloat16, 5),
    (64, 128, 2, 256, 1, True, torch.bfloat16, 5),   # Down backward-data: 64 channels, both-parities tile (512 px x 64)
    (64, 128, 4, 64, 8, True, torch.bfloat16, 5),    # ... 4 sample segments of 64 + 2 columns
    (64, 128, 2, 512, 1, True, torch.bfloat16, 5),   # ... two column tiles per row
    (64, 128, 2, 256, 1, True, torch.bfloat16, 9),   # force 9: the single-parity 256 x 64 tile those layers ran on before
    (512, 128, 2, 128, 2, True, torch.bfloat16, 5),  # 512 channels in the backward-data pass: four N tiles, bias-gradient rows
                                                     # of 512 floats per wave (round 5: they once overlapped the store strips)
]


@pytest.mark.parametrize("Ci,Co,H,W,B,ring,dtype,force", CASES)
def test_down_fwd_bwd_wgrad(L, Ci, Co, H, W, B, ring, dtype, force):
    """Down (dcgan_eqlr.py:75-82): forward, backward-data (+ bias grads, lrelu mask), weight gradient."""
    g = torch.Generator().manual_seed(Ci * 1000 + Co + H)
    tol = TOL32 if dtype == torch.float32 else TOLBF
    x = torch.randn(B, Ci, 2 * H, 2 * W, generator=g)
    w = torch.randn(Co, Ci, 4, 4, generator=g)
    b = torch.randn(Co, generator=g)
    if dtype == torch.bfloat16:
        x, w = x.bfloat16().float(), w.bfloat16().float()
    xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    y = O.down(xr, wr, br, ring)
    gy = torch.randn(y.shape, generator=g)
    fwd, bwd = pack_down(w)
    s = 1.0 / math.sqrt(Ci * 16)
    out = run_conv(L, L.MODE_S2, 0, ring, x, fwd, Co, s, L.EPI_LRELU, dtype, force, bias=b)
    assert rel_l2(out, y) < tol
    # backward: e = gy * lrelu'(y) * sqrt2 is the gradient w.r.t. the pre-activation
    e = gy * torch.where(y > 0, 1.0, 0.2) * math.sqrt(2.0)
    if dtype == torch.bfloat16:
        e = e.bfloat16().float()
    gx, gw, gb = torch.autograd.grad(y, [xr, wr, br], gy, retain_graph=True)
    # (oracle-internal sanity: the bias gradient autograd returns is the pixel sum of e)
    assert rel_l2(e.sum(dim=[0, 2, 3]), gb) < (1e-2 if dtype == torch.bfloat16 else 1e-4)
    # backward-data into a "previous layer" with its own activation mask (aux) and bias-gradient sums
    prev = torch.randn(x.shape, generator=g)
    rs = torch.rand(B, generator=g) + 0.5
    dx, db = run_conv(L, L.MODE_UP, 1, ring, e, bwd, Ci, s, L.EPI_MASK, dtype, force, aux=prev, want_db=True,
                      rowscale=rs)
    if dtype == torch.bfloat16:  # reference backward from the SAME rounded e
        gx = torch.autograd.grad(y, xr, e / (torch.where(y > 0, 1.0, 0.2) * math.sqrt(2.0)))[0]
    ref_dx = gx * torch.where(prev > 0, 1.0, 0.2) * math.sqrt(2.0)
    assert rel_l2(dx, ref_dx) < tol
    assert rel_l2(db, (ref_dx * rs.view(B, 1, 1, 1)).sum(dim=[0, 2, 3])) < (tol if dtype == torch.float32 else 5e-2)
    # weight gradient
    from dusty_gan_amd.engine import Ops
    o = Ops(dtype)
    o.force = 2 if force in (4, 5, 9) else force
    xd, ed = nhwc(x).to(DEV, dtype), nhwc(e).to(DEV, dtype)
    dw = torch.zeros(16, Ci, Co, device=DEV)
    o.wgrad(0, ring, B, H, W, Ci, Co, xd, (4 * H * W * Ci, Ci, 1), ed, (H * W * Co, Co, 1), dw.data_ptr(), s)
    torch.cuda.synchronize()
    if dtype == torch.bfloat16:
        gw = torch.autograd.grad(O.down(xr, wr, br, ring), wr, e / (torch.where(y > 0, 1.0, 0.2) * math.sqrt(2.0)))[0]
    got = dw.cpu().view(4, 4, Ci, Co).permute(3, 2, 0, 1)
    assert rel_l2(got, gw) < tol
    # per-sample weights (the real-batch reuse of the R1 chain)
    dw2 = torch.zeros(16, Ci, Co, device=DEV)
    o.wgrad(0, ring, B, H, W, Ci, Co, xd, (4 * H * W * Ci, Ci, 1), ed, (H * W * Co, Co, 1), dw2.data_ptr(), s,
            rowscale=rs.to(DEV))
    torch.cuda.synchronize()
    ew = e * rs.view(B, 1, 1, 1)
    gw2 = torch.autograd.grad(O.down(xr, wr, br, ring), wr, ew / (torch.where(y > 0, 1.0, 0.2) * math.sqrt(2.0)))[0]
    assert rel_l2(dw2.cpu().view(4, 4, Ci, Co).permute(3, 2, 0, 1), gw2) < tol


@pytest.mark.parametrize("Ci,Co,H,W,B,ring,dtype,force", CASES)
def test_up_fwd_bwd_wgrad(L, Ci, Co, H, W, B, ring, dtype, force):
    """Up (dcgan_eqlr.py:19-26): forward, backward-data, weight gradient."""
    g = torch.Generator().manual_seed(Ci * 77 + Co + W)
    tol = TOL32 if dtype == torch.float32 else TOLBF
    x = torch.randn(B, Ci, H, W, generator=g)
    w = torch.randn(Ci, Co, 4, 4, generator=g)
    b = torch.randn(Co, generator=g)
    if dtype == torch.bfloat16:
        x, w = x.bfloat16().float(), w.bfloat16().float()
    xr, wr, br = x.clone().requires_grad_(), w.clone().requires_grad_(), b.clone().requires_grad_()
    y = O.up(xr, wr, br, ring)
    fwd, bwd = pack_up(w)
    s = 1.0 / math.sqrt(Co * 16)
    out = run_conv(L, L.MODE_UP, 0, ring, x, fwd, Co, s, L.EPI_LRELU, dtype, force, bias=b)
    assert rel_l2(out, y) < tol
    gy = torch.randn(y.shape, generator=g)
    lr = torch.where(y > 0, 1.0, 0.2) * math.sqrt(2.0)
    e = gy * lr
    if dtype == torch.bfloat16:
        e = e.bfloat16().float()
    gx, gw = torch.autograd.grad(y, [xr, wr], e / lr)
    prev = torch.randn(x.shape, generator=g)
    dx, db = run_conv(L, L.MODE_S2, 1, ring, e, bwd, Ci, s, L.EPI_MASK, dtype, force, aux=prev, want_db=True)
    ref_dx = gx * torch.where(prev > 0, 1.0, 0.2) * math.sqrt(2.0)
    assert rel_l2(dx, ref_dx) < tol
    assert rel_l2(db, ref_dx.sum(dim=[0, 2, 3])) < (tol if dtype == torch.float32 else 5e-2)
    from dusty_gan_amd.engine import Ops
    o = Ops(dtype)
    o.force = 2 if force in (4, 5, 9) else force
    xd, ed = nhwc(x).to(DEV, dtype), nhwc(e).to(DEV, dtype)
    dw = torch.zeros(16, Ci, Co, device=DEV)
    o.wgrad(1, ring, B, H, W, Ci, Co, xd, (H * W * Ci, Ci, 1), ed, (4 * H * W * Co, Co, 1), dw.data_ptr(), s)
    torch.cuda.synchronize()
    assert rel_l2(dw.cpu().view(4, 4, Ci, Co).permute(2, 3, 0, 1), gw) < tol


@pytest.mark.parametrize("B,N,epi", [(32, 4096, "lrelu"), (17, 2048, "lrelu"), (8, 1024, "linear"), (32, 131072, "lrelu"),
                                     (8, 131072, "lrelu"), (16, 131072, "linear"), (4, 524288, "lrelu")])
def test_proj_forward_weight_streaming_kernel(L, B, N, epi):
    """Proj forward (dcgan_eqlr.py:6-16) on the weight-streaming kernel (proj_stream.hip: dg_conv force 10 / what force 0
    picks for bf16, K = 512, B <= 32) against the general MFMA kernel (force 2) and a float64 GEMM of the same bf16
    operands; ragged batch (17 of 32 MFMA columns), both epilogues, the benchmark's 131072-row shape at both register
    variants (B <= 16 / <= 32: eight ring turns per wave - a first version waited for one piece too few in the first turn
    and read one row in 8192 before it had landed) and the 128x2048 configuration's 524288 rows."""
    from dusty_gan_amd import engine as E
    K, C = 512, 64
    g = torch.Generator().manual_seed(B + N)
    z = torch.randn(B, K, generator=g).to(DEV, torch.bfloat16)
    w = torch.randn(N, K, generator=g).to(DEV, torch.bfloat16)
    bias = torch.randn(C, generator=g).to(DEV)
    s = 1.0 / math.sqrt(N)
    code = L.EPI_LRELU if epi == "lrelu" else L.EPI_LINEAR
    outs = {}
    for force in (10, 2, 0):
        o = E.Ops(torch.bfloat16)
        o.force = force
        out = torch.full((B, N), 7.0, device=DEV, dtype=torch.bfloat16)
        E.TRACE = []
        try:
            o.conv(L.MODE_GEMM, 0, 1, B, 1, 1, K, N, z, (K, 0, 1), out, (N, 0, 1), w.data_ptr(), s, code,
                   bias=bias.data_ptr(), bias_mod=C)
            fam = [t for t in E.TRACE if t[0] == "conv"][0][1]
        finally:
            E.TRACE = None
        torch.cuda.synchronize()
        assert fam == (2 if force == 2 else 6), (force, fam)
        outs[force] = out.float().cpu()
    ref = (z.double().cpu() @ w.double().cpu().t()) * s + bias.double().cpu().repeat(N // C)[None, :]
    if epi == "lrelu":
        ref = torch.where(ref > 0, ref, 0.2 * ref) * math.sqrt(2.0)
    assert bool(torch.isfinite(outs[10]).all())
    assert rel_l2(outs[10], ref) < TOLBF and rel_l2(outs[2], ref) < TOLBF
    assert rel_l2(outs[10], outs[2]) < 2e-3 and torch.equal(outs[0], outs[10])
    assert float((outs[10] - ref.float()).abs().max()) < 0.05 * float(ref.abs().max())   # no single stale row


@pytest.mark.parametrize("force", [5, 9], ids=["both-parities-tile", "single-parity-tile"])
@pytest.mark.parametrize("Ci,H,W,B", [(128, 4, 128, 2), (64, 2, 256, 1), (128, 4, 64, 4), (128, 2, 512, 1)])
def test_up_forward_64_channels_on_the_pingpong_kernel(L, Ci, H, W, B, force):
    """Up forward with 64 output channels (Up3: dcgan_eqlr.py:19-26 at 128 -> 64): MODE_UP, bias + leaky-relu epilogue, on
    the ping-pong kernel's both-parities tile (512 pixels x 64 channels, force 5) and on the single-parity 256 x 64 tile
    it replaces (force 9); 1 / 2 / 4 sample segments per tile and two column tiles per row.  (The Down / Up case list above
    cannot hold this shape: its backward-data pass has K = 64, which the adjoint MODE_UP flavour refuses.)"""
    from dusty_gan_amd import engine as E
    Co = 64
    g = torch.Generator().manual_seed(Ci + H + W)
    x = torch.randn(B, Ci, H, W, generator=g).bfloat16().float()
    w = torch.randn(Ci, Co, 4, 4, generator=g).bfloat16().float()
    b = torch.randn(Co, generator=g)
    y = O.up(x, w, b, True)
    fwd, _ = pack_up(w)
    E.TRACE = []
    try:
        out = run_conv(L, L.MODE_UP, 0, True, x, fwd, Co, 1.0 / math.sqrt(Co * 16), L.EPI_LRELU, torch.bfloat16, force, bias=b)
        tr = [t for t in E.TRACE if t[0] == "conv"][0]
    finally:
        E.TRACE = None
    assert tr[1] == 5 and (tr[2], tr[3]) == ((512, 64) if force == 5 else (256, 64)), tr
    assert rel_l2(out, y) < TOLBF


@pytest.mark.parametrize("wmode", [0, 1])
@pytest.mark.parametrize("force", [2, 7, 8])
def test_wgrad_group_is_the_single_launches_in_one_grid(L, wmode, force):
    """dg_wgrad_group (round 5): up to four layers' weight-gradient GEMMs as ONE launch - every tile shape of the LDS-DMA kernel
    (128 / 64 input and output channels), tap pairs forced / forbidden / chosen, per-sample weights and the 3n-sample map on
    one item, a fifth layer that overflows the group.  Same plan function, same partial tiles, same reduce: the gradients must
    equal those of the single launches BIT FOR BIT."""
    from dusty_gan_amd import engine as E
    g = torch.Generator().manual_seed(31 + wmode + force)
    n = 2
    fa, fg = (4, 1) if wmode == 0 else (1, 4)
    layers = [(128, 128, 4, 128), (64, 128, 8, 64), (128, 64, 4, 64), (64, 64, 2, 128), (128, 128, 2, 64)]
    data = []
    for Ci, Co, H, W in layers:
        a = torch.randn(3 * n * fa * H * W * Ci, generator=g).to(DEV, torch.bfloat16)
        e = torch.randn(2 * n * fg * H * W * Co, generator=g).to(DEV, torch.bfloat16)
        data.append((a, e, (fa * H * W * Ci, Ci, 1), (fg * H * W * Co, Co, 1)))
    rs = (torch.rand(3 * n, generator=g) + 0.5).to(DEV)

    def run(grouped, rounds=0):
        o = E.Ops(torch.bfloat16)
        o.force = force
        out = [torch.zeros(16, Ci, Co, device=DEV) for Ci, Co, _, _ in layers]
        E.TRACE = []
        prev_rounds, E.Ops.group_rounds = E.Ops.group_rounds, rounds
        try:
            prev, E.Ops.group_enabled = E.Ops.group_enabled, grouped
            with o.grouped():
                for k, ((Ci, Co, H, W), (a, e, sa, sg)) in enumerate(zip(layers, data)):
                    if k == 0:   # the discriminator's merged form: 3n input samples against 2n gradient samples, weighted
                        o.wgrad(wmode, True, 3 * n, H, W, Ci, Co, a, sa, e, sg, out[k].data_ptr(), 0.05, rowscale=rs, g_mod=2 * n,
                                defer=True)
                    else:
                        o.wgrad(wmode, True, 2 * n, H, W, Ci, Co, a, sa, e, sg, out[k].data_ptr(), 0.03 * k, defer=True)
            tr = E.TRACE
        finally:
            E.TRACE, E.Ops.group_enabled, E.Ops.group_rounds = None, prev, prev_rounds
        E.WGRAD_WS.flush()
        torch.cuda.synchronize()
        return out, tr
    want, tr0 = run(False)                                      # (also grows the split-K workspace to what the block needs)
    got, tr = run(True)
    groups = [t for t in tr if t[0] == "wgrad_group"]
    assert len(groups) == 1 and groups[0][1] == 4, tr          # four layers in the one launch, the fifth on its own
    assert not [t for t in tr0 if t[0] == "wgrad_group"]
    assert [t[1] for t in tr if t[0] == "wgrad"] == [5] * 5
    for k, (x, y) in enumerate(zip(got, want)):
        assert float(y.abs().max()) > 0
        assert torch.equal(x, y), (k, rel_l2(x.cpu(), y.cpu()))
    # rounds > 0: the group shares rounds x 512 workgroups among its layers - each layer's K split shrinks (fewer partial tiles),
    # so the sums are formed in another order: equal to rounding, with fewer splits than the single launches use
    for rounds in (1, 2):
        got_r, tr_r = run(True, rounds)
        sp0 = {t[2]: t[3] for t in tr if t[0] == "wgrad"}          # layer -> K splits
        sp_r = {t[2]: t[3] for t in tr_r if t[0] == "wgrad"}
        assert all(sp_r[k] <= sp0[k] for k in sp0) and sum(sp_r.values()) < sum(sp0.values()), (rounds, sp_r, sp0)
        for k, (x, y) in enumerate(zip(got_r, want)):
            assert rel_l2(x.cpu(), y.cpu()) < 1e-5, (rounds, k, rel_l2(x.cpu(), y.cpu()))
    # a workspace that has to GROW inside the block sums the pending partials before it moves: launches still queued in the
    # open group must be issued first (a first version reduced partials nobody had written yet)
    E.WGRAD_WS._by_stream.clear()
    got2, tr2 = run(True)
    assert E.WGRAD_WS.grows >= 2, E.WGRAD_WS.grows
    for k, (x, y) in enumerate(zip(got2, want)):
        assert torch.equal(x, y), ("grown inside the block", k, rel_l2(x.cpu(), y.cpu()))



@pytest.mark.parametrize("force", [2, 7, 8], ids=["auto", "tap-pairs", "single-taps"])
@pytest.mark.parametrize("wmode,Ci,Co,H,W", [(0, 128, 128, 4, 128), (1, 128, 64, 4, 64), (0, 64, 128, 8, 64)])
def test_wgrad_workspace_and_sample_map(L, wmode, Ci, Co, H, W, force):
    """The LDS-DMA weight-gradient kernel's split-K workspace form (DgWgrad.ws + dg_wgrad_reduce) and its gradient-sample
    index map (DgWgrad.g_mod) against the register-staged kernel with atomics (force 6, itself held to autograd above):
      (a) workspace partials, reduce adding onto a pre-filled dW == atomics onto the same dW;
      (b) deferred: two layers' partials summed by ONE reduce launch;
      (c) one launch over 3n input samples with g sample = b % 2n and per-sample weights == the two launches it replaces
          (the D phase's ordinary + R1 weight gradients, trainers/dcgan_amp.py:229-235).
    W-tap pairs forced / forbidden / chosen (force 7 / 8 / 2)."""
    from dusty_gan_amd import engine as E
    g = torch.Generator().manual_seed(wmode * 7 + Ci + W)
    n = 2
    fa, fg = (4, 1) if wmode == 0 else (1, 4)
    a = torch.randn(3 * n * fa * H * W * Ci, generator=g).to(DEV, torch.bfloat16)
    e = torch.randn(2 * n * fg * H * W * Co, generator=g).to(DEV, torch.bfloat16)
    rs = (torch.rand(3 * n, generator=g) + 0.5).to(DEV)
    rs[2 * n:] = 1.0
    sa, sg = (fa * H * W * Ci, Ci, 1), (fg * H * W * Co, Co, 1)
    base = torch.randn(16, Ci, Co, generator=g).to(DEV)
    ref_o, o = E.Ops(torch.bfloat16), E.Ops(torch.bfloat16)
    ref_o.force, ref_o.use_ws, o.force = 6, False, force

    def ref(B, a_off, rowscale):
        dw = base.clone()
        ref_o.wgrad(wmode, True, B, H, W, Ci, Co, a, sa, e, sg, dw.data_ptr(), 0.05, rowscale=rowscale, a_off=a_off)
        return dw
    want2 = ref(2 * n, 0, rs[:2 * n].contiguous())
    # (a) immediate reduce, accumulate onto a pre-filled dW
    E.TRACE = []
    try:
        dw = base.clone()
        o.wgrad(wmode, True, 2 * n, H, W, Ci, Co, a, sa, e, sg, dw.data_ptr(), 0.05, rowscale=rs[:2 * n].contiguous())
        tr = [t for t in E.TRACE if t[0] == "wgrad"][0]
    finally:
        E.TRACE = None
    assert tr[1] == 5 and tr[5] and tr[3] >= 1, tr                     # LDS-DMA kernel, workspace in use
    assert (tr[4] == 1) == (force == 7) or force == 2, tr               # tap pairs as forced
    torch.cuda.synchronize()
    assert rel_l2(dw.cpu() - base.cpu(), want2.cpu() - base.cpu()) < 1e-5
    # the same launch through atomics (no workspace) still works
    o_at = E.Ops(torch.bfloat16)
    o_at.force, o_at.use_ws = force, False
    dwa = base.clone()
    o_at.wgrad(wmode, True, 2 * n, H, W, Ci, Co, a, sa, e, sg, dwa.data_ptr(), 0.05, rowscale=rs[:2 * n].contiguous())
    torch.cuda.synchronize()
    assert rel_l2(dwa.cpu() - base.cpu(), want2.cpu() - base.cpu()) < 1e-5
    # (b) deferred: two launches into different gradients, one reduce
    d1, d2 = base.clone(), torch.zeros_like(base)
    o.wgrad(wmode, True, 2 * n, H, W, Ci, Co, a, sa, e, sg, d1.data_ptr(), 0.05, rowscale=rs[:2 * n].contiguous(), defer=True)
    o.wgrad(wmode, True, n, H, W, Ci, Co, a, sa, e, sg, d2.data_ptr(), 0.05, a_off=2 * n * sa[0], defer=True)
    assert len(E.WGRAD_WS.items) == 2
    E.WGRAD_WS.flush()
    torch.cuda.synchronize()
    want_t = ref(n, 2 * n * sa[0], None) - base
    assert rel_l2(d1.cpu() - base.cpu(), want2.cpu() - base.cpu()) < 1e-5
    assert rel_l2(d2.cpu(), want_t.cpu()) < 1e-5
    # (c) the merged launch: 3n input samples, gradient sample b % 2n
    assert o.wgrad_takes_map(wmode, True, 3 * n, H, W, Ci, Co, a, sa, e, sg, base.data_ptr())
    dm = torch.zeros_like(base)
    o.wgrad(wmode, True, 3 * n, H, W, Ci, Co, a, sa, e, sg, dm.data_ptr(), 0.05, rowscale=rs, g_mod=2 * n)
    torch.cuda.synchronize()
    assert rel_l2(dm.cpu(), (want2 - base + want_t).cpu()) < 1e-5
    # kernels without the map refuse it instead of ignoring it
    with pytest.raises(L.DgError):
        ref_o.wgrad(wmode, True, 3 * n, H, W, Ci, Co, a, sa, e, sg, dm.data_ptr(), 0.05, g_mod=2 * n)


@pytest.mark.parametrize("dtype,force,B,nz,C3", [(torch.float32, 1, 3, 5, 6), (torch.float32, 2, 5, 64, 64),
                                                 (torch.bfloat16, 2, 32, 128, 64)])
def test_proj_gemm_and_wgrad(L, dtype, force, B, nz, C3):
    """Proj (dcgan_eqlr.py:6-16) as a GEMM with the (y,x,c) output order + its weight gradient."""
    from dusty_gan_amd.engine import Ops
    g = torch.Generator().manual_seed(5)
    h0, w0 = 2, 4
    tol = TOL32 if dtype == torch.float32 else TOLBF
    z = torch.randn(B, nz, generator=g)
    w = torch.randn(nz, C3, h0, w0, generator=g)
    b = torch.randn(C3, generator=g)
    if dtype == torch.bfloat16:
        z, w = z.bfloat16().float(), w.bfloat16().float()
    wr = w.clone().requires_grad_()
    y = O.proj(z, wr, b)  # [B,C3,h0,w0]
    Np = h0 * w0 * C3
    o = Ops(dtype)
    o.force = force
    wm = w.permute(2, 3, 1, 0).contiguous().view(Np, nz).to(DEV, dtype)  # master [y][x][c][k]
    zd = z.to(DEV, dtype).contiguous()
    out = torch.empty(B * Np, device=DEV, dtype=dtype)
    bd = b.to(DEV)
    o.conv(L.MODE_GEMM, 0, 1, B, 1, 1, nz, Np, zd, (nz, 0, 1), out, (Np, 0, 1), wm.data_ptr(), 1.0 / math.sqrt(Np),
           L.EPI_LRELU, bias=bd.data_ptr(), bias_mod=C3)
    torch.cuda.synchronize()
    got = out.float().cpu().view(B, h0, w0, C3).permute(0, 3, 1, 2)
    assert rel_l2(got, y) < tol
    gy = torch.randn(y.shape, generator=g)
    lr = torch.where(y > 0, 1.0, 0.2) * math.sqrt(2.0)
    e = gy * lr
    if dtype == torch.bfloat16:
        e = e.bfloat16().float()
    (gw,) = torch.autograd.grad(y, wr, e / lr)
    ed = e.permute(0, 2, 3, 1).contiguous().view(B, Np).to(DEV, dtype)
    dw = torch.full((Np, nz), 7.0, device=DEV)
    o.wgrad(2, 1, 1, 1, B, Np, nz, ed, (0, Np, 1), zd, (0, nz, 1), dw.data_ptr(), 1.0 / math.sqrt(Np), accumulate=0)
    torch.cuda.synchronize()
    assert rel_l2(dw.cpu().view(h0, w0, C3, nz).permute(3, 2, 0, 1), gw) < tol


@pytest.mark.parametrize("ring", [True, False])
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_blur_and_final(L, ring, dtype):
    lib = L.lib()
    g = torch.Generator().manual_seed(3)
    B, H, W, C3 = 3, 32, 64, 8
    x = torch.randn(B, 1, H, W, generator=g).requires_grad_()
    y = O.blur_vh(x, ring)
    xd = x.detach().to(DEV)
    out = torch.empty(B * H * W * 2, device=DEV, dtype=dtype)
    L.check(lib.dg_blur_fwd(xd.data_ptr(), out.data_ptr(), L.dtype_code(dtype), B, H, W, int(ring), None))
    tol = 1e-6 if dtype == torch.float32 else 1e-2
    assert rel_l2(from_nhwc(out.float().cpu(), B, 2, H, W), y) < tol
    gy = torch.randn(y.shape, generator=g)
    if dtype == torch.bfloat16:
        gy = gy.bfloat16().float()
    (gx,) = torch.autograd.grad(y, x, gy)
    dx = torch.empty(B, 1, H, W, device=DEV)
    gyd = nhwc(gy).to(DEV, dtype)  # keep every device operand referenced until the kernel has run
    L.check(lib.dg_blur_bwd(gyd.data_ptr(), L.dtype_code(dtype), dx.data_ptr(), B, H, W, int(ring), None))
    assert rel_l2(dx.cpu(), gx) < 1e-6
    # final dot + its backward
    h0, w0 = 2, 4
    d4 = torch.randn(B, C3, h0, w0, generator=g)
    if dtype == torch.bfloat16:
        d4 = d4.bfloat16().float()
    d4r = d4.clone().requires_grad_()
    wf = torch.randn(1, C3, h0, w0, generator=g).requires_grad_()
    bf = torch.randn(1, generator=g)
    n = h0 * w0 * C3
    yf = F.conv2d(d4r * O.equal_lr_scale(wf), wf, bf).view(B)
    d4d = nhwc(d4).to(DEV, dtype)
    wfd = wf.detach().permute(0, 2, 3, 1).contiguous().view(-1).to(DEV)
    yd = torch.empty(B, device=DEV)
    bfd = bf.to(DEV)
    L.check(lib.dg_final_fwd(d4d.data_ptr(), L.dtype_code(dtype), wfd.data_ptr(), bfd.data_ptr(),
                             1.0 / math.sqrt(n), B, n, yd.data_ptr(), None))
    assert rel_l2(yd.cpu(), yf) < 1e-5
    up = torch.randn(B, generator=g)
    upd = up.to(DEV)
    gd4, gwf = torch.autograd.grad(yf, [d4r, wf], up)
    mask = torch.where(d4 > 0, 1.0, 0.2) * math.sqrt(2.0)
    dd4 = torch.empty_like(d4d)
    db = torch.zeros(C3, device=DEV)
    L.check(lib.dg_final_bwd_data(d4d.data_ptr(), L.dtype_code(dtype), wfd.data_ptr(), upd.data_ptr(), None,
                                  1.0 / math.sqrt(n), B, n, C3, dd4.data_ptr(), db.data_ptr(), None))
    ref = gd4 * mask
    assert rel_l2(from_nhwc(dd4.float().cpu(), B, C3, h0, w0), ref) < (1e-6 if dtype == torch.float32 else 1e-2)
    assert rel_l2(db.cpu(), ref.sum(dim=[0, 2, 3])) < (1e-5 if dtype == torch.float32 else 2e-2)
    dwf = torch.zeros(n, device=DEV)
    L.check(lib.dg_batch_wsum(d4d.data_ptr(), L.dtype_code(dtype), upd.data_ptr(), 1.0 / math.sqrt(n), B, n,
                              dwf.data_ptr(), None))
    assert rel_l2(dwf.cpu().view(h0, w0, C3).permute(2, 0, 1), gwf[0]) < 1e-5


@pytest.mark.parametrize("H,W", [(8, 32), (3, 7), (64, 1024)])
@pytest.mark.parametrize("arch", ["none", "dusty1", "dusty2"])
def test_head_post_fwd_bwd(L, arch, H, W):
    """tanh + GumbelSigmoid + maskout (dcgan_eqlr.py:71; dusty.py:45-59,77-91,107-127) forward and backward: the scalar
    kernels (3x7), the four-pixels-per-thread backward (HW % 4 == 0) and forward-with-sums (HW % 1024 == 0), every
    pixel-major padding, and the bf16 mode's call without the planar copy."""
    lib = L.lib()
    g = torch.Generator().manual_seed(11)
    B = 2
    k = {"none": 0, "dusty1": 1, "dusty2": 2}[arch]
    raw = torch.randn(B, 1 + k, H, W, generator=g).requires_grad_()
    noise = {"pixel": O.logistic_noise(torch.rand(B, 1, H, W, generator=g), torch.rand(B, 1, H, W, generator=g)),
             "image": O.logistic_noise(torch.rand(B, 1, 1, 1, generator=g), torch.rand(B, 1, 1, 1, generator=g))}
    out = {"depth": torch.tanh(raw[:, 0:1])}
    if k:
        out["confidence"] = raw[:, 1:]
    out = O.maskout(out, arch, noise, 1.0, -1.0, True)
    gd = raw.detach().clone().to(DEV)
    mask = torch.empty(B, max(k, 1), H, W, device=DEV)
    depth = torch.empty(B, 1, H, W, device=DEV)
    npx, nim = noise["pixel"].to(DEV).contiguous(), noise["image"].to(DEV).contiguous().view(B)
    if (H * W) % 256 == 0:
        dsum = torch.zeros(B, device=DEV)
        L.check(lib.dg_head_post_fwd_sum(gd.data_ptr(), npx.data_ptr(), nim.data_ptr(), k, 1, 1.0, -1.0, B, H * W,
                                         mask.data_ptr(), depth.data_ptr(), dsum.data_ptr(), None))
        assert rel_l2(dsum.cpu(), out["depth"].detach().sum(dim=[1, 2, 3])) < 1e-5
    else:
        L.check(lib.dg_head_post_fwd(gd.data_ptr(), npx.data_ptr(), nim.data_ptr(), k, 1, 1.0, -1.0, B, H * W,
                                     mask.data_ptr(), depth.data_ptr(), None))
    assert rel_l2(depth.cpu(), out["depth"]) < 1e-5
    if k:
        # (a logit within fp32 rounding of zero may take the other side of the straight-through threshold)
        assert (mask.cpu() != out["mask"].detach()).float().mean() <= (0 if H * W < 1000 else 1e-5)
        assert rel_l2(gd[:, 0:1].cpu(), out["depth_orig"]) < 1e-6
    go = torch.randn(B, 1, H, W, generator=g)
    (graw,) = torch.autograd.grad(out["depth"], raw, go)
    god = go.to(DEV)
    s_d, s_c = 0.25, 0.125
    scale = torch.tensor([s_d] + [s_c] * k).view(1, -1, 1, 1)
    same_mask = torch.equal(mask.cpu(), out["mask"].detach()) if k else True
    ws = torch.zeros(B * 1024, device=DEV)
    for cp in ([4] if k == 2 else [2, 4]):
        for planar in (True, False):
            if not planar and (H * W) % 4:
                draw_pm = torch.empty(B, H, W, cp, device=DEV, dtype=torch.bfloat16)
                rc = lib.dg_head_post_bwd(gd.data_ptr(), npx.data_ptr(), nim.data_ptr(), mask.data_ptr(), god.data_ptr(),
                                          k, 1.0, -1.0, B, H * W, s_d, s_c, None, None, draw_pm.data_ptr(), cp, None, None)
                assert rc == L.DG_EUNSUPPORTED
                continue
            draw = torch.full((B, 1 + k, H, W), 7.0, device=DEV)
            draw_pm = torch.full((B, H, W, cp), 7.0, device=DEV, dtype=torch.bfloat16)
            dbias = torch.zeros(3, device=DEV)
            L.check(lib.dg_head_post_bwd(gd.data_ptr(), npx.data_ptr(), nim.data_ptr(), mask.data_ptr(),
                                         god.data_ptr(), k, 1.0, -1.0, B, H * W, s_d, s_c,
                                         draw.data_ptr() if planar else None, dbias.data_ptr(), draw_pm.data_ptr(), cp,
                                         None if planar else ws.data_ptr(), None))
            assert float(ws.abs().max()) == 0.0          # (the staging slots are left zero: the next launch reuses them)
            tol = 1e-5 if same_mask else 1e-2
            if planar:
                assert rel_l2(draw.cpu(), graw * scale) < tol
            else:
                assert float((draw - 7.0).abs().max()) == 0.0
            assert rel_l2(dbias.cpu()[:1 + k], graw.sum(dim=[0, 2, 3])) < max(tol, 1e-4)
            pm = draw_pm.float().cpu().permute(0, 3, 1, 2)
            assert rel_l2(pm[:, :1 + k], graw * scale) < 1e-2
            if 1 + k < cp:
                assert float(pm[:, 1 + k:].abs().max()) == 0.0
    assert lib.dg_head_post_bwd(gd.data_ptr(), npx.data_ptr(), nim.data_ptr(), mask.data_ptr(), god.data_ptr(),
                                k, 1.0, -1.0, B, H * W, s_d, s_c, None, None, None, 0, None, None) == L.DG_EINVAL


@pytest.mark.parametrize("H,W", [(16, 32), (64, 1024)])
def test_diffaug_fwd_bwd(L, H, W):
    from dusty_gan_amd.utils.diff_augment import DiffAugment
    g = torch.Generator().manual_seed(H)
    B = 4
    A = DiffAugment()
    x = torch.randn(B, 1, H, W, generator=g).requires_grad_()
    for trial in range(3):
        rp = O.draw_augment_params(B, H, W, g)
        if trial == 0:  # extreme draws: clamp-to-border cutout, maximal shifts
            rp["o_x"][:] = torch.tensor([0, H, 0, H])[:B]
            rp["o_y"][:] = torch.tensor([0, W, W, 0])[:B]
            sh, sw = O.translation_shift(H, W)
            rp["t_h"][:] = torch.tensor([-sh, sh, 0, 1])[:B]
            rp["t_w"][:] = torch.tensor([-sw, sw, 1, 0])[:B]
        y = O.diff_augment(x, rp)
        rpd = DiffAugment.params_to_device(rp, DEV)
        yd = A.apply(x.detach().to(DEV), rpd)
        assert rel_l2(yd.cpu(), y) < 1e-5
        gy = torch.randn(y.shape, generator=g)
        (gx,) = torch.autograd.grad(y, x, gy)
        gxd = A.backward(gy.to(DEV), rpd)
        assert rel_l2(gxd.cpu(), gx) < 1e-5
    # the module call draws its own parameters from the Philox stream and stays in range
    rp = A.draw(B, H, W, torch.device(DEV))
    sh, sw = O.translation_shift(H, W)
    assert rp["u_b"].abs().max() <= 1 and rp["t_h"].abs().max() <= sh and rp["t_w"].abs().max() <= sw
    assert rp["o_x"].min() >= 0 and rp["o_x"].max() <= H and rp["o_y"].max() <= W


def test_losses_fetch_reals_adam(L):
    lib = L.lib()
    ops = load("ops")
    pr, pf = torch.from_numpy(ops["ganloss/pred_real"]).view(-1), torch.from_numpy(ops["ganloss/pred_fake"]).view(-1)
    B = pr.numel()
    dy, sc = torch.empty(2 * B, device=DEV), torch.empty(3, device=DEV)
    prd, pfd = pr.to(DEV), pf.to(DEV)
    L.check(lib.dg_nsgan_d(prd.data_ptr(), pfd.data_ptr(), B, 1.0, dy.data_ptr(), dy.data_ptr() + 4 * B,
                           sc.data_ptr(), None))
    assert abs(float(sc[2]) - float(ops["ganloss/nsgan/D"])) < 1e-6
    prr, pfr = pr.clone().requires_grad_(), pf.clone().requires_grad_()
    gr, gf = torch.autograd.grad(O.gan_loss("nsgan", prr, pfr, "D"), [prr, pfr])
    assert rel_l2(dy[:B].cpu(), gr) < 1e-5 and rel_l2(dy[B:].cpu(), gf) < 1e-5
    dyg, scg = torch.empty(B, device=DEV), torch.empty(1, device=DEV)
    L.check(lib.dg_nsgan_g(pfd.data_ptr(), B, 1.0, dyg.data_ptr(), scg.data_ptr(), None))
    assert abs(float(scg[0]) - float(ops["ganloss/nsgan/G"])) < 1e-6
    # fetch_reals against the reference vector
    pol = torch.from_numpy(ops["invert_depth/pol"])
    out = torch.empty_like(pol, device=DEV)
    pold, oned = pol.to(DEV), torch.ones_like(pol).to(DEV)
    L.check(lib.dg_fetch_reals(pold.data_ptr(), oned.data_ptr(), 0.9, 120.0, -1.0,
                               pol.numel(), out.data_ptr(), None))
    assert rel_l2((out.cpu() + 1) / 2, ops["invert_depth/inv"]) < 1e-5
    # Adam + EMA + shadow against the oracle's restatement of torch.optim.Adam
    g = torch.Generator().manual_seed(1)
    n = 1000
    p, gr_, m, v, ema = (torch.randn(n, generator=g) for _ in range(5))
    m.zero_(); v.abs_()
    pc, mc, vc, ec = p.clone(), m.clone(), v.clone(), ema.clone()
    O.adam_update(pc, gr_ * 0.5, mc, vc, 3, 0.002, 0.0, 0.99)
    ec = 0.9 * ec + 0.1 * pc
    pd, gd, md, vd, ed = (t.to(DEV) for t in (p, gr_, m, v, ema))
    sh = torch.empty(n, device=DEV, dtype=torch.bfloat16)
    L.check(lib.dg_adam_ema_step(pd.data_ptr(), gd.data_ptr(), md.data_ptr(), vd.data_ptr(), ed.data_ptr(),
                                 sh.data_ptr(), L.DG_BF16, n, 0.5, 0.002, 0.0, 0.99, 1e-8, 3, 0.9, None))
    assert rel_l2(pd.cpu(), pc) < 1e-6 and rel_l2(vd.cpu(), vc) < 1e-6 and rel_l2(ed.cpu(), ec) < 1e-6
    assert torch.equal(sh.cpu(), pd.cpu().bfloat16())


GAN_METRICS = ["nsgan", "wgan", "lsgan", "hinge", "ragan", "rahinge", "ralsgan"]


@pytest.mark.parametrize("metric", GAN_METRICS)
def test_gan_step_kernels_all_metrics(L, metric):
    """dg_gan_d_step / dg_gan_g_step: loss values against the reference's GANLoss (tests/golden/ops.npz), gradients
    w.r.t. the logits against autograd through the oracle's restatement (pinned to the same vectors on CPU), and
    the step bookkeeping (up / rs / acc / final-bias sum)."""
    from dusty_gan_amd.models.loss import GANLoss, METRICS
    lib = L.lib()
    ops = load("ops")
    code = METRICS.index(metric)
    pr, pf = torch.from_numpy(ops["ganloss/pred_real"]).view(-1), torch.from_numpy(ops["ganloss/pred_fake"]).view(-1)
    crit = GANLoss(metric)
    assert abs(float(crit(pr.to(DEV), pf.to(DEV), "D")) - float(ops[f"ganloss/{metric}/D"])) < 1e-5
    assert abs(float(crit(pr.to(DEV), pf.to(DEV), "G")) - float(ops[f"ganloss/{metric}/G"])) < 1e-5
    g = torch.Generator().manual_seed(5)
    for B in (1, 37, 300):
        yr, yf = torch.randn(B, generator=g) * 1.5, torch.randn(B, generator=g) * 1.5
        a, b = yr.clone().requires_grad_(), yf.clone().requires_grad_()
        loss = O.gan_loss(metric, a, b, "D")
        gr, gf = torch.autograd.grad(0.5 * loss, [a, b])
        dy, up, rs = (torch.empty(2 * B, device=DEV) for _ in range(3))
        acc = torch.tensor([1.0, 2.0, 3.0, 4.0, 5.0], device=DEV)
        fb = torch.tensor([0.25], device=DEV)
        yrd, yfd = yr.to(DEV), yf.to(DEV)
        L.check(lib.dg_gan_d_step(code, 1.0, yrd.data_ptr(), yfd.data_ptr(), B, 0.5, dy.data_ptr(), up.data_ptr(),
                                  rs.data_ptr(), acc.data_ptr(), fb.data_ptr(), None))
        assert (dy.cpu() - torch.cat([gr, gf])).abs().max() < 2e-6 * max(1.0, 37 / B), (metric, B)
        assert torch.equal(up, torch.cat([torch.ones(B, device=DEV), dy[B:]]))
        assert torch.equal(rs, torch.cat([dy[:B], torch.ones(B, device=DEV)]))
        want = torch.tensor([1.0 + float(yr.mean()), 2.0 + float(yf.mean()), 3.0 + float(loss.detach())])
        assert torch.allclose(acc[:3].cpu(), want, atol=1e-5)
        assert abs(float(fb) - 0.25 - float(dy.sum())) < 1e-5
        a, b = yr.clone(), yf.clone().requires_grad_()
        loss_g = O.gan_loss(metric, a, b, "G")
        (gg,) = torch.autograd.grad(0.5 * loss_g, [b])
        dg = torch.empty(B, device=DEV)
        L.check(lib.dg_gan_g_step(code, yrd.data_ptr(), yfd.data_ptr(), B, 0.5, dg.data_ptr(), acc.data_ptr() + 16, None))
        assert (dg.cpu() - gg).abs().max() < 2e-6 * max(1.0, 37 / B), (metric, B)
        assert abs(float(acc[4]) - 5.0 - float(loss_g.detach())) < 1e-5
        rc = lib.dg_gan_g_step(code, None, yfd.data_ptr(), B, 0.5, dg.data_ptr(), acc.data_ptr() + 16, None)
        assert (rc != 0) == crit.relativistic  # D(real) is only optional for the non-relativistic metrics
    assert lib.dg_gan_d_step(7, 1.0, yrd.data_ptr(), yfd.data_ptr(), B, 0.5, dy.data_ptr(), None, None,
                             acc.data_ptr(), None, None) != 0
    with pytest.raises(NotImplementedError):
        GANLoss("nope")(pr.to(DEV), pf.to(DEV), "D")
    with pytest.raises(ValueError):
        crit(pr.to(DEV), pf.to(DEV), "X")


def test_nsgan_step_kernels_match_plain_ones(L):
    """dg_nsgan_d_step / dg_nsgan_g_step / dg_mean_acc (loss + the step's per-sample vectors and running sums in one
    launch) against dg_nsgan_d / dg_nsgan_g, which test_losses_fetch_reals_adam pins to the reference's GANLoss."""
    lib = L.lib()
    g = torch.Generator().manual_seed(9)
    B = 37
    yr, yf = torch.randn(B, generator=g).to(DEV), torch.randn(B, generator=g).to(DEV)
    dy0, sc0 = torch.empty(2 * B, device=DEV), torch.empty(3, device=DEV)
    L.check(lib.dg_nsgan_d(yr.data_ptr(), yf.data_ptr(), B, 0.5, dy0.data_ptr(), dy0.data_ptr() + 4 * B, sc0.data_ptr(), None))
    dy, up, rs = (torch.empty(2 * B, device=DEV) for _ in range(3))
    acc = torch.tensor([1.0, 2.0, 3.0, 4.0, 5.0], device=DEV)
    fb = torch.tensor([0.25], device=DEV)
    L.check(lib.dg_nsgan_d_step(yr.data_ptr(), yf.data_ptr(), B, 0.5, dy.data_ptr(), up.data_ptr(), rs.data_ptr(),
                                acc.data_ptr(), fb.data_ptr(), None))
    assert torch.equal(dy, dy0)
    assert torch.equal(up, torch.cat([torch.ones(B, device=DEV), dy0[B:]]))
    assert torch.equal(rs, torch.cat([dy0[:B], torch.ones(B, device=DEV)]))
    assert torch.allclose(acc[:3], torch.tensor([1.0, 2.0, 3.0], device=DEV) + sc0, atol=1e-6)
    assert abs(float(fb) - 0.25 - float(dy0.sum())) < 1e-6
    dg0, sg0 = torch.empty(B, device=DEV), torch.empty(1, device=DEV)
    L.check(lib.dg_nsgan_g(yf.data_ptr(), B, 0.5, dg0.data_ptr(), sg0.data_ptr(), None))
    dg = torch.empty(B, device=DEV)
    L.check(lib.dg_nsgan_g_step(yf.data_ptr(), B, 0.5, dg.data_ptr(), acc.data_ptr() + 16, None))
    assert torch.equal(dg, dg0) and abs(float(acc[4]) - 5.0 - float(sg0)) < 1e-6
    L.check(lib.dg_mean_acc(yr.data_ptr(), B, acc.data_ptr() + 12, None))
    assert abs(float(acc[3]) - 4.0 - float(yr.mean())) < 1e-6
    # NULL per-sample vectors (no R1) are allowed
    L.check(lib.dg_nsgan_d_step(yr.data_ptr(), yf.data_ptr(), B, 0.5, dy.data_ptr(), None, None, acc.data_ptr(), None, None))


@pytest.mark.parametrize("nb,Np,K", [(4, 200, 8), (32, 1000, 512), (64, 130, 256),
                                     # larger (all-gathered) batches: the optimizer runs as the epilogue of the MFMA GEMM
                                     (64, 1024, 512), (128, 256, 128), (256, 384, 512), (100, 128, 256), (6, 128, 128),
                                     # the timed shapes: Proj.weight of the 64x1024 nets (Np = 4*64*512) at the per-GPU
                                     # batch and at the all-gathered batch of an 8-GPU run
                                     (32, 131072, 512), (256, 131072, 512)])
def test_adam_proj_fused_matches_gemm_plus_adam(L, nb, Np, K):
    """dg_adam_proj_fused (Proj.weight's gradient GEMM inside the optimizer kernel) against the oracle's Adam applied to
    the explicitly formed gradient wscale * dp0^T z of the same bf16 operands, incl. EMA, bf16 shadow, device step."""
    lib = L.lib()
    g = torch.Generator().manual_seed(nb + K)
    dp0 = torch.randn(nb, Np, generator=g).bfloat16()
    z = torch.randn(nb, K, generator=g).bfloat16()
    wscale, gscale, lr, b2, eps, decay, step = 1.0 / math.sqrt(Np), 0.5, 0.002, 0.99, 1e-8, 0.9, 3
    grad = (dp0.float().t() @ z.float()) * wscale  # [Np][K]
    p, v, ema = (torch.randn(Np * K, generator=g) for _ in range(3))
    v.abs_()
    pc, mc, vc, ec = p.clone(), torch.zeros_like(p), v.clone(), ema.clone()
    O.adam_update(pc, grad.reshape(-1) * gscale, mc, vc, step + 1, lr, 0.0, b2)
    ec = decay * ec + (1 - decay) * pc
    pd, vd, ed = (t.to(DEV) for t in (p, v, ema))
    sh = torch.empty(Np * K, device=DEV, dtype=torch.bfloat16)
    dpd, zd = dp0.to(DEV).contiguous(), z.to(DEV).contiguous()
    stepd = torch.full((1,), step, dtype=torch.int64, device=DEV)
    L.check(lib.dg_adam_proj_fused(pd.data_ptr(), vd.data_ptr(), ed.data_ptr(), sh.data_ptr(), L.DG_BF16, dpd.data_ptr(),
                                   zd.data_ptr(), L.DG_BF16, nb, Np, K, wscale, gscale, lr, b2, eps, stepd.data_ptr(),
                                   decay, None), "dg_adam_proj_fused")
    torch.cuda.synchronize()
    assert rel_l2(pd.cpu(), pc) < 1e-5 and rel_l2(vd.cpu(), vc) < 1e-4 and rel_l2(ed.cpu(), ec) < 1e-5
    assert torch.equal(sh.cpu(), pd.cpu().bfloat16())
    # operand types / shapes the kernel refuses (the trainer then forms the gradient with dg_wgrad)
    assert lib.dg_adam_proj_fused(pd.data_ptr(), vd.data_ptr(), None, None, L.DG_BF16, dpd.data_ptr(), zd.data_ptr(),
                                  L.DG_BF16X2, nb, Np, K, wscale, gscale, lr, b2, eps, stepd.data_ptr(), decay,
                                  None) == L.DG_EUNSUPPORTED
    assert lib.dg_adam_proj_fused(pd.data_ptr(), vd.data_ptr(), None, None, L.DG_BF16, dpd.data_ptr(), zd.data_ptr(),
                                  L.DG_BF16 | L.DG_FORCE_FP32X3, nb, Np, K, wscale, gscale, lr, b2, eps, stepd.data_ptr(), decay,
                                  None) == L.DG_EUNSUPPORTED
    if nb > 64:  # neither the LDS-resident kernel (batch) nor the MFMA epilogue (Np % 128) takes this one
        assert lib.dg_adam_proj_fused(pd.data_ptr(), vd.data_ptr(), None, None, L.DG_BF16, dpd.data_ptr(),
                                      zd.data_ptr(), L.DG_BF16, nb, Np - 64, K, wscale, gscale, lr, b2, eps,
                                      stepd.data_ptr(), decay, None) == L.DG_EUNSUPPORTED


@pytest.mark.parametrize("Ci,Co,H,W,B,force", [(24, 20, 16, 32, 4, 1), (64, 128, 4, 128, 4, 2)], ids=["direct", "one-tile"])
def test_bias_gradient_staging_scratch_protocol(L, Ci, Co, H, W, B, force):
    """DgConv.dbias_ws as the direct / one-tile MFMA kernels use it (round 6): (1) the scratch is zero again when the launch
    has finished - word pairs and ticket - so the next launch of the stream can share it; (2) WITHOUT the scratch (a C-ABI caller
    that passes NULL) the same launch adds with float atomics and gives the same sums to rounding."""
    import ctypes as C
    from dusty_gan_amd import engine as E
    g = torch.Generator().manual_seed(5 + Ci)
    e = torch.randn(B, Co, H, W, generator=g)
    wt = torch.randn(Co, Ci, 4, 4, generator=g)
    prev = torch.randn(B, Ci, 2 * H, 2 * W, generator=g)
    rs = torch.rand(B, generator=g) + 0.5
    _, bwd = pack_down(wt)
    s = 1.0 / math.sqrt(Ci * 16)
    _, db = run_conv(L, L.MODE_UP, 1, True, e, bwd, Ci, s, L.EPI_MASK, torch.float32, force, aux=prev, want_db=True, rowscale=rs)
    torch.cuda.synchronize()
    assert E.Ops._dbias_ws, "no staging scratch was handed to the launch"
    for ws in E.Ops._dbias_ws.values():
        assert int(ws.view(torch.int32).ne(0).sum()) == 0, "the staging scratch was not left zero"

    class NoScratch:                       # the library with DgConv.dbias_ws forced to NULL in front of every conv entry point
        def __init__(self, lib):
            self._lib = lib

        def __getattr__(self, name):
            f = getattr(self._lib, name)
            if name not in ("dg_conv", "dg_conv_ex", "dg_conv_plan"):
                return f

            def call(pref, *a):
                pref._obj.dbias_ws = None
                return f(pref, *a)
            return call
    init = E.Ops.__init__

    def init_nows(self, dtype, x3=False):
        init(self, dtype, x3=x3)
        self.lib = NoScratch(self.lib)
    try:
        E.Ops.__init__ = init_nows
        _, db2 = run_conv(L, L.MODE_UP, 1, True, e, bwd, Ci, s, L.EPI_MASK, torch.float32, force, aux=prev, want_db=True, rowscale=rs)
    finally:
        E.Ops.__init__ = init
    assert rel_l2(db2, db) < 1e-6, rel_l2(db2, db)


@pytest.mark.parametrize("x3", [False, True], ids=["fp32", "fp32x3"])
@pytest.mark.parametrize("nb,Np,K", [(8, 256, 128), (32, 1024, 512), (100, 128, 256), (8, 131072, 512)])
def test_adam_proj_fused_takes_fp32_operands(L, nb, Np, K, x3):
    """Round 6: the parity-class modes' Proj.weight runs the same fused launch - fp32 operand rows on the fp32 matrix
    instructions (op_dtype DG_F32) or split into bf16 pairs in registers (DG_F32 | DG_FORCE_FP32X3; operands with 16 mantissa
    bits, as the mode's split storage holds them, are then exact) - with no shadow (the master is what the forward reads).
    Against the oracle's Adam on the float64 gradient; shapes off the 128 grid are refused."""
    lib = L.lib()
    g = torch.Generator().manual_seed(nb * 7 + K)
    dp0 = torch.randn(nb, Np, generator=g)
    z = torch.randn(nb, K, generator=g)
    if x3:   # hi + lo of a split-bf16 pair: 16 mantissa bits
        dp0 = dp0.bfloat16().float() + (dp0 - dp0.bfloat16().float()).bfloat16().float()
        z = z.bfloat16().float() + (z - z.bfloat16().float()).bfloat16().float()
    wscale, gscale, lr, b2, eps, decay, step = 1.0 / math.sqrt(Np), 0.5, 0.002, 0.99, 1e-8, 0.9, 3
    grad = ((dp0.double().t() @ z.double()) * wscale).float()
    p, v, ema = (torch.randn(Np * K, generator=g) for _ in range(3))
    v.abs_()
    pc, mc, vc, ec = p.clone(), torch.zeros_like(p), v.clone(), ema.clone()
    O.adam_update(pc, grad.reshape(-1) * gscale, mc, vc, step + 1, lr, 0.0, b2)
    ec = decay * ec + (1 - decay) * pc
    pd, vd, ed = (t.to(DEV) for t in (p, v, ema))
    dpd, zd = dp0.to(DEV).contiguous(), z.to(DEV).contiguous()
    stepd = torch.full((1,), step, dtype=torch.int64, device=DEV)
    op = L.DG_F32 | (L.DG_FORCE_FP32X3 if x3 else 0)
    L.check(lib.dg_adam_proj_fused(pd.data_ptr(), vd.data_ptr(), ed.data_ptr(), None, L.DG_F32, dpd.data_ptr(), zd.data_ptr(),
                                   op, nb, Np, K, wscale, gscale, lr, b2, eps, stepd.data_ptr(), decay, None), "dg_adam_proj_fused")
    torch.cuda.synchronize()
    # (fp32x3 drops the lo x lo products: 2^-16 relative per product, far below what Adam's normalised step passes on)
    assert rel_l2(pd.cpu(), pc) < 1e-5 and rel_l2(vd.cpu(), vc) < 1e-4 and rel_l2(ed.cpu(), ec) < 1e-5
    assert lib.dg_adam_proj_fused(pd.data_ptr(), vd.data_ptr(), None, None, L.DG_F32, dpd.data_ptr(), zd.data_ptr(), op, nb,
                                  Np - 64, K, wscale, gscale, lr, b2, eps, stepd.data_ptr(), decay, None) == L.DG_EUNSUPPORTED


def test_philox_known_answer(L):
    """Philox4x32-10 known-answer vectors from the Random123 distribution (kat_vectors):
    counter=0,key=0 -> 6627e8d5 e169c58d bc57ac4c 9b00dbd8; counter=ff..,key=ff.. -> 408f276d 41c83b0e a20bc7c6 6d5451fd"""
    lib = L.lib()
    out = torch.empty(4, dtype=torch.int32, device=DEV)
    L.check(lib.dg_philox_bits(0, 0, 0, 1, out.data_ptr(), None))
    got = [int(v) & 0xFFFFFFFF for v in out.cpu().tolist()]
    assert got == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    L.check(lib.dg_philox_bits(2**64 - 1, 2**64 - 1, 2**64 - 1, 1, out.data_ptr(), None))
    got = [int(v) & 0xFFFFFFFF for v in out.cpu().tolist()]
    assert got == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    # distribution sanity of the derived draws
    n = 1 << 20
    u = torch.empty(n, device=DEV)
    L.check(lib.dg_philox_fill(123, 0, 0, 0, 0.0, 1.0, 0, 1, n, u.data_ptr(), None))
    assert 0.0 <= float(u.min()) and float(u.max()) < 1.0 and abs(float(u.mean()) - 0.5) < 2e-3
    z = torch.empty(n, device=DEV)
    L.check(lib.dg_philox_fill(123, 1, 0, 1, 0.0, 1.0, 0, 1, n, z.data_ptr(), None))
    assert abs(float(z.mean())) < 5e-3 and abs(float(z.std()) - 1.0) < 5e-3


@pytest.mark.parametrize("Ci,Co,H,W,B,dtype,cforce,wforce,family,variant", [
    (6, 4, 8, 16, 3, torch.float32, 1, 1, 1, 1),       # direct conv + direct weight gradient (narrow nets, golden cases)
    (24, 20, 16, 32, 4, torch.float32, 1, 1, 1, 1),
    (64, 128, 4, 128, 4, torch.float32, 2, 2, 2, 2),   # one tile per workgroup + register-staged MFMA weight gradient (fp32 modes)
    (128, 64, 8, 64, 4, torch.bfloat16, None, 6, 0, 2),  # ... that weight-gradient kernel at bf16 (force 6: behind the LDS-DMA form)
    (128, 256, 2, 64, 8, torch.float32, 4, 2, 4, 2),   # lock-step persistent conv at fp32: one bias-gradient row per workgroup
], ids=["direct-6x4", "direct-24x20", "one-tile-fp32", "register-staged-bf16", "lock-step-fp32"])
def test_kernels_off_the_timed_path_sum_in_a_fixed_order(L, Ci, Co, H, W, B, dtype, cforce, wforce, family, variant):
    """Round 6: the kernels the fp32 modes and narrow nets run no longer add with float atomics in arrival order.
    Weight gradients: the register-staged MFMA kernel and the direct kernel store one partial per K split / slab in the
    split-K workspace (dg_wgrad_plan reports splits > 1 and ws_floats), summed by dg_wgrad_reduce in index order.  Bias
    gradients: the direct kernel and the one-tile MFMA kernel sum as two-word fixed point (common.h dg_fix2: exact for every
    float down to 2^-36) in LDS / across workgroups through DgConv.dbias_ws; the lock-step persistent kernel at fp32 keeps one
    LDS row per wave row and leaves one partial row per workgroup (DgConvPlan.dbias_rows).  Checked: the plans say so, two
    launches over the same data agree BIT FOR BIT (data spread over six orders of magnitude, so that another order of the
    additions would show), and the bias-gradient sums match a float64 sum of the launch's own output (the values of the
    weight gradients are test_down_fwd_bwd_wgrad's business: same kernels, same workspace path)."""
    from dusty_gan_amd import engine as E
    from dusty_gan_amd.engine import Ops
    assert E.DETERMINISTIC
    g = torch.Generator().manual_seed(Ci * 131 + Co)
    mag = lambda *s: torch.randn(*s, generator=g) * torch.exp(3.0 * torch.randn(*s, generator=g))
    e = mag(B, Co, H, W)
    x = mag(B, Ci, 2 * H, 2 * W)
    wt = torch.randn(Co, Ci, 4, 4, generator=g)
    prev = torch.randn(B, Ci, 2 * H, 2 * W, generator=g)
    rs = torch.rand(B, generator=g) + 0.5
    if dtype == torch.bfloat16:
        e, x, wt = e.bfloat16().float(), x.bfloat16().float(), wt.bfloat16().float()
    _, bwd = pack_down(wt)
    s = 1.0 / math.sqrt(Ci * 16)
    # ---- bias-gradient sums of the backward-data pass
    if cforce is not None:
        E.TRACE = []
        try:
            outs = [run_conv(L, L.MODE_UP, 1, True, e, bwd, Ci, s, L.EPI_MASK, dtype, cforce, aux=prev, want_db=True, rowscale=rs)
                    for _ in range(2)]
        finally:
            trace, E.TRACE = E.TRACE, None
        assert {t[1] for t in trace if t[0] == "conv"} == {family}, trace
        (dx1, db1), (dx2, db2) = outs
        assert torch.equal(db1, db2), (db1 - db2).abs().max()
        ref = (dx1.double() * rs.double().view(B, 1, 1, 1)).sum(dim=[0, 2, 3])
        assert rel_l2(db1.double(), ref) < 2e-6, rel_l2(db1.double(), ref)
    # ---- weight gradient
    o = Ops(dtype)
    o.force = wforce
    xd, ed = nhwc(x).to(DEV, dtype), nhwc(e).to(DEV, dtype)
    dw0 = torch.zeros(16, Ci, Co, device=DEV)
    p = o._wgrad_params(0, True, B, H, W, Ci, Co, xd, (4 * H * W * Ci, Ci, 1), ed, (H * W * Co, Co, 1), dw0.data_ptr(), s, None,
                        None, None, 0, 0, 0)
    pl = o.wgrad_plan(p, 1)
    assert pl.variant == variant and pl.splits > 1 and pl.ws_floats == pl.splits * 16 * Ci * Co, (pl.variant, pl.splits, pl.ws_floats)
    dws = []
    for _ in range(2):
        dw = torch.zeros(16, Ci, Co, device=DEV)
        o.wgrad(0, True, B, H, W, Ci, Co, xd, (4 * H * W * Ci, Ci, 1), ed, (H * W * Co, Co, 1), dw.data_ptr(), s, rowscale=rs.to(DEV))
        torch.cuda.synchronize()
        dws.append(dw.cpu())
    assert torch.isfinite(dws[0]).all() and dws[0].abs().max() > 0
    assert torch.equal(dws[0], dws[1]), (dws[0] - dws[1]).abs().max()
