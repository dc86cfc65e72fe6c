"""Micro-benchmark of the MFMA conv / wgrad kernels on the layer shapes of the 64x1024 step (random data, HIP events,
interleaved rounds in one process).  usage: python scripts/bench_conv.py [bf16|fp32] [B]"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from dusty_gan_amd import _lib as L
from dusty_gan_amd.engine import Ops

dtype = torch.bfloat16 if (len(sys.argv) < 2 or sys.argv[1] == "bf16") else torch.float32
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
dev = "cuda"
o = Ops(dtype)
o.force = 2
torch.manual_seed(0)

# (name, mode, adj, Hc, Wc, K, N, batch multiplier)
CONV = [
    ("down2 fwd", L.MODE_S2, 0, 16, 256, 64, 128, 2), ("down3 fwd", L.MODE_S2, 0, 8, 128, 128, 256, 2),
    ("down4 fwd", L.MODE_S2, 0, 4, 64, 256, 512, 2),
    ("down4 bwd", L.MODE_UP, 1, 4, 64, 512, 256, 2), ("down3 bwd", L.MODE_UP, 1, 8, 128, 256, 128, 2),
    ("down2 bwd", L.MODE_UP, 1, 16, 256, 128, 64, 2),
    ("up1 fwd", L.MODE_UP, 0, 4, 64, 512, 256, 1), ("up2 fwd", L.MODE_UP, 0, 8, 128, 256, 128, 1),
    ("up3 fwd", L.MODE_UP, 0, 16, 256, 128, 64, 1),
    ("up3 bwd", L.MODE_S2, 1, 16, 256, 64, 128, 1), ("up2 bwd", L.MODE_S2, 1, 8, 128, 128, 256, 1),
    ("up1 bwd", L.MODE_S2, 1, 4, 64, 256, 512, 1),
]
WGRAD = [("down2 wg", 0, 16, 256, 64, 128, 2), ("down3 wg", 0, 8, 128, 128, 256, 2), ("down4 wg", 0, 4, 64, 256, 512, 2),
         ("up3 wg", 1, 16, 256, 128, 64, 1), ("up2 wg", 1, 8, 128, 256, 128, 1), ("up1 wg", 1, 4, 64, 512, 256, 1)]


def timeit(fn, iters=10):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


tot_ms, tot_fl = 0.0, 0.0
if len(sys.argv) > 3 and sys.argv[3] == "wgradonly":
    CONV = []
for name, mode, adj, Hc, Wc, K, N, bm in CONV:
    n = B * bm
    if mode == L.MODE_S2:
        hin, win, ho, wo, taps = 2 * Hc, 2 * Wc, Hc, Wc, 16
    else:
        hin, win, ho, wo, taps = Hc, Wc, 2 * Hc, 2 * Wc, 4
    x = torch.randn(n * hin * win * K, device=dev).to(dtype)
    w = torch.randn(16 * N * K, device=dev).to(dtype)
    out = torch.empty(n * ho * wo * N, device=dev, dtype=dtype)
    aux = torch.randn(n * ho * wo * N, device=dev).to(dtype)
    bias = torch.randn(N, device=dev)
    db = torch.zeros(N, device=dev)
    epi = L.EPI_MASK if adj else L.EPI_LRELU
    if dtype == torch.bfloat16 and os.environ.get("BENCH_BITS", "1") != "0":   # the step's form: saved 1-bit leaky-relu masks
        from dusty_gan_amd.engine import MaskBits
        if adj:
            aux._dg_bits = torch.randint(0, 256, (aux.numel() // 8,), device=dev, dtype=torch.uint8)
        else:
            MaskBits.register(out)

    def run():
        o.conv(mode, adj, True, n, Hc, Wc, K, N, x, (hin * win * K, K, 1), out, (ho * wo * N, N, 1), w.data_ptr(),
               0.01, epi, bias=None if adj else bias.data_ptr(), bias_mod=N, aux=aux if adj else None,
               dbias=db.data_ptr() if (adj and not os.environ.get('BENCH_NO_DB')) else None)
    ms = timeit(run)
    fl = 2.0 * n * ho * wo * N * K * taps
    tot_ms += ms
    tot_fl += fl
    print(f"{name:10s} B{n:3d} {Hc:2d}x{Wc:3d} K{K:3d} N{N:3d}: {ms * 1e3:7.1f} us {fl / ms / 1e9:7.1f} TFLOP/s")
    if int(os.environ.get("DG_CONV_DBG", "0")) & 8:  # cycle stamps of workgroup 0 (waves 0 and 4), see conv_mfma_pp.hip
        st = out[:256].view(torch.float32).tolist()
        for k in (0, 4):
            ph = st[64 + 8 * k:64 + 8 * k + 5]
            print(f"    wave{k} phases (cycles from entry; {ph[4]:.0f} tiles): prologue done {ph[0]:.0f}, first matrix half {ph[1]:.0f}, "
                  f"loop done {ph[2]:.0f}, final epilogue + stores acknowledged {ph[3]:.0f}")
        for wv, v in ((f"wave{k}", st[8 * k:8 * k + 6]) for k in (0, 4)):
            tot = sum(v) or 1.0
            print(f"    {wv}: cycles load {v[0]:.0f} ({100 * v[0] / tot:.0f}%) wait {v[1]:.0f} ({100 * v[1] / tot:.0f}%) bar1 {v[2]:.0f} "
                  f"({100 * v[2] / tot:.0f}%) mfma {v[3]:.0f} ({100 * v[3] / tot:.0f}%) bar2 {v[4]:.0f} ({100 * v[4] / tot:.0f}%) gap {v[5]:.0f} ({100 * v[5] / tot:.0f}%) total {tot:.0f}")
if CONV:
    print(f"conv total {tot_ms * 1e3:.1f} us  {tot_fl / tot_ms / 1e9:.1f} TFLOP/s")

if len(sys.argv) > 3 and sys.argv[3] == "convonly":
    sys.exit(0)
# weight gradients: the step's launches (D layers: ONE launch over 3B input samples real | fake | tangent against the 2B
# gradient chain; G layers: B samples) in four forms - fp32 atomics onto dW, split-K workspace + reduce with the launcher's
# tap-pair choice, with pairs forced, with single taps forced
from dusty_gan_amd import engine as E
FORMS = [("atomics", 2, False), ("ws auto", 2, True), ("ws pairs", 7, True), ("ws single", 8, True)]
tot = {f[0]: 0.0 for f in FORMS}
tot_fl = 0.0
for name, wmode, Hc, Wc, Ci, Co, bm in WGRAD:
    merged = bm == 2          # a D layer: 3B input samples, gradient sample b % 2B
    n = 3 * B if merged else B
    ng = 2 * B if merged else B
    wa = 2 * Wc if wmode == 0 else Wc
    ha = 2 * Hc if wmode == 0 else Hc
    wg = 2 * Wc if wmode == 1 else Wc
    hg = 2 * Hc if wmode == 1 else Hc
    a = torch.randn(n * ha * wa * Ci, device=dev).to(dtype)
    g = torch.randn(ng * hg * wg * Co, device=dev).to(dtype)
    dw = torch.zeros(16 * Ci * Co, device=dev)
    rs = torch.rand(n, device=dev)
    fl = 2.0 * n * Hc * Wc * Ci * Co * 16
    tot_fl += fl
    line = f"{name:10s} B{n:3d} {Hc:2d}x{Wc:3d} Ci{Ci:3d} Co{Co:3d}:"
    for form, force, ws in FORMS:
        o.force, o.use_ws = force, ws
        E.TRACE = []

        def run():
            if merged and not ws:   # the atomics form is what round 2 ran: two launches per D layer
                o.wgrad(wmode, True, ng, Hc, Wc, Ci, Co, a, (ha * wa * Ci, Ci, 1), g, (hg * wg * Co, Co, 1), dw.data_ptr(),
                        0.01, rowscale=rs)
                o.wgrad(wmode, True, B, Hc, Wc, Ci, Co, a, (ha * wa * Ci, Ci, 1), g, (hg * wg * Co, Co, 1), dw.data_ptr(),
                        0.01, a_off=ng * ha * wa * Ci)
            else:
                o.wgrad(wmode, True, n, Hc, Wc, Ci, Co, a, (ha * wa * Ci, Ci, 1), g, (hg * wg * Co, Co, 1), dw.data_ptr(),
                        0.01, rowscale=rs, g_mod=ng if merged else 0)
        ms = timeit(run)
        tr = [t for t in E.TRACE if t[0] == "wgrad"][-1]
        E.TRACE = None
        tot[form] += ms
        line += f"  {form} {ms * 1e3:6.1f} us {fl / ms / 1e9:6.1f} TF (split {tr[3]}{' pairs' if tr[4] else ''})"
    print(line)
for form, _, _ in FORMS:
    print(f"wgrad total [{form:9s}] {tot[form] * 1e3:.1f} us  {tot_fl / tot[form] / 1e9:.1f} TFLOP/s (reduce launches included)")
