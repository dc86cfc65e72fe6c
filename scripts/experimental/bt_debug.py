"""Debug probe: the big-tile conv (force 11) against the ping-pong conv (force 5) on one layer; prints where outputs / mask bits differ."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from dusty_gan_amd import _lib as L
from dusty_gan_amd.engine import MaskBits, Ops

def pack_bits(t):
    b = (t.float().reshape(-1, 8) > 0).to(torch.int32)
    w = torch.tensor([1, 2, 4, 8, 16, 32, 64, 128], dtype=torch.int32, device=t.device)
    return (b * w).sum(dim=1).to(torch.uint8)

def run(mode, adj, B, Hc, Wc, K, N, epi, force, seed=0, want_bits=True):
    torch.manual_seed(seed)
    dev = "cuda"
    if mode == L.MODE_S2:
        hin, win, ho, wo = 2 * Hc, 2 * Wc, Hc, Wc
    else:
        hin, win, ho, wo = Hc, Wc, 2 * Hc, 2 * Wc
    x = torch.randn(B * hin * win * K, device=dev).bfloat16()
    w = torch.randn(16 * N * K, device=dev).bfloat16()
    out = torch.zeros(B * ho * wo * N, device=dev, dtype=torch.bfloat16)
    aux = torch.randn(B * ho * wo * N, device=dev).bfloat16()
    bias = torch.randn(N, device=dev)
    db = torch.zeros(N, device=dev)
    o = Ops(torch.bfloat16)
    o.force = force
    bits = None
    if epi == L.EPI_LRELU:
        bits = MaskBits.register(out) if want_bits else None
    else:
        aux._dg_bits = pack_bits(aux)
    o.conv(mode, adj, True, B, Hc, Wc, K, N, x, (hin * win * K, K, 1), out, (ho * wo * N, N, 1), w.data_ptr(), 0.02, epi,
           bias=None if epi == L.EPI_MASK else bias.data_ptr(), bias_mod=N, aux=aux if epi == L.EPI_MASK else None,
           dbias=db.data_ptr() if epi == L.EPI_MASK else None)
    torch.cuda.synchronize()
    return out, bits, db, (ho, wo)

for name, args in [("down fwd", (L.MODE_S2, 0, 2, 4, 128, 256, 128, L.EPI_LRELU)),
                   ("down bwd", (L.MODE_UP, 1, 2, 4, 128, 128, 256, L.EPI_MASK)),
                   ("up fwd dual", (L.MODE_UP, 0, 1, 2, 256, 128, 64, L.EPI_LRELU))]:
    if args[7] == L.EPI_LRELU:
        a0 = run(*args, force=11, want_bits=False)[0]
        b0 = run(*args, force=5, want_bits=False)[0]
        print(name, "WITHOUT mask_out: max diff", float((a0.float() - b0.float()).abs().max()))
    a, abits, adb, (ho, wo) = run(*args, force=11)
    b, bbits, bdb, _ = run(*args, force=5)
    N = args[6]
    d = (a.float() - b.float()).abs()
    bad = d > 1e-2 * b.float().abs().max()
    print(name, "max diff", float(d.max()), "bad elements", int(bad.sum()), "of", d.numel())
    if bad.any():
        idx = bad.nonzero().flatten()
        pix, ch = idx // N, idx % N
        print("  bad channels (unique, first 32):", ch.unique()[:32].tolist())
        print("  bad pixel x (unique, first 32):", (pix % wo).unique()[:32].tolist())
        print("  bad pixel rows:", ((pix // wo) % ho).unique().tolist(), "samples:", (pix // (wo * ho)).unique().tolist())
        af, bf = a.float(), b.float()
        for k in idx[:8].tolist():
            g = float(af[k])
            where = (bf == g).nonzero().flatten()[:4].tolist() if g == g else "nan"
            print(f"    elem {k} (pix {k // N} ch {k % N}): got {g} expected {float(bf[k])}; got-value found in expected at {where}")
        nn = torch.isnan(af)
        print("  NaN count", int(nn.sum()), " non-NaN bad", int((bad & ~nn).sum()))
    if abits is not None:
        exp = pack_bits(a)
        mb = (abits != exp)
        print("  mask bytes wrong vs own output:", int(mb.sum()), "of", mb.numel())
        if mb.any():
            idx = mb.nonzero().flatten()
            el = idx * 8
            pix, ch = el // N, el % N
            print("    wrong byte channel offsets:", ch.unique()[:16].tolist(), " x:", (pix % wo).unique()[:40].tolist(),
                  " rows:", ((pix // wo) % ho).unique().tolist(), "got", abits[idx[:6]].tolist(), "exp", exp[idx[:6]].tolist())
    else:
        print("  dbias rel diff:", float((adb - bdb).norm() / bdb.norm()))
