"""Down1 forward (thin_s2_mfma) at bench size against torch: where do mismatches sit (sample, row, x tile)?"""
import math, sys
sys.path.insert(0, ".")
import torch
from dusty_gan_amd import _lib as L
from dusty_gan_amd.engine import Ops, MaskBits
from oracle import dusty_oracle as O
from tests.golden_util import rel_l2
from tests.test_gpu_ops import nhwc, from_nhwc
B, Hc, Wc, Ci, Co = int(sys.argv[1]) if len(sys.argv) > 1 else 32, 32, 512, 2, 64
g = torch.Generator().manual_seed(1)
x = torch.randn(B, Ci, 2 * Hc, 2 * Wc, generator=g).bfloat16().float()
w = torch.randn(Co, Ci, 4, 4, generator=g)
b = torch.randn(Co, generator=g)
wq = w.bfloat16().float()
y = O.down(x, wq, b, True)
s = 1.0 / math.sqrt(Ci * 16)
o = Ops(torch.bfloat16); o.force = 3
xd = nhwc(x).to("cuda", torch.bfloat16)
coci = w.permute(2, 3, 0, 1).contiguous().to("cuda", torch.bfloat16)
out = torch.empty(B * Hc * Wc * Co, device="cuda", dtype=torch.bfloat16)
MaskBits.register(out)
o.conv(L.MODE_S2, 0, True, B, Hc, Wc, Ci, Co, xd, (4 * Hc * Wc * Ci, Ci, 1), out, (Hc * Wc * Co, Co, 1), coci.data_ptr(), s,
       L.EPI_LRELU, bias=b.to("cuda").data_ptr(), bias_mod=Co)
torch.cuda.synchronize()
got = from_nhwc(out.float().cpu(), B, Co, Hc, Wc)
print("rel_l2", rel_l2(got, y), "nan", int(torch.isnan(got).sum()))
err = (got - y).abs().amax(dim=1)  # [B, Hc, Wc]
bad = (err > 0.1) | torch.isnan(err)
print("bad pixels", int(bad.sum()), "of", bad.numel())
tiles = bad.view(B, Hc, Wc // 32, 32).any(dim=3)   # [B, Hc, tiles_x]
idx = tiles.nonzero()
print("bad tiles", len(idx), "of", tiles.numel())
lin = (idx[:, 0] * Hc + idx[:, 1]) * (Wc // 32) + idx[:, 2]
print("first bad linear tile ids", lin[:40].tolist())
nt = tiles.numel(); nw = min((nt + 3) // 4, 768) * 4; tq, tr = nt // nw, nt % nw
print("tiles per wave", tq, "+1 for the first", tr, "waves")
# position of each bad tile inside its wave's range
pos = []
for t in lin[:2000].tolist():
    gw = t // (tq + 1) if t < tr * (tq + 1) else tr + (t - tr * (tq + 1)) // tq
    t0 = gw * tq + min(gw, tr)
    pos.append(t - t0)
print("positions in the wave's range (histogram)", torch.bincount(torch.tensor(pos)).tolist() if pos else [])
