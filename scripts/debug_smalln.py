import sys, math; sys.path.insert(0,'.')
import torch
from dusty_gan_amd import _lib as L
from dusty_gan_amd.engine import Ops
torch.manual_seed(0)
B,Hc,Wc,Co,Ci=2,16,128,64,2
e=torch.randn(B,Hc,Wc,Co).bfloat16()
w=torch.randn(16,Ci,Co).bfloat16()
outs=[]
for dt in (torch.float32, torch.bfloat16):
    o=Ops(dt); o.force=3
    ed=e.to('cuda',dt).contiguous(); wd=w.to('cuda',dt).contiguous()
    dx=torch.zeros(B*4*Hc*Wc*Ci,device='cuda',dtype=dt)
    o.conv(L.MODE_UP,1,True,B,Hc,Wc,Co,Ci,ed,(Hc*Wc*Co,Co,1),dx,(4*Hc*Wc*Ci,Ci,1),wd.data_ptr(),0.1,L.EPI_LINEAR)
    torch.cuda.synchronize()
    outs.append(dx.float().cpu().view(B,2*Hc,2*Wc,Ci))
d=(outs[0]-outs[1]).abs()
ref=outs[0].abs().mean()
print("mean abs ref", ref.item(), "max diff", d.max().item())
bad=(d>0.05*ref).nonzero()
print("bad count", bad.shape[0], "of", d.numel())
import collections
print("bad rows", collections.Counter(bad[:,1].tolist()).most_common(8))
print("bad cols", sorted(collections.Counter(bad[:,2].tolist()).items())[:40])
