"""Are the ~1e-3 G-gradient gaps of the fp32 step at full width leaky-relu mask flips?  Count sign disagreements
between the HIP path's saved activations and the oracle's, layer by layer."""
import sys; sys.path.insert(0,'.')
import torch
from oracle import dusty_oracle as O
from tests.test_gpu_step import make_trainer, grads_by_name
from tests.golden_util import rel_l2

arch, shape, nz, B = "dusty2", (64, 1024), 512, 2
tr = make_trainer(arch, True, shape, nz, 64, 512, B, amp=False)
G = {k: v.detach().cpu().clone() for k, v in tr.G.state_dict().items()}
D = {k: v.detach().cpu().clone() for k, v in tr.D.state_dict().items() if not k.endswith("kernel")}
gen = torch.Generator().manual_seed(0)
H, W = shape
pol = torch.rand(B, 1, H, W, generator=gen); mask = torch.rand(B, 1, H, W, generator=gen) > 0.15; pol = pol * mask
rand = {"z": torch.randn(B, nz, generator=gen),
        "noise": {"pixel": O.logistic_noise(torch.rand(B,1,H,W,generator=gen), torch.rand(B,1,H,W,generator=gen)),
                  "image": O.logistic_noise(torch.rand(B,1,1,1,generator=gen), torch.rand(B,1,1,1,generator=gen))},
        "aug": [O.draw_augment_params(B, H, W, gen) for _ in range(4)]}
x_real, m_real = tr.fetch_reals({"depth": pol, "mask": mask})
tr.optimize_D(reals=[(x_real, m_real)], rands=[rand])
geng = tr._mb[0]["geng"]
# oracle G activations
p = "backbone."
h = O.proj(rand["z"], G[p+"0.0.module.weight"], G[p+"0.1.bias"]); acts=[h]
for i in (1,2,3):
    h = O.up(h, G[p+f"{i}.1.module.weight"], G[p+f"{i}.2.bias"]); acts.append(h)
for i,a in enumerate(acts):
    Bc, C, Hh, Ww = a.shape
    mine = geng.a[i].float().cpu().view(Bc, Hh, Ww, C).permute(0,3,1,2)
    flips = ((mine > 0) != (a > 0)).sum().item()
    print(f"G a{i}: rel {rel_l2(mine, a):.2e} sign flips {flips} of {a.numel()}  min|ref| at flips:",
          a[(mine > 0) != (a > 0)].abs().max().item() if flips else 0)
# D activations of the D phase (slot 0..2B = real|fake aug)
deng = tr.D.engine()
xr = O.diff_augment(O.fetch_reals(pol, mask)[0], rand["aug"][0])
out = O.generator(G, rand["z"], arch, rand["noise"])
xf = O.diff_augment(out["depth"], rand["aug"][1])
x = torch.cat([xr, xf])
h = O.blur_vh(x)
for i in (1,2,3,4):
    h = O.down(h, D[f"{i}.1.module.weight"], D[f"{i}.2.bias"])
    n, C, Hh, Ww = h.shape
    mine = deng.h[i][:n*C*Hh*Ww].float().cpu().view(n, Hh, Ww, C).permute(0,3,1,2)
    flips = ((mine > 0) != (h > 0)).sum().item()
    print(f"D h{i}: rel {rel_l2(mine, h):.2e} sign flips {flips} of {h.numel()}")
