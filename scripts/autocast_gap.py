"""How far is stock PyTorch bf16 autocast (what `enable_amp` means in the reference, with bf16 instead of fp16) from
its own fp32 on one oracle step?  Reference point for the bf16 tolerances of tests/test_gpu_step.py."""
import sys; sys.path.insert(0, ".")
import torch
from oracle import dusty_oracle as O
from tests.golden_util import rel_l2

torch.manual_seed(4321)
arch, shape, nz, cb, cm, B = "dusty2", (64, 256), 128, 64, 256, 4
gen = torch.Generator().manual_seed(0)
G = O.init_G(f"{arch}/dcgan_eqlr", nz, cb, cm, shape, gen)
D = O.init_D(1, cb, cm, shape, gen)
H, W = shape
x = torch.rand(B, 1, H, W, generator=gen) * 2 - 1
rand = {"z": torch.randn(B, nz, generator=gen),
        "noise": {"pixel": O.logistic_noise(torch.rand(B, 1, H, W, generator=gen), torch.rand(B, 1, H, W, generator=gen)),
                  "image": O.logistic_noise(torch.rand(B, 1, 1, 1, generator=gen), torch.rand(B, 1, 1, 1, generator=gen))},
        "aug": [O.draw_augment_params(B, H, W, gen) for _ in range(4)]}
cfg = O.StepConfig(arch=arch, lr_g=0.0, lr_d=0.0, ema_decay=1.0)

def run(amp):
    Gc = {k: v.clone() for k, v in G.items()}; Dc = {k: v.clone() for k, v in D.items()}
    with torch.autocast(device_type="cpu", dtype=torch.bfloat16, enabled=amp):
        sc, ex = O.train_step(Gc, Dc, {k: v.clone() for k, v in Gc.items()}, O.new_optim_state(Gc), O.new_optim_state(Dc),
                              1, cfg, x, rand, return_grads=True)
    return sc, ex

s32, e32 = run(False)
s16, e16 = run(True)
for k in ("depth_orig", "confidence"):
    print(k, rel_l2(e16["synth"][k].float(), e32["synth"][k]))
for tag in ("grad_D", "grad_G"):
    for k, v in e32[tag].items():
        if v.abs().max() > 0:
            print(tag, k, f"{rel_l2(e16[tag][k].float(), v):.3f}")
