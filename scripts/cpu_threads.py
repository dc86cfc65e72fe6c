"""Thread sweep of the CPU oracle step on the GPU box's host (picks bench.py's --cpu-threads default)."""
import sys, time
sys.path.insert(0, ".")
import torch
from oracle import dusty_oracle as O

H, W, B = 64, 1024, 4
for nt in (8, 16, 32, 64, 128):
    torch.set_num_threads(nt)
    gen = torch.Generator().manual_seed(0)
    G = O.init_G("none/dcgan_eqlr", 512, 64, 512, (H, W), gen)
    D = O.init_D(1, 64, 512, (H, W), gen)
    Ge = {k: v.clone() for k, v in G.items()}
    oG, oD = O.new_optim_state(G), O.new_optim_state(D)
    cfg = O.StepConfig(arch="none")
    ts = []
    for it in range(2):
        x = torch.rand(B, 1, H, W, generator=gen) * 2 - 1
        rand = {"z": torch.randn(B, 512, generator=gen), "noise": None,
                "aug": [O.draw_augment_params(B, H, W, gen) for _ in range(4)]}
        t0 = time.perf_counter()
        O.train_step(G, D, Ge, oG, oD, it + 1, cfg, x, rand)
        ts.append(time.perf_counter() - t0)
    print(f"threads {nt}: {ts[-1]:.2f} s/step at B={B} -> {B/ts[-1]:.2f} img/s", flush=True)
    if ts[-1] > 30:
        break
