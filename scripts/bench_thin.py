"""time the thin (Down1 / Head) kernels inside a real step: per-kernel averages from the engine's PROFILE hook"""
import sys
sys.path.insert(0, ".")
import torch
from bench import make_trainer, parse
from dusty_gan_amd import engine as E
sys.argv = [sys.argv[0]] + sys.argv[1:]
args = parse()
tr, arch = make_trainer(args, 0, 0, 1)
for i in range(3):
    tr.step(i)
E.PROFILE = []
for i in range(3):
    tr.step(i)
torch.cuda.synchronize()
rec, E.PROFILE = E.PROFILE, None
agg = {}
for name, flops, nbytes, e0, e1, tag in rec:
    if "thin" not in name:
        continue
    a = agg.setdefault((name, tag), [0.0, 0, nbytes])
    a[0] += e0.elapsed_time(e1); a[1] += 1
for (name, tag), (ms, n, nb) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print(f"{name:18s} {tag:44s} n={n:2d} avg {1e3*ms/n:7.1f} us  {nb/(ms/n*1e-3)/1e9:7.0f} GB/s algorithmic")
