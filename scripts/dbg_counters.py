import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tests.test_gpu_step import make_trainer
from dusty_gan_amd import _lib as L

def run(graph):
    os.environ["DUSTY_GAN_GRAPH"] = "1" if graph else "0"
    torch.manual_seed(77)
    tr = make_trainer("none", True, (32, 64), 8, 4, 16, 4)
    rows = []
    for i in range(7):
        o0 = int(tr.rng.ctr.item())
        p0 = {k: v[1] for k, v in L.Counters.pending.items()}
        if i >= 2:
            ev = tr.sample_latents(4).clone()
        o1 = int(tr.rng.ctr.item())
        tr.step(i)
        torch.cuda.synchronize()
        o2 = int(tr.rng.ctr.item())
        z = tr._g_engines()[0].zT.float().view(4, -1)[0, :3].tolist()
        rows.append((i, o0, o1, o2, len(p0), [round(v, 3) for v in z], int(tr.A._rng.ctr.item())))
    return rows
for g in (True, False):
    print("graph" if g else "eager")
    for r in run(g):
        print("  ", r)
