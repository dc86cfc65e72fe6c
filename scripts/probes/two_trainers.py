"""two trainers of one process stepping alternately vs alone: where do they diverge?"""
import os, sys
sys.path.insert(0, ".")
import torch
from tests.test_gpu_step import make_trainer
from tests.golden_util import rel_l2

def make(seed):
    torch.manual_seed(seed)
    return make_trainer("none", True, (32, 64), 8, 4, 16, 4)

def solo(seed, n, gen):
    tr = make(seed)
    for i in range(n):
        tr.step(i)
        if gen: tr.generate()
    return tr

def cmp(tag, x, y):
    print(tag, {net: float(rel_l2(getattr(x, net).store.flat.cpu(), getattr(y, net).store.flat.cpu())) for net in ("G", "D", "G_ema")})

for graph in ("1", "0"):
    os.environ["DUSTY_GAN_GRAPH"] = graph
    for gen in (False, True):
        r1, r2 = solo(11, 4, gen), solo(11, 4, gen)
        cmp(f"graph={graph} gen={gen} solo vs solo", r1, r2)
        rb = solo(12, 4, gen)
        a, b = make(11), make(12)
        for i in range(4):
            a.step(i); b.step(i)
            if gen: b.generate(); a.generate()
        cmp(f"graph={graph} gen={gen} interleaved a", a, r1)
        cmp(f"graph={graph} gen={gen} interleaved b", b, rb)
