"""how far apart do the graph-replayed and the eager accumulated step end (bf16, n_acc = 2)? - the numbers behind the bounds of
tests/test_gpu_step.py::test_accumulated_step_replays_from_a_graph (parameters: rel-L2; scalars: per key and step)"""
import os, sys
sys.path.insert(0, ".")
import torch
from tests.test_gpu_step import make_trainer
from tests.golden_util import rel_l2

def run(graph, seed=515):
    os.environ["DUSTY_GAN_GRAPH"] = "1" if graph else "0"
    torch.manual_seed(seed)
    tr = make_trainer("dusty1", True, (64, 256), 128, 64, 256, 8, amp=True, n_acc=2)
    sc = [dict(tr.step(i).items()) for i in range(5)]
    torch.cuda.synchronize()
    return tr, sc

def worst(sa, sb):
    w = {}
    for x, y in zip(sa, sb):
        for k in x:
            w[k] = max(w.get(k, 0.0), abs(x[k] - y[k]) / max(1.0, abs(y[k])))
    return {k.split("/", 1)[1]: round(v, 4) for k, v in w.items()}

for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    (a, sa), (b, sb), (c, sc) = run(True), run(False), run(False)
    print(rep, "graph vs eager G", round(float(rel_l2(a.G.store.flat.cpu(), b.G.store.flat.cpu())), 5), worst(sa, sb),
          "| eager vs eager", round(float(rel_l2(c.G.store.flat.cpu(), b.G.store.flat.cpu())), 5), worst(sc, sb), flush=True)
