"""debug: run-to-run spread of the 2-rank (gloo, one GPU) step at the benchmark's size: eager vs eager, graph vs graph,
eager vs graph.  usage: python scripts/debug_ddp_noise.py"""
import os, sys, tempfile
sys.path.insert(0, ".")
import torch, torch.multiprocessing as mp
from tests.test_gpu_ddp import graph_worker
from tests.golden_util import rel_l2

def run(use_graph):
    with tempfile.TemporaryDirectory() as td:
        mp.spawn(graph_worker, args=(2, os.path.join(td, "init"), td, use_graph, True), nprocs=2, join=True)
        return [torch.load(os.path.join(td, f"g{int(use_graph)}_r{r}.pt")) for r in range(2)]

if __name__ == "__main__":
    runs = {"e1": run(False), "e2": run(False), "g1": run(True), "g2": run(True)}
    names = list(runs)
    for i in range(len(names)):
        for j in range(i + 1, len(names)):
            a, b = runs[names[i]][0], runs[names[j]][0]
            print(names[i], names[j], {k: f"{rel_l2(a[k], b[k]):.2e}" for k in ("G", "D", "E")},
                  "scal0", [round(a["scal"][s]["loss/D/adversarial"] - b["scal"][s]["loss/D/adversarial"], 5) for s in range(6)], flush=True)
    print("ranks equal:", [torch.equal(runs[n][0]["D"], runs[n][1]["D"]) for n in names])
