"""debug: single graph replay + explicit stream sync: what gets corrupted first?"""
import os, sys
sys.path.insert(0, ".")
import torch
from tests.test_gpu_step import make_trainer
mode = os.environ.get("DBG_SYNC", "stream")
torch.manual_seed(300)
tr = make_trainer("none", True, (64, 1024), 512, 64, 512, 8, amp=True)
def stats(name, t):
    t = t.float()
    return f"{name}: max|.| {float(t.abs().max()):.3g} nan {int(torch.isnan(t).sum())}"
for i in range(7):
    s = tr.step(i)
    if i % 2 == 1 and mode == "stream":
        torch.cuda.current_stream().synchronize()
    vals = [f"{x:.4g}" for x in s.values()]
    D, G = tr.D.store, tr.G.store
    deng = tr.D.engine()
    print(i, "graph" if tr._graph is not None else "eager", vals, flush=True)
    print("   ", stats("D.flat", D.flat), stats("D.grad", D.grad), stats("D.v", D.v), "final_b", float(D.view("final_b")[0]),
          "final_b.grad", float(D.view("final_b", D.grad)[0]), flush=True)
    print("   ", stats("G.flat", G.flat), stats("G.grad[tail]", G.grad[G.seg["proj_b"].off:]), stats("y", deng.y[:24]),
          stats("h4", deng.h[4]), stats("h0", deng.h[0]), stats("gout", tr._g_engines()[0].gout), "stepD", int(tr.optim_D._step_dev), "stepG", int(tr.optim_G._step_dev), flush=True)
