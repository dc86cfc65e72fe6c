"""debug: single graph replay + explicit stream sync; dump state only AFTER garbage shows up (no extra GPU work before)"""
import os, sys
sys.path.insert(0, ".")
import torch
from tests.test_gpu_step import make_trainer
torch.manual_seed(300)
tr = make_trainer("none", True, (64, 1024), 512, 64, 512, 8, amp=True)
def stats(name, t):
    t = t.float()
    return f"{name}: max|.| {float(t.abs().max()):.3g} nan {int(torch.isnan(t).sum())}"
for i in range(6):
    s = tr.step(i)
    if i % 2 == 1:
        torch.cuda.current_stream().synchronize()
    vals = list(s.values())
    print(i, "graph" if tr._graph is not None else "eager", [f"{x:.4g}" for x in vals], flush=True)
    if any(abs(v) > 1e3 or v != v for v in vals):
        D, G = tr.D.store, tr.G.store
        deng = tr.D.engine()
        geng = tr._g_engines()[0]
        print("   ", stats("D.flat", D.flat), stats("D.grad", D.grad), stats("D.v", D.v), "final_b", float(D.view("final_b")[0]),
              "final_b.grad", float(D.view("final_b", D.grad)[0]))
        for name in D.seg:
            print("      D", name, stats("p", D.view(name)), stats("g", D.view(name, D.grad)), stats("sh", D.shadow[D.seg[name].off:D.seg[name].off + D.seg[name].numel]))
        print("   ", stats("G.flat", G.flat), stats("G.grad[tail]", G.grad[G.seg["proj_b"].off:]), stats("y", deng.y[:24]))
        for k in range(5):
            print("      h", k, stats("h", deng.h[k]), stats("e", deng.e[k]))
        print("   ", stats("gout", geng.gout), stats("depth", geng.depth), [stats(f"a{k}", geng.a[k]) for k in range(4)])
        print("    stepD", int(tr.optim_D._step_dev), "stepG", int(tr.optim_G._step_dev), "scal", tr._dev_scal.tolist())
        break
