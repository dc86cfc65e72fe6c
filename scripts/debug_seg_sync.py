"""debug: ONE process, DUSTY_GAN_FORCE_SEG=1 (segmented replay, no collectives): does torch.cuda.synchronize() between
replays break the next replays?"""
import os, sys
sys.path.insert(0, ".")
os.environ["DUSTY_GAN_FORCE_SEG"] = os.environ.get("DUSTY_GAN_FORCE_SEG", "1")
import torch
from tests.test_gpu_step import make_trainer
torch.manual_seed(300)
tr = make_trainer("none", True, (64, 1024), 512, 64, 512, 8, amp=True)
for i in range(8):
    s = tr.step(i)
    if i % 2 == 1:
        torch.cuda.synchronize()
    print(i, "graph" if tr._graph is not None else "eager", [f"{x:.4g}" for x in s.values()], flush=True)
