"""debug: single process, graph split into segments by no-op 'collectives' (DUSTY_GAN_FORCE_SEG=1), bench.py's flow"""
import os, sys
sys.path.insert(0, ".")
import torch
from bench import make_trainer, parse
sys.argv = [sys.argv[0], "--batch", os.environ.get("DBG_B", "32"), "--arch", "dusty2"]
args = parse()
tr, arch = make_trainer(args, 0, 0, 1)
last = None
for i in range(4):
    last = tr.step(i)
print("warm", list(last.values())[:3], "segments", sum(isinstance(g, torch.cuda.CUDAGraph) for g in tr._graph))
if os.environ.get("DBG_VAR", "1") == "1":
    torch.cuda.synchronize()
prev = None
for i in range(6):
    cur = tr.step(i)
    if prev is not None:
        print(i, list(prev.values())[:3])
    prev = cur
print(9, list(prev.values())[:3])
