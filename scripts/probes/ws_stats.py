"""Probe: the split-K workspaces after a few benchmark steps (per stream: buffer size, high-water mark, growths, early flushes)
and the weight-gradient / reduce launches of one eager step in issue order."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench
from dusty_gan_amd import engine as E

args = bench.parse(["--no-cpu-baseline", "--no-other-configs"])
tr, arch = bench.make_trainer(args, 0, 0, 1)
for i in range(6):
    tr.step(i)
torch.cuda.synchronize()
for key, ws in E.WGRAD_WS._by_stream.items():
    print("stream", key, "buf MB", None if ws.buf is None else ws.buf.numel() * 4 / 2**20, "hwm MB", ws.hwm * 4 / 2**20,
          "grows", ws.grows, "early_flushes", ws.early_flushes, "refused", ws.refused, "pending", len(ws.items))
os.environ["DUSTY_GAN_GRAPH"] = "0"
tr.use_graph = False
E.TRACE = []
orig = E.WgradWorkspace._reduce_pending


def traced(self):
    E.TRACE.append(("reduce", len(self.items), self.pos * 4 / 2**20))
    return orig(self)


E.WgradWorkspace._reduce_pending = traced
tr.step(7)
torch.cuda.synchronize()
for t in E.TRACE:
    if t[0] in ("wgrad", "wgrad_group", "reduce"):
        print(t)
