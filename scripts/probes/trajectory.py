"""Probe: the logged scalars of the benchmark workload every `every` steps (is an excursion of the losses a transient?).
usage: trajectory.py <steps> <every> [bench.py arguments, e.g. --precision fp32x3]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import bench

n, every = int(sys.argv[1]), int(sys.argv[2])
extra = sys.argv[3:]                                  # e.g. --precision fp32x3
if "fp32x3" in extra:
    os.environ["DUSTY_GAN_FP32_SPLIT"] = "1"          # (bench.main sets it from --precision; this probe builds the trainer itself)
args = bench.parse(["--no-cpu-baseline", "--no-other-configs"] + extra)
tr, arch = bench.make_trainer(args, 0, 0, 1)
for i in range(10):           # bench.py's warm-up
    tr.step(i)
worst = 0.0
for i in range(n):
    s = tr.step(i)
    if (i + 1) % every == 0:
        d = dict(s.items())
        worst = max(worst, abs(d["loss/D/adversarial"]))
        print(i + 1, {k.split("loss/")[1]: round(v, 3) for k, v in d.items()}, flush=True)
print("worst |loss/D/adversarial| seen:", worst)
