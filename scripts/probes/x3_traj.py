"""Probe: per-step scalars of the same net in fp32, fp32x3 with split storage and fp32x3 with register split (same seeds)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from dusty_gan_amd.trainers.dcgan_amp import Trainer
from tests.test_gpu_step import make_trainer

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
graph = os.environ.get("DUSTY_GAN_GRAPH", "1")
runs = {}
for name, split, pairs in (("fp32", "0", False), ("x3pairs", "1", True), ("x3regs", "1", False)):
    os.environ["DUSTY_GAN_FP32_SPLIT"] = split
    Trainer.fp32_pairs_default = pairs
    torch.manual_seed(31)
    tr = make_trainer("dusty2", True, (64, 1024), 128, 64, 256, 8, amp=False)
    runs[name] = [dict(tr.step(i).items()) for i in range(n)]
    del tr
keys = list(runs["fp32"][0].keys())
for i in range(n):
    print(i, " | ".join(f"{k.split('loss/')[1]}: " + " ".join(f"{runs[r][i][k]:+.5f}" for r in runs) for k in keys[:4]), flush=True)
