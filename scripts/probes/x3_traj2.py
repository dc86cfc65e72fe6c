"""Probe: graph-mode divergence of the split-storage form - which network / kernel?  usage: x3_traj2.py <steps> <variant>
variant: both | g | d (which networks keep split storage)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from dusty_gan_amd.trainers.dcgan_amp import Trainer, _backbone
from tests.test_gpu_step import make_trainer

n, variant = int(sys.argv[1]), sys.argv[2]
runs = {}
for name, pairs in (("regs", False), ("pairs", True)):
    os.environ["DUSTY_GAN_FP32_SPLIT"] = "1"
    Trainer.fp32_pairs_default = pairs
    torch.manual_seed(31)
    tr = make_trainer("dusty2", True, (64, 1024), 128, 64, 256, 8, amp=False)
    if pairs and variant == "g":
        tr.D.fp32_pairs = False
    if pairs and variant == "d":
        _backbone(tr.G).fp32_pairs = False
    runs[name] = [dict(tr.step(i).items()) for i in range(n)]
    print(name, "D x2", tr.D.engine().x2, "G x2", tr._g_engines()[0].x2, tr.launch_mode())
    del tr
keys = list(runs["regs"][0].keys())
for i in range(n):
    print(i, " | ".join(f"{k.split('loss/')[1]}: " + " ".join(f"{runs[r][i][k]:+.5f}" for r in runs) for k in keys), flush=True)
