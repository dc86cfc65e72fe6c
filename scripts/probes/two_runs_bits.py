"""Which parameters differ between two runs from one seed, per numeric mode and net size (the determinism work list of round 6).
usage: python scripts/probes/two_runs_bits.py"""
import os
import sys

import torch

sys.path.insert(0, ".")
from tests.test_gpu_step import make_trainer  # noqa: E402

CASES = [  # name, arch, size, latent, ch lo, ch hi, B, amp, x3, n_acc, steps
    ("fp32 exact 64x1024 ch64-512", "dusty2", (64, 1024), 128, 64, 512, 8, False, False, 1, 3),
    ("fp32 exact 64x256 ch64-256 acc2", "dusty1", (64, 256), 128, 64, 256, 8, False, False, 2, 3),
    ("bf16 64x256 ch64-256 acc2", "dusty1", (64, 256), 128, 64, 256, 8, True, False, 2, 3),
    ("fp32 tiny 32x64 ch4-16", "dusty2", (32, 64), 8, 4, 16, 2, False, False, 1, 3),
    ("bf16 tiny 32x64 ch4-16", "dusty2", (32, 64), 8, 4, 16, 2, True, False, 1, 3),
    ("fp32 none 32x128 ch8-32", "none", (32, 128), 16, 8, 32, 4, False, False, 1, 3),
    ("fp32x3 64x256 ch64-256", "dusty2", (64, 256), 128, 64, 256, 8, False, True, 1, 3),
]
for name, arch, size, nz, lo, hi, B, amp, x3, n_acc, steps in CASES:
    os.environ["DUSTY_GAN_FP32_SPLIT"] = "1" if x3 else "0"

    def run():
        torch.manual_seed(99)
        tr = make_trainer(arch, True, size, nz, lo, hi, B, amp=amp, n_acc=n_acc)
        for i in range(steps):
            tr.step(i)
        torch.cuda.synchronize()
        return tr
    a, b = run(), run()
    bad = []
    for net in ("G", "D"):
        sa, sb = getattr(a, net).store, getattr(b, net).store
        for k, sg in sa.seg.items():
            x, y = sa.flat[sg.off:sg.off + sg.numel], sb.flat[sg.off:sg.off + sg.numel]
            if not torch.equal(x, y):
                bad.append(f"{net}.{k} ({(x - y).abs().max().item():.2e})")
    print(f"{name}: {'IDENTICAL' if not bad else 'differ: ' + ', '.join(bad)}", flush=True)
