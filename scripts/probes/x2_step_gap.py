import os, sys, torch
sys.path.insert(0, "/root/repo")
from tests.test_gpu_step import make_trainer
def run(x3, seed):
    os.environ["DUSTY_GAN_FP32_SPLIT"] = "1" if x3 else "0"
    torch.manual_seed(seed)
    tr = make_trainer("dusty2", True, (64, 1024), 128, 64, 256, 8, amp=False)
    return [dict(tr.step(i).items()) for i in range(8)]
for seed in (31, 32, 33):
    a, b = run(False, seed), run(True, seed)
    for i in range(8):
        worst = max((abs(a[i][k] - b[i][k]) / max(1.0, abs(a[i][k])), k) for k in a[0])
        print(seed, i, "%.3e" % worst[0], worst[1])
