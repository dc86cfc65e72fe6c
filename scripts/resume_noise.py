"""Run-to-run noise against resume distance for tests/test_gpu_step.py::test_resume_continues_like_the_uninterrupted_run
(prints the three relative-L2 distances per architecture).  usage: python scripts/resume_noise.py"""
import os
import sys
import tempfile

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dusty_gan_amd.trainers.dcgan_amp import Trainer  # noqa: E402
from dusty_gan_amd.utils.config import load_config  # noqa: E402


def rel(a, b):
    return float((a - b).norm() / b.norm())


for arch in ("none", "dusty2"):
    def cfg(resume=None):
        model = {"none": "dcgan_eqlr", "dusty2": "dusty2_dcgan_eqlr"}[arch]
        c = load_config([f"model={model}", "dataset=synthetic", "dataset.shape=[32,64]", "model.gen.in_ch=8", "model.gen.ch_base=4",
                         "model.gen.ch_max=16", "model.dis.ch_base=4", "model.dis.ch_max=16", "solver.batch_size=4",
                         "enable_amp=false", "dataset.pool=3"])
        c.resume = resume
        return c
    lc = {"gpu": 0, "ngpus": 1, "batch_size": 4, "num_workers": 0}
    par = lambda t: torch.cat([getattr(t, n).store.flat.cpu() for n in ("G", "D", "G_ema")])
    runs = []
    for _ in range(4):
        torch.manual_seed(11)
        a = Trainer(cfg(), lc)
        for i in range(6):
            a.step(i)
        runs.append(par(a))
    with tempfile.TemporaryDirectory() as td:
        torch.manual_seed(11)
        b = Trainer(cfg(), lc)
        for i in range(3):
            b.step(i)
        path = b.save_models("mid", 12, directory=td)
        torch.manual_seed(999)
        c = Trainer(cfg(resume=path), lc)
        for i in range(3, 6):
            c.step(i)
        sd = torch.load(path, weights_only=False)
        sd.pop("resume_state")
        torch.save(sd, os.path.join(td, "np.pth"))
        torch.manual_seed(999)
        d = Trainer(cfg(resume=os.path.join(td, "np.pth")), lc)
        for i in range(3, 6):
            d.step(i)
    print(arch, "run-to-run", [f"{rel(r, runs[0]):.2e}" for r in runs[1:]], "resumed", f"{rel(par(c), runs[0]):.2e}",
          "no position", f"{rel(par(d), runs[0]):.2e}")
