"""Probe: after a few replayed steps, are D's fp32 shadows and their split twins what the master says?"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from dusty_gan_amd import engine as E
from dusty_gan_amd.trainers.dcgan_amp import Trainer
from tests.test_gpu_step import make_trainer

os.environ["DUSTY_GAN_FP32_SPLIT"] = "1"
torch.manual_seed(31)
tr = make_trainer("dusty2", True, (64, 1024), 128, 64, 256, 8, amp=False)
for i in range(int(sys.argv[1])):
    s = dict(tr.step(i).items())
    torch.cuda.synchronize()
    for tag, st in (("D", tr.D.store), ("G", tr.G.backbone.store if hasattr(tr.G, "backbone") else tr.G.store)):
        msg = []
        for name, seg in st.seg.items():
            if seg.kind != "conv":
                continue
            master = st.view(name)                                   # [4,4,ci,co]
            sh = st.shadow[seg.off:seg.off + seg.numel].view(seg.shape)
            d_sh = float((sh - master).abs().max())
            co = st.coci[name].view(16, seg.shape[3], seg.shape[2])
            d_co = float((co - master.view(16, seg.shape[2], seg.shape[3]).transpose(1, 2)).abs().max())
            d_tw = d_tc = -1.0
            if name in st.x2_cico:
                d_tw = float((E.x2_unpack(st.x2_cico[name]).view(seg.shape) - master).abs().max())
                d_tc = float((E.x2_unpack(st.x2_coci[name]).view(16, seg.shape[3], seg.shape[2]) - master.view(16, seg.shape[2], seg.shape[3]).transpose(1, 2)).abs().max())
            msg.append(f"{name}: sh {d_sh:.1e} coci {d_co:.1e} twin {d_tw:.1e} {d_tc:.1e}")
        print(i, tag, " | ".join(msg), flush=True)
    print(i, {k.split("loss/")[1]: round(v, 4) for k, v in s.items()}, tr.launch_mode(), flush=True)
