"""GAN losses -- reference: models/loss.py:21-88.

On the hot path the NSGAN loss and its gradient w.r.t. the logits are one tiny kernel (csrc/pointwise.hip
nsgan_d / nsgan_g) called by the trainer; this class keeps the reference's dispatch surface (metric names,
ValueError / NotImplementedError behaviour) and evaluates a loss VALUE for callers that hold logits.
"""
import torch
from torch import nn

from .. import _lib as L

METRICS = ("nsgan", "wgan", "lsgan", "hinge", "ragan", "rahinge", "ralsgan")


class GANLoss(nn.Module):
    def __init__(self, metric: str, smoothing: float = 1.0):
        super().__init__()
        self.metric = metric
        self.smoothing = smoothing

    def forward(self, pred_real, pred_fake, mode):
        if mode == "G":
            return self.loss_G(pred_real, pred_fake)
        elif mode == "D":
            return self.loss_D(pred_real, pred_fake)
        else:
            raise ValueError

    def _nsgan(self, pred_real, pred_fake, mode):
        B = pred_fake.numel()
        dev = pred_fake.device
        f32 = dict(dtype=torch.float32, device=dev)
        lib, sp = L.lib(), L.stream_ptr()
        pf = pred_fake.contiguous().float().view(-1)
        if mode == "D":
            pr = pred_real.contiguous().float().view(-1)
            dy, sc = torch.empty(2 * B, **f32), torch.empty(3, **f32)
            L.check(lib.dg_nsgan_d(L.ptr(pr), L.ptr(pf), B, 1.0, L.ptr(dy), L.ptr(dy) + 4 * B, L.ptr(sc), sp))
            return sc[2]
        dy, sc = torch.empty(B, **f32), torch.empty(1, **f32)
        L.check(lib.dg_nsgan_g(L.ptr(pf), B, 1.0, L.ptr(dy), L.ptr(sc), sp))
        return sc[0]

    def loss_D(self, pred_real, pred_fake):
        if self.metric == "nsgan":
            return self._nsgan(pred_real, pred_fake, "D")
        if self.metric in METRICS:
            raise NotImplementedError(f"gan_mode={self.metric}: only nsgan has HIP kernels (SURVEY.md §8f row 4)")
        raise NotImplementedError

    def loss_G(self, pred_real, pred_fake):
        if self.metric == "nsgan":
            return self._nsgan(pred_real, pred_fake, "G")
        if self.metric in METRICS:
            raise NotImplementedError(f"gan_mode={self.metric}: only nsgan has HIP kernels (SURVEY.md §8f row 4)")
        raise NotImplementedError
