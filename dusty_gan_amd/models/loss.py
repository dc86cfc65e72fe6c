"""GAN losses -- reference: models/loss.py:21-88.

On the hot path the loss and its gradient w.r.t. the logits are one tiny kernel (csrc/pointwise.hip
gan_step_kernel, all seven metrics of the reference) called by the trainer; this class keeps the reference's
dispatch surface (metric names, ValueError / NotImplementedError behaviour) and evaluates a loss VALUE for callers
that hold logits.
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

    @property
    def code(self):
        """DG_GAN_* of include/dusty_gan_hip.h (= position in models/loss.py's if-chain)"""
        if self.metric not in METRICS:
            raise NotImplementedError
        return METRICS.index(self.metric)

    @property
    def relativistic(self):
        """loss_G reads pred_real only for these (models/loss.py:76-85)"""
        return self.metric in ("ragan", "rahinge", "ralsgan")

    def _eval(self, pred_real, pred_fake, mode):
        code = self.code
        B = pred_fake.numel()
        f32 = dict(dtype=torch.float32, device=pred_fake.device)
        lib, sp = L.lib(), L.stream_ptr()
        pf = pred_fake.contiguous().float().view(-1)
        pr = pred_real.contiguous().float().view(-1) if pred_real is not None else None
        acc = torch.zeros(3, **f32)
        if mode == "D":
            dy = torch.empty(2 * B, **f32)
            L.check(lib.dg_gan_d_step(code, float(self.smoothing), L.ptr(pr), L.ptr(pf), B, 1.0, L.ptr(dy), None, None,
                                      L.ptr(acc), None, sp), "dg_gan_d_step")
            return acc[2]
        dy = torch.empty(B, **f32)
        L.check(lib.dg_gan_g_step(code, L.ptr(pr) if self.relativistic else None, L.ptr(pf), B, 1.0, L.ptr(dy),
                                  L.ptr(acc), sp), "dg_gan_g_step")
        return acc[0]

    def loss_D(self, pred_real, pred_fake):
        return self._eval(pred_real, pred_fake, "D")

    def loss_G(self, pred_real, pred_fake):
        return self._eval(pred_real, pred_fake, "G")
