"""DUSty measurability wrappers -- reference: models/dusty.py.

The Gumbel-sigmoid sampling and the mask-out are fused into the generator's head kernels
(csrc/pointwise.hip head_post_fwd/bwd); these classes keep the reference's module names, buffers and the
`GumbelSigmoid.fixed_noise` / `logistic_noise` hooks that utils.setup relies on (utils/__init__.py:141-149).
"""
import torch
from torch import nn

from ..utils.rng import Philox


class GumbelSigmoid(nn.Module):
    """reference: models/dusty.py:6-62 (tau fixed, hard=True).  Holds the noise policy; the arithmetic is fused."""

    def __init__(self, tau: float = 1.0, tau_max: float = 1.0, hard: bool = True, eps: float = 1e-10,
                 pixelwise: bool = True):
        super().__init__()
        if tau is None or not hard:
            raise NotImplementedError("learnable tau / soft masks are not on the training path of the reference "
                                      "configs (configs/model/dusty*_dcgan_eqlr.yaml: tau: 1)")
        self.tau, self.tau_max, self.hard, self.eps, self.pixelwise = tau, tau_max, hard, eps, pixelwise
        self.fixed_noise = None
        self.fix_on_first_use = False  # utils.setup(fix_noise=True): what the reference's forward-pre-hook does
        self._rng = None

    def rng(self, device):
        if self._rng is None or self._rng.device != device:
            self._rng = Philox(torch.initial_seed() + (17 if self.pixelwise else 29), device, stream_id=3)
        return self._rng

    def logistic_noise(self, logits):
        B, _, H, W = logits.shape
        shape = (B, 1, H, W) if self.pixelwise else (B, 1, 1, 1)
        return self.rng(logits.device).logistic_noise(shape, self.eps)

    def noise_for(self, B, H, W, device):
        if self.fixed_noise is not None:
            shape = (-1, -1, -1) if self.pixelwise else (-1, -1, -1)
            return self.fixed_noise.to(device).expand(B, *shape).contiguous()
        shape = (B, 1, H, W) if self.pixelwise else (B, 1, 1, 1)
        noise = self.rng(device).logistic_noise(shape, self.eps)
        if self.fix_on_first_use:
            # utils/__init__.py:141-149: `m.fixed_noise = m.logistic_noise(i[0])[[0]]` on the first call, reused after
            self.fixed_noise = noise[[0]].clone()
            return self.fixed_noise.expand(B, -1, -1, -1).contiguous()
        return noise

    def forward(self, logits, threshold: float = 0.5):
        raise RuntimeError("GumbelSigmoid is fused into the generator head kernels; call the DUSty wrapper")

    def extra_repr(self):
        return f"hard={self.hard}, eps={self.eps}"


class _DUSty(nn.Module):
    def __init__(self, backbone, tau, drop_const=-1):
        super().__init__()
        self.backbone = backbone
        self.register_buffer("drop_const", torch.tensor(drop_const).float())
        self._tau = float(tau)
        self._drop = float(drop_const)

    def _sync(self, latent=None):
        if latent is not None:
            self.backbone._require_gpu(latent)
        self.backbone.tau = self._tau
        self.backbone.drop_const = self._drop

    def set_precision(self, dtype):
        self.backbone.set_precision(dtype)

    @property
    def store(self):
        return self.backbone.store


class DUSty1(_DUSty):
    """reference: models/dusty.py:65-91"""

    def __init__(self, backbone, tau, drop_const=-1):
        super().__init__(backbone, tau, drop_const)
        self.gumbel = GumbelSigmoid(hard=True, tau=tau, pixelwise=True)
        assert backbone.masker == "dusty1", "DUSty1 needs a generator with a 1-channel confidence head"

    def forward(self, latent, noise=None, **kwargs):
        self._sync(latent)
        H, W = self.backbone.shape
        if noise is None:
            noise = {"pixel": self.gumbel.noise_for(latent.shape[0], H, W, latent.device)}
        return self.backbone.run(latent, noise, self.training)


class DUSty2(_DUSty):
    """reference: models/dusty.py:94-127 (eval mode thresholds the image-level logit, still samples pixel noise)"""

    def __init__(self, backbone, tau, drop_const=-1):
        super().__init__(backbone, tau, drop_const)
        self.gumbel_pixel = GumbelSigmoid(hard=True, tau=tau, pixelwise=True)
        self.gumbel_image = GumbelSigmoid(hard=True, tau=tau, pixelwise=False)
        assert backbone.masker == "dusty2", "DUSty2 needs a generator with a 2-channel confidence head"

    def forward(self, latent, noise=None, **kwargs):
        self._sync(latent)
        H, W = self.backbone.shape
        B = latent.shape[0]
        if noise is None:
            noise = {"pixel": self.gumbel_pixel.noise_for(B, H, W, latent.device)}
            if self.training:
                noise["image"] = self.gumbel_image.noise_for(B, H, W, latent.device)
        return self.backbone.run(latent, noise, self.training)
