"""Parameter-holding mirrors of the reference's models/ops/common.py building blocks.

Only the module TREE (names, parameter shapes, extra_repr) mirrors the reference so that state_dict keys match
(SURVEY.md §8b).  The arithmetic of Pad / Blur / FusedLeakyReLU / EqualLR is fused into the HIP kernels
(dusty_gan_amd/csrc); these leaves are never called on their own.
"""
import math

import torch
import torch.nn as nn


class _Fused(nn.Module):
    def forward(self, *a, **k):
        raise RuntimeError(f"{type(self).__name__} is fused into the HIP kernels of its parent network; "
                           "call the Generator / Discriminator instead")


class Pad(_Fused):
    """reference: models/ops/common.py:9-23 (index arithmetic in the conv tile loaders here)."""

    def __init__(self, padding, horizontal="constant", vertical="constant"):
        super().__init__()
        self.padding = (padding,) * 4 if isinstance(padding, int) else tuple(padding)
        self.horizontal, self.vertical = horizontal, vertical

    def extra_repr(self):
        return f"padding={self.padding}, horizontal={self.horizontal}, vertical={self.vertical}"


class Blur(_Fused):
    """reference: models/ops/common.py:26-71; holds the (constant) `kernel` buffer for state_dict parity."""

    def __init__(self, filter_type, direction, ring=True):
        super().__init__()
        k = torch.tensor(filter_type, dtype=torch.float32)
        k = k[:, None] if direction == "v" else k[None, :]
        k = k / k.sum()
        self.register_buffer("kernel", k[None, None])
        self.direction, self.ring = direction, ring


class BlurVH(_Fused):
    """reference: models/ops/common.py:74-88."""

    def __init__(self, ring=True):
        super().__init__()
        self.blur_v = Blur([1, 2, 1], "v", ring)
        self.blur_h = Blur([1, 2, 1], "h", ring)


class FusedLeakyReLU(_Fused):
    """reference: models/ops/common.py:91-109; `bias` is a view into the engine's flat store."""

    def __init__(self, ch, negative_slope=0.2, scale=math.sqrt(2)):
        super().__init__()
        self.ch, self.negative_slope, self.gain = ch, negative_slope, scale
        self.bias = nn.Parameter(torch.zeros(ch))

    def extra_repr(self):
        return f"ch={self.ch}, negative_slope={self.negative_slope}, gain={self.gain}"


class ConvParams(_Fused):
    """Stands where the reference has nn.Conv2d / nn.ConvTranspose2d inside EqualLR: holds `weight` (reference
    shape, strided view of the engine's [ky][kx][ci][co] storage) and optionally `bias`."""

    def __init__(self, weight_shape, bias_ch=None, kind="Conv2d"):
        super().__init__()
        self.kind = kind
        self.weight = nn.Parameter(torch.zeros(weight_shape))
        if bias_ch is not None:
            self.bias = nn.Parameter(torch.zeros(bias_ch))
        else:
            self.register_parameter("bias", None)

    def extra_repr(self):
        return f"{self.kind}, weight={tuple(self.weight.shape)}, bias={self.bias is not None}"


class EqualLR(_Fused):
    """reference: models/ops/common.py:112-136.  scale = 1/sqrt(weight[0].numel()) is applied in the kernels'
    epilogues; weights are N(0,1), biases 0 at init (:128-130)."""

    def __init__(self, module, gain: float = 1.0):
        super().__init__()
        self.module = module
        self.gain = gain
        self.scale = 1.0 / math.sqrt(self.module.weight[0].numel())

    def extra_repr(self):
        return f"gain={self.gain}"
