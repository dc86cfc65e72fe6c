"""DCGAN with equalized learning rate -- reference: models/gans/dcgan_eqlr.py.

Same module tree / parameter names / parameter shapes as the reference (so checkpoints interchange),
but the parameters are strided views of one flat fp32 buffer in engine layout and `forward` launches the HIP
kernels of dusty_gan_amd.engine.  No autograd graph is built: training goes through
dusty_gan_amd.trainers.dcgan_amp.Trainer, which runs the explicit backward schedule.
"""
import torch
from torch import nn

from ... import engine as E
from ..ops import common as ops


class Proj(nn.Sequential):
    """reference: dcgan_eqlr.py:6-16"""

    def __init__(self, in_ch, out_ch, kernel=(4, 16)):
        super().__init__(
            ops.EqualLR(ops.ConvParams((in_ch, out_ch, kernel[0], kernel[1]), None, "ConvTranspose2d")),
            ops.FusedLeakyReLU(out_ch),
        )


class Up(nn.Sequential):
    """reference: dcgan_eqlr.py:19-26"""

    def __init__(self, in_ch, out_ch, ring=True):
        horizontal = "circular" if ring else "reflect"
        super().__init__(
            ops.Pad(padding=1, horizontal=horizontal, vertical="reflect"),
            ops.EqualLR(ops.ConvParams((in_ch, out_ch, 4, 4), None, "ConvTranspose2d")),
            ops.FusedLeakyReLU(out_ch),
        )


class Head(nn.Module):
    """reference: dcgan_eqlr.py:29-46"""

    def __init__(self, in_ch, out_ch={"rgb": 3}, ring=True):
        super().__init__()
        assert isinstance(out_ch, dict)
        self.in_ch = in_ch
        self.heads = nn.ModuleDict()
        horizontal = "circular" if ring else "reflect"
        for name, ch in out_ch.items():
            self.heads[name] = nn.Sequential(
                ops.Pad(padding=1, horizontal=horizontal, vertical="reflect"),
                ops.EqualLR(ops.ConvParams((in_ch, ch, 4, 4), ch, "ConvTranspose2d")),
            )


class Down(nn.Sequential):
    """reference: dcgan_eqlr.py:75-82"""

    def __init__(self, in_ch, out_ch, ring=True):
        horizontal = "circular" if ring else "reflect"
        super().__init__(
            ops.Pad(padding=1, horizontal=horizontal, vertical="reflect"),
            ops.EqualLR(ops.ConvParams((out_ch, in_ch, 4, 4), None, "Conv2d")),
            ops.FusedLeakyReLU(out_ch),
        )


class _EngineNet(nn.Sequential):
    """Common machinery: flat ParamStore + rebinding of the reference-shaped parameter views."""

    def _bind_pairs(self):  # -> list of (Parameter, view-of-flat factory)
        raise NotImplementedError

    def _rebind(self):
        with torch.no_grad():
            for p, mk in self._bind_pairs():
                p.data = mk(self.store)

    def _apply(self, fn, recurse=True):
        # parameters live in the flat store: move the store, then re-create the views (never per-parameter copies)
        self.store.apply(fn)
        self._rebind()
        for m in self.modules():
            for k, b in m._buffers.items():
                if b is not None:
                    m._buffers[k] = fn(b)
        return self

    def refresh(self):
        """Call after editing parameters in place by hand; load_state_dict does it automatically."""
        self.store._seen_version = -1

    def _post_load(self, *_):
        self.refresh()

    def set_precision(self, dtype):
        if dtype not in (torch.float32, torch.bfloat16):
            raise ValueError("compute dtype must be torch.float32 or torch.bfloat16")
        self.compute_dtype = dtype
        self._eng = None

    def _require_gpu(self, t):
        if not t.is_cuda or not self.store.flat.is_cuda:
            raise RuntimeError("dusty_gan_amd networks run on MI355X only (no CPU fallback): move the module and "
                               "its inputs to a cuda device")


class Generator(_EngineNet):
    """reference: dcgan_eqlr.py:49-72.  `masker` ('none'|'dusty1'|'dusty2') tells the fused head kernel which
    DUSty post-processing follows (models/dusty.py); the wrappers in dusty_gan_amd/models/dusty.py set it."""

    def __init__(self, in_ch, out_ch, ch_base=64, ch_max=512, shape=(64, 256), ring=True):
        shape_in = (shape[0] >> 4, shape[1] >> 4)
        ch = lambda i: min(ch_base << i, ch_max)
        super().__init__(
            Proj(in_ch, ch(3), shape_in),
            Up(ch(3), ch(2), ring),
            Up(ch(2), ch(1), ring),
            Up(ch(1), ch(0), ring),
            Head(ch(0), out_ch, ring),
        )
        out_ch = dict(out_ch)
        if list(out_ch.keys())[0] != "depth" or out_ch["depth"] != 1:
            raise NotImplementedError("the fused head expects out_ch = {'depth': 1[, 'confidence': k]}")
        k = int(out_ch.get("confidence", 0))
        if set(out_ch) - {"depth", "confidence"} or k > 2:
            raise NotImplementedError(f"unsupported generator heads {out_ch}")
        self.masker = {0: "none", 1: "dusty1", 2: "dusty2"}[k]
        self.in_ch, self.ring, self.shape = in_ch, ring, (int(shape[0]), int(shape[1]))
        self.chs = [ch(0), ch(1), ch(2), ch(3)]
        self.tau, self.drop_const = 1.0, -1.0
        self.compute_dtype = torch.float32
        self._eng = None
        self.store = E.ParamStore(E.g_segments(in_ch, self.chs, self.shape, 1 + k))
        with torch.no_grad():  # EqualLR init (reference: models/ops/common.py:128-130): W ~ N(0,1), biases 0
            for s in self.store.seg.values():
                if s.kind != "bias":
                    self.store.view(s.name).normal_(0.0, 1.0)
        self._rebind()
        self.register_load_state_dict_post_hook(self._post_load)

    def _bind_pairs(self):
        pairs = [(self[0][0].module.weight, lambda st: st.view("proj_w").permute(3, 2, 0, 1)),
                 (self[0][1].bias, lambda st: st.view("proj_b"))]
        for i in (1, 2, 3):
            pairs.append((self[i][1].module.weight, lambda st, i=i: st.view(f"up{i}_w").permute(2, 3, 0, 1)))
            pairs.append((self[i][2].bias, lambda st, i=i: st.view(f"up{i}_b")))
        hd = self[4].heads
        pairs.append((hd["depth"][1].module.weight, lambda st: st.view("head_w")[..., 0:1].permute(2, 3, 0, 1)))
        pairs.append((hd["depth"][1].module.bias, lambda st: st.view("head_b")[0:1]))
        if "confidence" in hd:
            pairs.append((hd["confidence"][1].module.weight, lambda st: st.view("head_w")[..., 1:].permute(2, 3, 0, 1)))
            pairs.append((hd["confidence"][1].module.bias, lambda st: st.view("head_b")[1:]))
        return pairs

    def engine(self):
        x3 = bool(getattr(self, "fp32_split", False))   # (set by the Trainer on ITS networks: fp32x3 parity mode)
        x2 = bool(getattr(self, "fp32_pairs", False))   # (... with split-bf16 storage of the fat feature maps: DG_BF16X2)
        if (self._eng is None or self._eng.dtype != self.compute_dtype
                or self._eng.ops.x3 != (x3 and self.compute_dtype == torch.float32) or self._eng.x2_asked != x2):
            cfg = E.NetCfg(self.shape, self.in_ch, self.chs, self.masker, self.ring, self.tau, self.drop_const)
            self._eng = E.GEngine(cfg, self.compute_dtype, x3=x3, x2=x2)
        self._eng.cfg.tau, self._eng.cfg.drop_const = float(self.tau), float(self.drop_const)
        return self._eng

    def run(self, latent, noise=None, training=True):
        self._require_gpu(latent)
        return self.engine().forward(self.store, latent, noise, training)

    def forward(self, latent):
        if self.masker != "none":
            raise RuntimeError("a generator with a confidence head is driven by its DUSty1/DUSty2 wrapper")
        return self.run(latent)


class Discriminator(_EngineNet):
    """reference: dcgan_eqlr.py:85-96"""

    def __init__(self, in_ch, ch_base=64, ch_max=512, shape=(64, 256), ring=True):
        shape_out = (shape[0] >> 4, shape[1] >> 4)
        ch = lambda i: min(ch_base << i, ch_max)
        super().__init__(
            ops.BlurVH(ring),
            Down(in_ch * 2, ch(0), ring),
            Down(ch(0), ch(1), ring),
            Down(ch(1), ch(2), ring),
            Down(ch(2), ch(3), ring),
            ops.EqualLR(ops.ConvParams((1, ch(3), shape_out[0], shape_out[1]), 1, "Conv2d")),
        )
        self.in_ch, self.ring, self.shape = in_ch, ring, (int(shape[0]), int(shape[1]))
        self.chs = [ch(0), ch(1), ch(2), ch(3)]
        self.compute_dtype = torch.float32
        self._eng = None
        self.store = E.ParamStore(E.d_segments(in_ch, self.chs, self.shape))
        with torch.no_grad():
            for s in self.store.seg.values():
                if s.kind != "bias":
                    self.store.view(s.name).normal_(0.0, 1.0)
        self._rebind()
        self.register_load_state_dict_post_hook(self._post_load)

    def _bind_pairs(self):
        pairs = []
        for i in (1, 2, 3, 4):
            pairs.append((self[i][1].module.weight, lambda st, i=i: st.view(f"d{i}_w").permute(3, 2, 0, 1)))
            pairs.append((self[i][2].bias, lambda st, i=i: st.view(f"d{i}_b")))
        pairs.append((self[5].module.weight, lambda st: st.view("final_w").permute(2, 0, 1).unsqueeze(0)))
        pairs.append((self[5].module.bias, lambda st: st.view("final_b")))
        return pairs

    def engine(self):
        x3 = bool(getattr(self, "fp32_split", False))
        x2 = bool(getattr(self, "fp32_pairs", False))
        if (self._eng is None or self._eng.dtype != self.compute_dtype
                or self._eng.ops.x3 != (x3 and self.compute_dtype == torch.float32) or self._eng.x2_asked != x2):
            cfg = E.NetCfg(self.shape, 1, self.chs, "none", self.ring, dis_in_ch=self.in_ch)
            self._eng = E.DEngine(cfg, self.compute_dtype, x3=x3, x2=x2)
        return self._eng

    def forward(self, x):
        self._require_gpu(x)
        eng = self.engine()
        x = x.contiguous().float()
        eng.alloc(x.shape[0], x.device)
        return eng.forward(self.store, x, 0).clone().view(-1, 1, 1, 1)
