"""Model registry -- reference: models/__init__.py:5-50 (same arch strings, same NotImplementedError behaviour)."""
import torch

from . import dusty
from .gans import dcgan_eqlr


def _get(cfg, key, default=None):
    try:
        return cfg[key] if not hasattr(cfg, key) else getattr(cfg, key)
    except (KeyError, AttributeError, TypeError):
        return default


def _precision(cfg):
    """bf16 when the config asks for AMP (reference: enable_amp -> torch.cuda.amp, trainers/dcgan_amp.py:128-129;
    bf16 needs no loss scaling), else fp32."""
    return torch.bfloat16 if _get(cfg, "enable_amp", False) else torch.float32


def define_G(cfg):
    masker_type, backbone_type = cfg.model.gen.arch.split("/")

    if backbone_type.lower() == "dcgan_eqlr":
        G = dcgan_eqlr.Generator(
            in_ch=cfg.model.gen.in_ch,
            out_ch=dict(cfg.model.gen.out_ch),
            ch_base=cfg.model.gen.ch_base,
            ch_max=cfg.model.gen.ch_max,
            shape=cfg.model.gen.shape,
            ring=cfg.model.ring,
        )
    else:
        raise NotImplementedError

    G.set_precision(_precision(cfg))
    if masker_type == "dusty1":
        G = dusty.DUSty1(backbone=G, tau=cfg.model.gen.tau, drop_const=cfg.model.gen.drop_const)
    elif masker_type == "dusty2":
        G = dusty.DUSty2(backbone=G, tau=cfg.model.gen.tau, drop_const=cfg.model.gen.drop_const)
    elif masker_type == "none":
        if G.masker != "none":
            raise NotImplementedError("arch 'none/...' with a confidence head")
    else:
        raise NotImplementedError
    return G


def define_D(cfg):
    if cfg.model.dis.arch.lower() == "dcgan_eqlr":
        D = dcgan_eqlr.Discriminator(
            in_ch=cfg.model.dis.in_ch,
            ch_base=cfg.model.dis.ch_base,
            ch_max=cfg.model.dis.ch_max,
            shape=cfg.model.dis.shape,
            ring=cfg.model.ring,
        )
    else:
        raise NotImplementedError
    D.set_precision(_precision(cfg))
    return D
