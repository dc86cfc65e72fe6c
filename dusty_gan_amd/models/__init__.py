"""Model registry: `define_G(cfg)` / `define_D(cfg)` with the reference's arch strings ("<masker>/<backbone>" for the
generator, "<backbone>" for the discriminator) and its NotImplementedError for anything unknown -- reference:
models/__init__.py:5-50.  The registries are tables: an entry names the module class and the config keys it is built from
(the reference's keyword names, which are the constructor contract of models/gans/dcgan_eqlr.py and models/dusty.py)."""
import torch

from . import dusty
from .gans import dcgan_eqlr

# backbone name -> (class, {constructor keyword: config key under cfg.model.gen / cfg.model.dis})
GENERATORS = {
    "dcgan_eqlr": (dcgan_eqlr.Generator, {"in_ch": "in_ch", "out_ch": "out_ch", "ch_base": "ch_base", "ch_max": "ch_max",
                                          "shape": "shape"}),
}
DISCRIMINATORS = {
    "dcgan_eqlr": (dcgan_eqlr.Discriminator, {"in_ch": "in_ch", "ch_base": "ch_base", "ch_max": "ch_max", "shape": "shape"}),
}
# masker name -> wrapper class taking (backbone, tau, drop_const); None = the bare backbone
MASKERS = {"none": None, "dusty1": dusty.DUSty1, "dusty2": dusty.DUSty2}


def _get(cfg, key, default=None):
    try:
        return cfg[key] if not hasattr(cfg, key) else getattr(cfg, key)
    except (KeyError, AttributeError, TypeError):
        return default


def _precision(cfg):
    """bf16 when the config asks for AMP (reference: enable_amp -> torch.cuda.amp, trainers/dcgan_amp.py:128-129;
    bf16 needs no loss scaling), else fp32."""
    return torch.bfloat16 if _get(cfg, "enable_amp", False) else torch.float32


def _build(table, name, node, ring):
    entry = table.get(str(name).lower())
    if entry is None:
        raise NotImplementedError(name)
    cls, keys = entry
    kwargs = {kw: (dict(node[key]) if kw == "out_ch" else node[key]) for kw, key in keys.items()}
    return cls(ring=ring, **kwargs)


def define_G(cfg):
    masker_type, backbone_type = cfg.model.gen.arch.split("/")
    G = _build(GENERATORS, backbone_type, cfg.model.gen, cfg.model.ring)
    G.set_precision(_precision(cfg))
    if masker_type not in MASKERS:
        raise NotImplementedError(masker_type)
    wrapper = MASKERS[masker_type]
    if wrapper is None:
        if G.masker != "none":
            raise NotImplementedError("arch 'none/...' with a confidence head")
        return G
    return wrapper(backbone=G, tau=cfg.model.gen.tau, drop_const=cfg.model.gen.drop_const)


def define_D(cfg):
    D = _build(DISCRIMINATORS, cfg.model.dis.arch, cfg.model.dis, cfg.model.ring)
    D.set_precision(_precision(cfg))
    return D
