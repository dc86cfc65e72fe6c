"""Config loading for the reference's Hydra tree (configs/config.yaml + dataset/ model/ solver/ groups).

hydra-core / omegaconf are optional: when they are missing (this image) a PyYAML loader composes the same tree and
applies Hydra-style overrides (`group=name`, `a.b.c=value`).  SURVEY.md §0.6: the reference's default `solver: nsgan`
does not exist; `nsgan_eqlr` is what every documented command passes, so it is the default here.
"""
import os

import yaml

CONFIG_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "configs")
GROUPS = ("dataset", "model", "solver")


class Cfg(dict):
    """dict with attribute access (enough of OmegaConf's surface for the trainer)."""

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v

    @staticmethod
    def wrap(o):
        if isinstance(o, dict):
            return Cfg({k: Cfg.wrap(v) for k, v in o.items()})
        if isinstance(o, list):
            return [Cfg.wrap(v) for v in o]
        return o


def _set(cfg, dotted, value):
    keys = dotted.split(".")
    cur = cfg
    for k in keys[:-1]:
        if k not in cur or not isinstance(cur[k], dict):
            cur[k] = Cfg()
        cur = cur[k]
    cur[keys[-1]] = value


def load_config(overrides=(), config_dir=CONFIG_DIR):
    with open(os.path.join(config_dir, "config.yaml")) as f:
        root = yaml.safe_load(f)
    choices = {}
    for d in root.pop("defaults", []):
        if isinstance(d, dict):
            choices.update(d)
    root.pop("hydra", None)
    plain = []
    for ov in overrides:
        k, _, v = ov.partition("=")
        if k in GROUPS:
            choices[k] = v
        else:
            plain.append((k, yaml.safe_load(v)))
    for g in GROUPS:
        path = os.path.join(config_dir, g, f"{choices[g]}.yaml")
        if not os.path.exists(path):
            raise FileNotFoundError(f"config group '{g}' has no option '{choices[g]}' ({path})")
        with open(path) as f:
            root[g] = yaml.safe_load(f)
    cfg = Cfg.wrap(root)
    for k, v in plain:
        _set(cfg, k, Cfg.wrap(v))
    # torch >= 2 rejects int betas (`beta1: 0`)
    cfg.solver.lr.beta1 = float(cfg.solver.lr.beta1)
    cfg.solver.lr.beta2 = float(cfg.solver.lr.beta2)
    return cfg


def load_config_file(path):
    """A fully composed config (what Hydra dumps to .hydra/config.yaml and utils.setup reads, utils/__init__.py:122)."""
    with open(path) as f:
        cfg = Cfg.wrap(yaml.safe_load(f))
    cfg.solver.lr.beta1 = float(cfg.solver.lr.beta1)
    cfg.solver.lr.beta2 = float(cfg.solver.lr.beta2)
    return cfg


def dump_config(cfg, path):
    """the inverse: write the composed tree as plain YAML"""
    def plain(o):
        if isinstance(o, dict):
            return {k: plain(v) for k, v in o.items()}
        if isinstance(o, (list, tuple)):
            return [plain(v) for v in o]
        return o
    with open(path, "w") as f:
        yaml.safe_dump(plain(cfg), f, sort_keys=False)  # key order matters: out_ch = {depth, confidence}
