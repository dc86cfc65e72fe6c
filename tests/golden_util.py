"""Helpers to read the committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py)."""
import os
from collections import OrderedDict

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

STEP_CASES = ["none_ring", "dusty1_ring", "dusty2_ring", "dusty2_noring", "dusty2_nogp", "dusty2_mid",
              # one per remaining `solver.gan_mode` (models/loss.py:42-61,70-85)
              "none_wgan", "none_lsgan", "dusty1_hinge", "dusty2_ragan", "dusty1_rahinge", "dusty2_ralsgan"]
# path-length regularisation on (trainers/dcgan_amp.py:268-306), two steps each
PL_CASES = ["none_pl", "dusty1_pl", "dusty2_pl"]


def load(name):
    return np.load(os.path.join(GOLDEN_DIR, name + ".npz"))


def sub(npz, prefix, as_torch=True):
    """All entries under `prefix/` as an ordered dict keyed by the remainder."""
    out = OrderedDict()
    p = prefix.rstrip("/") + "/"
    for k in npz.files:
        if k.startswith(p):
            v = npz[k]
            out[k[len(p):]] = torch.from_numpy(np.array(v)) if as_torch else v
    return out


def rel_l2(a, b):
    a = torch.as_tensor(a).detach().to(torch.float64).flatten()
    b = torch.as_tensor(b).detach().to(torch.float64).flatten()
    den = float(b.norm())
    if den == 0.0:
        return float((a - b).norm())
    return float((a - b).norm() / den)


def step_rand(npz, it):
    pre = f"s{it}"
    rand = {"z": torch.from_numpy(npz[f"{pre}/z"]), "noise": sub(npz, f"{pre}/noise"),
            "aug": [sub(npz, f"{pre}/aug{j}") for j in range(4)]}
    if f"{pre}/pl/z" in npz.files:
        rand["pl"] = {"z": torch.from_numpy(npz[f"{pre}/pl/z"]), "noise": sub(npz, f"{pre}/pl/noise"),
                      "y": torch.from_numpy(npz[f"{pre}/pl/y"]), "pl_ema": torch.from_numpy(npz[f"{pre}/pl/pl_ema"])}
    return rand


def meta_pl(npz):
    return float(npz["meta/pl"]) if "meta/pl" in npz.files else 0.0
