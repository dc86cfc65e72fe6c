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


# ---------------------------------------------------------------- full-width pin (tests/golden/full_dusty2.npz)
# At 64x1024 / 512 channels the parameters alone are 280 MB, so the fixture holds DIGESTS of what the reference computed
# (three float64 statistics + a strided sample of <= 2048 elements per tensor) and the seeds from which both sides
# regenerate bit-identical parameters and inputs with torch's CPU generator (torch version recorded in the file).
FULL_SAMPLE = 2048


def digest(t):
    """(stats [sum, sum |x|, l2 norm] float64, strided sample float32) of a tensor"""
    f = torch.as_tensor(t).detach().flatten()
    d = f.to(torch.float64)
    stride = max(1, f.numel() // FULL_SAMPLE)
    return (np.array([float(d.sum()), float(d.abs().sum()), float(d.norm())], dtype=np.float64),
            f[::stride][:FULL_SAMPLE].to(torch.float32).numpy().copy())


def full_inputs(arch, in_ch, ch_base, ch_max, shape, B, seed):
    """parameters (the reference's state_dict layout, N(0,1) / zero bias = EqualLR's init, models/ops/common.py:128-130),
    one batch, the latent and the Gumbel noise of the full-width case, from torch's CPU generator seeded `seed`"""
    from oracle import dusty_oracle as O
    gen = torch.Generator().manual_seed(int(seed))
    H, W = shape
    G = O.init_G(f"{arch}/dcgan_eqlr", in_ch, ch_base, ch_max, shape, gen)
    D = O.init_D(1, ch_base, ch_max, shape, gen)
    pol = torch.rand(B, 1, H, W, generator=gen)
    mask = torch.rand(B, 1, H, W, generator=gen) > 0.15
    pol = pol * mask
    z = torch.randn(B, in_ch, generator=gen)
    u = [torch.rand(B, 1, H, W, generator=gen) for _ in range(2)] + [torch.rand(B, 1, 1, 1, generator=gen) for _ in range(2)]
    return G, D, pol, mask, z, u


def sample_dev(npz, key, t):
    """relative L2 distance of tensor `t`'s strided sample from the one stored under `key` (what `check_digest` bounds)"""
    _, sample = digest(t)
    return rel_l2(sample, npz[f"{key}/sample"])


def check_digest(npz, key, t, tol, what="", cos=None):
    """hold tensor `t` to the digest stored under `key`: l2 norm and sum |x| within `tol` relative, the sum within `tol` of
    sum |x|, the strided sample within `tol` relative L2 (and, with `cos`, at least that cosine to it)"""
    stats, sample = digest(t)
    ref_stats, ref_sample = npz[f"{key}/stats"], npz[f"{key}/sample"]
    scale = max(float(ref_stats[1]), 1e-30)
    assert abs(stats[0] - ref_stats[0]) <= tol * scale, (what, key, "sum", stats[0], ref_stats[0])
    assert abs(stats[1] - ref_stats[1]) <= tol * scale, (what, key, "abs sum", stats[1], ref_stats[1])
    assert abs(stats[2] - ref_stats[2]) <= tol * max(float(ref_stats[2]), 1e-30), (what, key, "norm", stats[2], ref_stats[2])
    err = rel_l2(sample, ref_sample)
    assert err <= tol, (what, key, "sample", err)
    if cos is not None:
        a, b = np.asarray(sample, dtype=np.float64).ravel(), np.asarray(ref_sample, dtype=np.float64).ravel()
        c = float((a * b).sum() / max(np.linalg.norm(a) * np.linalg.norm(b), 1e-300))
        assert c >= cos, (what, key, "cosine", c)


def full_case(npz):
    """regenerated inputs of the full-width fixture: (G, D, x-side tensors, rand bundle)"""
    from oracle import dusty_oracle as O
    arch, shape = str(npz["meta/arch"]), tuple(int(v) for v in npz["meta/shape"])
    G, D, pol, mask, z, u = full_inputs(arch, int(npz["meta/in_ch"]), int(npz["meta/ch_base"]), int(npz["meta/ch_max"]),
                                        shape, int(npz["meta/B"]), int(npz["meta/seed"]))
    rand = {"z": z, "noise": {"pixel": O.logistic_noise(u[0], u[1]), "image": O.logistic_noise(u[2], u[3])},
            "aug": [sub(npz, f"aug{j}") for j in range(4)]}
    return G, D, pol, mask, rand
