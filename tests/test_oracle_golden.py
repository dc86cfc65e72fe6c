"""The oracle (oracle/dusty_oracle.py) against the vectors generated from the reference's own modules."""
import numpy as np
import pytest
import torch

from oracle import dusty_oracle as O
from tests.golden_util import STEP_CASES, load, rel_l2, step_rand, sub

TOL = 2e-5  # fp32 CPU vs fp32 CPU, same library: rounding-order differences only


@pytest.fixture(scope="module")
def ops():
    return load("ops")


def t(a):
    return torch.from_numpy(np.array(a))


@pytest.mark.parametrize("ring", [True, False])
def test_pad_and_blur(ops, ring):
    r = f"ring{int(ring)}"
    assert torch.equal(O.pad_ring(t(ops[f"pad/{r}/x"]), ring), t(ops[f"pad/{r}/y"]))
    assert rel_l2(O.blur_vh(t(ops[f"blurvh/{r}/x"]), ring), ops[f"blurvh/{r}/y"]) < 1e-6


@pytest.mark.parametrize("ring", [True, False])
@pytest.mark.parametrize("kind", ["up", "down"])
def test_up_down_fwd_bwd(ops, kind, ring):
    tag = f"{kind}/ring{int(ring)}"
    x = t(ops[f"{tag}/x"]).requires_grad_()
    w = t(ops[f"{tag}/param/1.module.weight"]).requires_grad_()
    b = t(ops[f"{tag}/param/2.bias"]).requires_grad_()
    y = (O.up if kind == "up" else O.down)(x, w, b, ring)
    assert rel_l2(y, ops[f"{tag}/y"]) < TOL
    gx, gw, gb = torch.autograd.grad(y, [x, w, b], t(ops[f"{tag}/gy"]))
    assert rel_l2(gx, ops[f"{tag}/gx"]) < TOL
    assert rel_l2(gw, ops[f"{tag}/grad/1.module.weight"]) < TOL
    assert rel_l2(gb, ops[f"{tag}/grad/2.bias"]) < TOL


def test_proj_and_head(ops):
    y = O.proj(t(ops["proj/x"]), t(ops["proj/param/0.module.weight"]), t(ops["proj/param/1.bias"]))
    assert rel_l2(y, ops["proj/y"]) < TOL
    x = t(ops["head/x"])
    for name in ("depth", "confidence"):
        y = O.head(x, t(ops[f"head/param/heads.{name}.1.module.weight"]), t(ops[f"head/param/heads.{name}.1.module.bias"]))
        assert rel_l2(y, ops[f"head/y/{name}"]) < TOL


@pytest.mark.parametrize("fname", ["brightness", "saturation", "contrast", "translation", "cutout"])
def test_augment_functions(ops, fname):
    rp = sub(ops, f"aug/{fname}")
    y = O.diff_augment(t(ops["aug/x"]), rp, policy=(fname,))
    assert rel_l2(y, ops[f"aug/{fname}/y"]) < 1e-6


def test_gumbel_sigmoid(ops):
    lg = t(ops["gumbel/logits"]).requires_grad_()
    y = O.gumbel_sigmoid(lg, t(ops["gumbel/noise"]), 1.0)
    assert torch.equal(y.detach(), t(ops["gumbel/y"]))
    assert set(np.unique(y.detach().numpy())) <= {0.0, 1.0}
    (g,) = torch.autograd.grad(y, lg, t(ops["gumbel/gy"]))
    assert rel_l2(g, ops["gumbel/glogits"]) < 1e-6


@pytest.mark.parametrize("metric", ["nsgan", "wgan", "lsgan", "hinge", "ragan", "rahinge", "ralsgan"])
def test_gan_loss(ops, metric):
    pr, pf = t(ops["ganloss/pred_real"]), t(ops["ganloss/pred_fake"])
    for mode in ("D", "G"):
        assert abs(float(O.gan_loss(metric, pr, pf, mode)) - float(ops[f"ganloss/{metric}/{mode}"])) < 1e-6
    with pytest.raises(NotImplementedError):
        O.gan_loss("nope", pr, pf, "D")
    with pytest.raises(ValueError):
        O.gan_loss("nsgan", pr, pf, "X")


def test_invert_depth(ops):
    pol = t(ops["invert_depth/pol"])
    inv, _ = O.fetch_reals(pol, torch.ones_like(pol))
    assert rel_l2((inv + 1) / 2, ops["invert_depth/inv"]) < 1e-6


@pytest.mark.parametrize("case", STEP_CASES)
def test_train_step_matches_reference(case):
    g = load("step_" + case)
    arch, ring = str(g["meta/arch"]), bool(g["meta/ring"])
    cfg = O.StepConfig(arch=arch, ring=ring, gan_mode=str(g["meta/gan_mode"]), w_gp=float(g["meta/gp"]),
                       lr_g=float(g["meta/lr"]), lr_d=float(g["meta/lr"]), beta1=float(g["meta/beta1"]),
                       beta2=float(g["meta/beta2"]), ema_decay=float(g["meta/ema_decay"]))
    G, D = sub(g, "init/G"), sub(g, "init/D")
    D = {k: v for k, v in D.items() if not k.endswith("kernel")}  # BlurVH buffers are constants
    G_ema = {k: v.clone() for k, v in G.items()}
    oG, oD = O.new_optim_state(G), O.new_optim_state(D)
    for it in range(int(g["meta/steps"])):
        pre = f"s{it}"
        pol, mask = t(g[f"{pre}/pol"]), t(g[f"{pre}/mask"])
        x_real, _ = O.fetch_reals(pol, mask)
        assert rel_l2(x_real, g[f"{pre}/x_real"]) < 1e-6
        sc, ex = O.train_step(G, D, G_ema, oG, oD, it + 1, cfg, x_real, step_rand(g, it), return_grads=True)
        for k, v in sub(g, f"{pre}/scalar").items():
            assert abs(sc[k] - float(v)) <= 1e-4 * max(1.0, abs(float(v))), (k, sc[k], float(v))
        for k, v in sub(g, f"{pre}/synth").items():
            if k == "mask":
                assert torch.equal(ex["synth"][k], v)
            else:
                assert rel_l2(ex["synth"][k], v) < TOL, k
        assert rel_l2(ex["x_real_aug"], g[f"{pre}/x_real_aug"]) < TOL
        assert rel_l2(ex["x_fake_aug"], g[f"{pre}/x_fake_aug"]) < TOL
        assert rel_l2(ex["y_real"], g[f"{pre}/y_real"]) < TOL
        assert rel_l2(ex["y_fake2"], g[f"{pre}/y_fake2"]) < 1e-4
        if cfg.w_gp > 0:
            assert rel_l2(ex["r1_grads"], g[f"{pre}/r1_grads"]) < TOL
        for k, v in sub(g, f"{pre}/grad_D").items():
            assert rel_l2(ex["grad_D"][k], v) < 1e-4, k
        for k, v in sub(g, f"{pre}/grad_G").items():
            assert rel_l2(ex["grad_G"][k], v) < 1e-4, k
        for tag, cur in (("G", G), ("D", D), ("G_ema", G_ema)):
            for k, v in sub(g, f"{pre}/after/{tag}").items():
                if k.endswith("kernel"):
                    continue  # BlurVH buffers, not parameters
                assert rel_l2(cur[k], v) < 1e-4, (tag, k)
    for tag, opt in (("G", oG), ("D", oD)):
        for k in opt:
            assert rel_l2(opt[k]["v"], g[f"final/optim_{tag}/{k}/exp_avg_sq"]) < 1e-4
            assert rel_l2(opt[k]["m"], g[f"final/optim_{tag}/{k}/exp_avg"]) < 1e-4
