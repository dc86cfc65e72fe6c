"""CPU oracle for the dusty-gan training hot path.  TEST INFRASTRUCTURE ONLY.

This file is a plain-PyTorch (CPU, fp32) restatement of the reference's
algorithm for one `Trainer.step` and the modules under it.  It exists to CHECK
the HIP path: only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline`
leg of `bench.py` may import it.  Nothing under `dusty_gan_amd/` imports it;
the product path has no CPU fallback.

Parity pin: the reference ships no tests or golden vectors, so
the oracle is pinned against outputs of the reference's own modules imported
in the build container -- see `tests/golden/make_golden.py` (the generator,
which imports /root/reference) and `tests/test_oracle_golden.py` (which checks
this file against the committed vectors without the reference present).

Every function cites the reference file:line it follows (paths relative to the
reference repository root).  Randomness is always INJECTED (the reference never
seeds, SURVEY.md §7 "Randomness parity"): z, Gumbel uniforms and DiffAugment
parameters are arguments.

Parameters are plain dicts keyed by the reference's `state_dict()` names, in
the reference's layouts (ConvTranspose2d weights (Cin,Cout,kh,kw); Conv2d
weights (Cout,Cin,kh,kw); raw N(0,1) weights, EqualLR scale NOT baked in).
"""
import math
from collections import OrderedDict

import torch
import torch.nn.functional as F

LRELU_SLOPE = 0.2
LRELU_GAIN = math.sqrt(2.0)


# --------------------------------------------------------------------------
# bf16 emulation (checker for the engine's `enable_amp` mode; off = the fp32 restatement pinned by tests/golden)
# --------------------------------------------------------------------------
class _Emu:
    """With `bf16` on, the restatement rounds to bfloat16 exactly where the HIP engine STORES bfloat16:
    weight shadows, the latent, every feature map after its activation, BlurVH's output, and every gradient with
    respect to a pre-activation (the backward-data chains, which are also the weight-gradient operands); all
    arithmetic, bias gradients, head outputs, images, losses and the optimizer stay fp32, as in the engine.  The
    reference's own autocast path (trainers/dcgan_amp.py:194,219,226) is fp16 and rounds at ATen's op boundaries
    instead; this mode exists so that the timed bf16 path can be held to a tight gradient bound against a checker
    that makes the same leaky-relu slope decisions, not as a model of autocast."""
    bf16 = False


EMU = _Emu()


class _Round(torch.autograd.Function):
    """x -> bf16(x); its derivative is taken to be the same rounding, so the function is closed under differentiation
    (R1's double backward rounds the tangent where the first backward rounded the gradient)."""

    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return _Round.apply(g)


class _RoundFwd(torch.autograd.Function):
    """stored-as-bf16 value: rounded forward, straight-through backward"""

    @staticmethod
    def forward(ctx, x):
        return x.bfloat16().float()

    @staticmethod
    def backward(ctx, g):
        return g


class _RoundGrad(torch.autograd.Function):
    """stored-as-bf16 gradient: identity forward, the gradient passing back through this point is rounded"""

    @staticmethod
    def forward(ctx, x):
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return _Round.apply(g)


def _rf(x):
    return _RoundFwd.apply(x) if EMU.bf16 else x


def _rg(x):
    return _RoundGrad.apply(x) if EMU.bf16 else x


# --------------------------------------------------------------------------
# models/ops/common.py
# --------------------------------------------------------------------------
def pad_ring(h, ring=True):
    """`Pad(padding=1, horizontal=circular|reflect, vertical=reflect)`
    models/ops/common.py:9-20 as built by dcgan_eqlr.py:21,37,77."""
    h = F.pad(h, (1, 1, 0, 0), mode="circular" if ring else "reflect")
    h = F.pad(h, (0, 0, 1, 1), mode="reflect")
    return h


def equal_lr_scale(weight):
    """EqualLR runtime scale 1/sqrt(weight[0].numel()) models/ops/common.py:124-125
    (for ConvTranspose2d this is Cout*kh*kw, not Cin*kh*kw)."""
    return 1.0 / math.sqrt(weight[0].numel())


def fused_leaky_relu(x, bias):
    """FusedLeakyReLU.forward models/ops/common.py:99-106."""
    if x.ndim == 4:
        bias = bias.view(1, -1, 1, 1)
    return F.leaky_relu(x + bias, negative_slope=LRELU_SLOPE) * LRELU_GAIN


def blur_vh(x, ring=True):
    """BlurVH.forward models/ops/common.py:74-88 with Blur.forward :63-68:
    [1,2,1]/4 along H (reflect pad) and along W (circular pad), concatenated."""
    C = x.shape[1]
    k = torch.tensor([1.0, 2.0, 1.0], dtype=x.dtype) / 4.0
    kv = k.view(1, 1, 3, 1).repeat(C, 1, 1, 1)
    kh = k.view(1, 1, 1, 3).repeat(C, 1, 1, 1)
    # blur_v: padding=(0,0,1,1) -> only vertical reflect pad
    hv = F.pad(x, (0, 0, 1, 1), mode="reflect")
    hv = F.conv2d(hv, kv, groups=C)
    # blur_h: padding=(1,1,0,0) -> only horizontal pad
    hh = F.pad(x, (1, 1, 0, 0), mode="circular" if ring else "reflect")
    hh = F.conv2d(hh, kh, groups=C)
    return _rf(_rg(torch.cat([hv, hh], dim=1)))


# --------------------------------------------------------------------------
# models/gans/dcgan_eqlr.py
# --------------------------------------------------------------------------
def _g_prefix(params):
    return "backbone." if any(k.startswith("backbone.") for k in params) else ""


def proj(z, weight, bias):
    """Proj.forward dcgan_eqlr.py:6-16 (EqualLR(ConvTranspose2d(k=shape_in,s=1,p=0)) + FusedLeakyReLU)."""
    h = _rf(z)[..., None, None]
    h = _rg(F.conv_transpose2d(h * equal_lr_scale(weight), _rf(weight), None, 1, 0))
    return _rf(fused_leaky_relu(h, bias))


def up(x, weight, bias, ring=True):
    """Up dcgan_eqlr.py:19-26: Pad(1) -> EqualLR(ConvTranspose2d(4,2,padding=3,bias=False)) -> FusedLeakyReLU."""
    h = pad_ring(x, ring)
    h = _rg(F.conv_transpose2d(h * equal_lr_scale(weight), _rf(weight), None, 2, 3))
    return _rf(fused_leaky_relu(h, bias))


def head(x, weight, bias, ring=True):
    """One Head branch dcgan_eqlr.py:36-40: Pad(1) -> EqualLR(ConvTranspose2d(4,2,3,bias=True)).
    The bias is added after the (input-)scaled conv, i.e. it is NOT scaled."""
    h = pad_ring(x, ring)
    if not EMU.bf16:
        return F.conv_transpose2d(h * equal_lr_scale(weight), weight, bias, 2, 3)
    # (the engine's head output is fp32; the gradient it sends back into the backbone / the weight gradient is bf16)
    return _rg(F.conv_transpose2d(h * equal_lr_scale(weight), _rf(weight), None, 2, 3)) + bias.view(1, -1, 1, 1)


def generator_backbone(params, z, ring=True):
    """Generator.forward dcgan_eqlr.py:49-72 -> dict(depth[, confidence]); tanh on depth only."""
    p = _g_prefix(params)
    h = proj(z, params[p + "0.0.module.weight"], params[p + "0.1.bias"])
    for i in (1, 2, 3):
        h = up(h, params[p + f"{i}.1.module.weight"], params[p + f"{i}.2.bias"], ring)
    out = OrderedDict()
    for name in ("depth", "confidence"):
        wk = p + f"4.heads.{name}.1.module.weight"
        if wk in params:
            out[name] = head(h, params[wk], params[p + f"4.heads.{name}.1.module.bias"], ring)
    out["depth"] = torch.tanh(out["depth"])
    return out


def down(x, weight, bias, ring=True):
    """Down dcgan_eqlr.py:75-82: Pad(1) -> EqualLR(Conv2d(4,2,0,bias=False)) -> FusedLeakyReLU."""
    h = pad_ring(x, ring)
    h = _rg(F.conv2d(h * equal_lr_scale(weight), _rf(weight), None, 2, 0))
    return _rf(fused_leaky_relu(h, bias))


def discriminator(params, x, ring=True):
    """Discriminator dcgan_eqlr.py:85-96: BlurVH, Down x4, EqualLR(Conv2d(ch,1,shape_out))."""
    h = blur_vh(x, ring)
    for i in (1, 2, 3, 4):
        h = down(h, params[f"{i}.1.module.weight"], params[f"{i}.2.bias"], ring)
    w = params["5.module.weight"]
    return F.conv2d(h * equal_lr_scale(w), w, params["5.module.bias"], 1, 0)


# --------------------------------------------------------------------------
# models/dusty.py
# --------------------------------------------------------------------------
def logistic_noise(u1, u2, eps=1e-10):
    """GumbelSigmoid.logistic_noise models/dusty.py:30-36 with injected uniforms."""
    return -torch.log(torch.log(u1 + eps) / torch.log(u2 + eps) + eps)


def gumbel_sigmoid(logits, noise, tau=1.0, threshold=0.5):
    """GumbelSigmoid.forward models/dusty.py:45-59 (hard=True, straight-through)."""
    soft = torch.sigmoid((logits + noise) / tau)
    hard = (soft > threshold).float()
    return hard - soft.detach() + soft


def maskout(out, arch, noise, tau=1.0, drop_const=-1.0, training=True):
    """DUSty1.maskout models/dusty.py:77-91 / DUSty2.maskout :107-127.
    noise: dict with 'pixel' [B,1,H,W] and (dusty2, training) 'image' [B,1,1,1] logistic noise."""
    if arch == "none":
        return out
    depth, logit = out["depth"], out["confidence"]
    if arch == "dusty1":
        mask = gumbel_sigmoid(logit, noise["pixel"], tau)
        mask_cat = mask
    elif arch == "dusty2":
        mask_pixel = gumbel_sigmoid(logit[:, [0]], noise["pixel"], tau)
        if training:
            mask_image = gumbel_sigmoid(logit[:, [1]], noise["image"], tau)
        else:
            mask_image = (logit[:, [1]] > 0.0).float()
        mask = mask_pixel * mask_image
        mask_cat = torch.cat([mask_pixel, mask_image], dim=1)
    else:
        raise NotImplementedError(arch)
    out["depth_orig"] = depth
    out["mask"] = mask_cat
    out["depth"] = mask * depth + (1 - mask) * drop_const
    return out


def generator(params, z, arch="none", noise=None, tau=1.0, ring=True, training=True):
    """define_G product models/__init__.py:5-36: backbone (+ DUSty wrapper forward dusty.py:72-75,102-105)."""
    out = generator_backbone(params, z, ring)
    dc = float(params["drop_const"]) if "drop_const" in params else -1.0
    return maskout(out, arch, noise, tau, dc, training)


# --------------------------------------------------------------------------
# utils/diff_augment.py  (p = 1.0; every random draw is an argument)
# --------------------------------------------------------------------------
def translation_shift(H, W, ratio=1.0 / 8.0):
    """shift_h, shift_w of rand_translation utils/diff_augment.py:58."""
    return int(H * ratio / 2 + 0.5), int(W * ratio / 2 + 0.5)


def cutout_size(H, W, ratio=0.5):
    """cut_h, cut_w of rand_cutout utils/diff_augment.py:85."""
    return int(H * ratio + 0.5), int(W * ratio + 0.5)


def diff_augment(x, rp, policy=("brightness", "saturation", "contrast", "translation", "cutout")):
    """DiffAugment.forward utils/diff_augment.py:114-132 at p=1 with injected draws.

    rp: dict of per-sample tensors
        u_b, u_s, u_c : float [B]   the uniform_(-1,1) draw of brightness / saturation / contrast
        t_h, t_w      : long  [B]   translation draws (randint(-shift, shift+1))
        o_x, o_y      : long  [B]   cutout centre draws
    Quirks reproduced: `factor.bernoulli_(p) * factor.uniform_(-1,1)` aliases ONE tensor,
    so the factor is u*u; translation wraps columns modulo (W-1); shifted-out rows read zero.
    """
    B, C, H, W = x.shape
    for p in policy:
        if p == "brightness":  # :24-30
            u = rp["u_b"].view(B, 1, 1, 1)
            x = x + (u * u) * 0.5
        elif p == "saturation":  # :33-40 (identity for C == 1)
            u = rp["u_s"].view(B, 1, 1, 1)
            x_mean = x.mean(dim=1, keepdim=True)
            x = torch.lerp(x_mean, x, (u * u) * 1.0 + 1.0)
        elif p == "contrast":  # :43-50
            u = rp["u_c"].view(B, 1, 1, 1)
            x_mean = x.mean(dim=[1, 2, 3], keepdim=True)
            x = torch.lerp(x_mean, x, (u * u) * 0.5 + 1.0)
        elif p == "translation":  # :53-79
            th = rp["t_h"].view(B, 1, 1)
            tw = rp["t_w"].view(B, 1, 1)
            gb, gh, gw = torch.meshgrid(torch.arange(B), torch.arange(H), torch.arange(W), indexing="ij")
            x_pad = F.pad(x, [0, 0, 1, 1, 0, 0, 0, 0])
            gh = torch.clamp(gh + th + 1, min=0, max=H + 1)
            gw = (gw + tw) % (W - 1)
            x = x_pad.permute(0, 2, 3, 1).contiguous()[gb, gh, gw].permute(0, 3, 1, 2).contiguous()
        elif p == "cutout":  # :82-102
            cut_h, cut_w = cutout_size(H, W)
            ox = rp["o_x"].view(B, 1, 1)
            oy = rp["o_y"].view(B, 1, 1)
            gb, gx, gy = torch.meshgrid(torch.arange(B), torch.arange(cut_h), torch.arange(cut_w), indexing="ij")
            gx = torch.clamp(gx + ox - cut_h // 2, min=0, max=H - 1)
            gy = torch.clamp(gy + oy - cut_w // 2, min=0, max=W - 1)
            m = torch.ones(B, H, W, dtype=x.dtype)
            m[gb, gx, gy] = 0
            x = x * m.unsqueeze(1)
        else:
            raise KeyError(p)
    return x


def draw_augment_params(B, H, W, gen):
    """Draw one DiffAugment parameter set with a torch.Generator (test/bench convenience; the draw ORDER of the
    reference is documented in SURVEY.md §8c and replayed in tests/golden/make_golden.py, not here)."""
    sh, sw = translation_shift(H, W)
    ch, cw = cutout_size(H, W)
    return {
        "u_b": torch.rand(B, generator=gen) * 2 - 1,
        "u_s": torch.rand(B, generator=gen) * 2 - 1,
        "u_c": torch.rand(B, generator=gen) * 2 - 1,
        "t_h": torch.randint(-sh, sh + 1, (B,), generator=gen),
        "t_w": torch.randint(-sw, sw + 1, (B,), generator=gen),
        "o_x": torch.randint(0, H + (1 - ch % 2), (B,), generator=gen),
        "o_y": torch.randint(0, W + (1 - cw % 2), (B,), generator=gen),
    }


# --------------------------------------------------------------------------
# models/loss.py
# --------------------------------------------------------------------------
def _avg_diff(a, b):
    """average_diff models/loss.py:11-18 (tensor case)."""
    return a - b.mean(0, keepdim=True)


def gan_loss(metric, pred_real, pred_fake, mode, smoothing=1.0):
    """GANLoss.forward/loss_D/loss_G models/loss.py:29-88, all seven metrics."""
    if mode == "D":
        if metric == "nsgan":
            return F.softplus(-pred_real).mean() + F.softplus(pred_fake).mean()
        if metric == "wgan":
            return -pred_real.mean() + pred_fake.mean()
        if metric == "lsgan":
            return F.mse_loss(pred_real, torch.ones_like(pred_real) * smoothing) + F.mse_loss(
                pred_fake, torch.zeros_like(pred_fake)
            )
        if metric == "hinge":
            return F.relu(1 - pred_real).mean() + F.relu(1 + pred_fake).mean()
        if metric == "ragan":
            return (
                F.softplus(-1 * _avg_diff(pred_real, pred_fake)).mean()
                + F.softplus(_avg_diff(pred_fake, pred_real)).mean()
            )
        if metric == "rahinge":
            return (
                F.relu(1 - _avg_diff(pred_real, pred_fake)).mean()
                + F.relu(1 + _avg_diff(pred_fake, pred_real)).mean()
            )
        if metric == "ralsgan":
            return torch.mean((_avg_diff(pred_real, pred_fake) - 1.0) ** 2) + torch.mean(
                (_avg_diff(pred_fake, pred_real) + 1.0) ** 2
            )
        raise NotImplementedError(metric)
    if mode == "G":
        if metric == "nsgan":
            return F.softplus(-pred_fake).mean()
        if metric == "wgan":
            return -pred_fake.mean()
        if metric == "lsgan":
            return F.mse_loss(pred_fake, torch.ones_like(pred_fake))
        if metric == "hinge":
            return -pred_fake.mean()
        if metric == "ragan":
            return (
                F.softplus(_avg_diff(pred_real, pred_fake)).mean()
                + F.softplus(-1 * _avg_diff(pred_fake, pred_real)).mean()
            )
        if metric == "rahinge":
            return (
                F.relu(1 + _avg_diff(pred_real, pred_fake)).mean()
                + F.relu(1 - _avg_diff(pred_fake, pred_real)).mean()
            )
        if metric == "ralsgan":
            return torch.mean((_avg_diff(pred_real, pred_fake) + 1.0) ** 2) + torch.mean(
                (_avg_diff(pred_fake, pred_real) - 1.0) ** 2
            )
        raise NotImplementedError(metric)
    raise ValueError(mode)


# --------------------------------------------------------------------------
# trainers/dcgan_amp.py helpers
# --------------------------------------------------------------------------
def fetch_reals(pol, mask, min_depth=0.9, max_depth=120.0, drop_const=-1.0):
    """Trainer.fetch_reals trainers/dcgan_amp.py:154-160 with Coordinate.invert_depth utils/lidar.py:31-36
    and sigmoid_to_tanh utils/__init__.py:70-73."""
    mask = mask.float()
    depth = pol * (max_depth - min_depth) + min_depth
    disp = 1 / depth
    inv = (disp - 1 / max_depth) / (1 / min_depth - 1 / max_depth)
    inv = inv * 2.0 - 1.0
    inv = mask * inv + (1 - mask) * drop_const
    return inv, mask


def adam_update(p, g, m, v, step, lr, beta1, beta2, eps=1e-8):
    """torch.optim.Adam (no weight decay, no amsgrad) as configured at trainers/dcgan_amp.py:116-125.
    In place on p, m, v; `step` is the 1-based step count AFTER increment."""
    m.mul_(beta1).add_(g, alpha=1 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1 - beta2)
    bc1 = 1 - beta1 ** step
    bc2 = 1 - beta2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-(lr / bc1))


def ema_update(ema, new, decay):
    """ema_inplace trainers/dcgan_amp.py:30-35 over PARAMETERS only (buffers such as drop_const are skipped)."""
    for k in ema:
        if k == "drop_const":
            continue
        ema[k].copy_(decay * ema[k] + (1.0 - decay) * new[k])


def ema_decay(batch_size, smoothing_kimg):
    """trainers/dcgan_amp.py:75-77."""
    return 0.5 ** (batch_size / (smoothing_kimg * 1000))


PARAM_BUFFERS = ("drop_const",)


class StepConfig:
    """The live solver/model keys used inside Trainer.step."""

    def __init__(self, arch="none", ring=True, tau=1.0, gan_mode="nsgan", w_gan=1.0, w_gp=1.0, w_pl=0.0,
                 lr_g=0.002, lr_d=0.002, beta1=0.0, beta2=0.99, ema_decay=0.998,
                 policy=("brightness", "saturation", "contrast", "translation", "cutout"), emulate_bf16=False):
        self.emulate_bf16 = bool(emulate_bf16)  # round where the engine's bf16 mode stores bf16 (see _Emu)
        self.arch, self.ring, self.tau = arch, ring, tau
        self.gan_mode, self.w_gan, self.w_gp, self.w_pl = gan_mode, w_gan, w_gp, w_pl
        self.lr_g, self.lr_d, self.beta1, self.beta2 = lr_g, lr_d, beta1, beta2
        self.ema_decay = ema_decay
        self.policy = tuple(policy)


def new_optim_state(params):
    return {k: {"m": torch.zeros_like(v), "v": torch.zeros_like(v)} for k, v in params.items()
            if k not in PARAM_BUFFERS}


def train_step(G, D, G_ema, opt_G, opt_D, step_no, cfg, x_real, rand, return_grads=False):
    """`_train_step` under the configuration's precision model (fp32, or the bf16 storage emulation of `_Emu`)."""
    prev, EMU.bf16 = EMU.bf16, bool(getattr(cfg, "emulate_bf16", False))
    try:
        return _train_step(G, D, G_ema, opt_G, opt_D, step_no, cfg, x_real, rand, return_grads)
    finally:
        EMU.bf16 = prev


def _train_step(G, D, G_ema, opt_G, opt_D, step_no, cfg, x_real, rand, return_grads=False):
    """One `Trainer.step` trainers/dcgan_amp.py:162-325 (num_accumulation=1, world size 1, no AMP scaling,
    path-length regulariser off), restated with stock torch CPU ops + autograd.

    G, D, G_ema: parameter dicts (leaf tensors, updated IN PLACE); opt_*: new_optim_state() dicts.
    step_no: 1-based Adam step count for this call.
    x_real: [B,1,H,W] already through fetch_reals.
    rand: {"z":[B,nz], "noise":{"pixel","image"} logistic noise (dusty archs),
           "aug":[rpD_real, rpD_fake, rpG_real, rpG_fake]}  -- the four A(.) calls in reference order
           (:199, :200, :255, :256).
    Returns (scalars, extras).
    """
    scalars = {}
    extras = {}
    for d in (G, D):
        for k, p in d.items():
            if k not in PARAM_BUFFERS:
                p.requires_grad_(True)
                p.grad = None

    # ---- train D (:171-238)
    z = rand["z"]
    synth = generator(G, z, cfg.arch, rand.get("noise"), cfg.tau, cfg.ring, training=True)  # :195
    x_real_aug = diff_augment(x_real, rand["aug"][0], cfg.policy).detach().requires_grad_()  # :199
    x_fake_aug = diff_augment(synth["depth"], rand["aug"][1], cfg.policy).detach()  # :200
    y_real = discriminator(D, x_real_aug, cfg.ring)  # :203
    y_fake = discriminator(D, x_fake_aug, cfg.ring)  # :204
    scalars["loss/D/output/real"] = y_real.mean().detach()
    scalars["loss/D/output/fake"] = y_fake.mean().detach()
    loss_gan = gan_loss(cfg.gan_mode, y_real, y_fake, "D")
    loss_D = cfg.w_gan * loss_gan
    scalars["loss/D/adversarial"] = loss_gan.detach()
    if cfg.w_gp > 0:  # :216-232
        (grads,) = torch.autograd.grad(outputs=y_real.sum(), inputs=[x_real_aug], create_graph=True)
        r1 = (grads ** 2).sum(dim=[1, 2, 3]).mean()
        scalars["loss/D/gradient_penalty"] = r1.detach()
        loss_D = loss_D + (cfg.w_gp / 2) * r1 + 0.0 * y_real.squeeze()[0]
        extras["r1_grads"] = grads.detach()
    d_params = [p for k, p in D.items()]
    d_grads = torch.autograd.grad(loss_D, d_params, allow_unused=True)
    d_grads = [torch.zeros_like(p) if g is None else g for p, g in zip(d_params, d_grads)]
    if return_grads:
        extras["grad_D"] = {k: g.detach().clone() for k, g in zip(D.keys(), d_grads)}
        extras["x_real_aug"] = x_real_aug.detach()
        extras["x_fake_aug"] = x_fake_aug.detach()
        extras["y_real"] = y_real.detach()
        extras["y_fake"] = y_fake.detach()
        extras["synth"] = {k: v.detach() for k, v in synth.items()}
    with torch.no_grad():
        for (k, p), g in zip(D.items(), d_grads):
            adam_update(p, g, opt_D[k]["m"], opt_D[k]["v"], step_no, cfg.lr_d, cfg.beta1, cfg.beta2)

    # ---- train G (:240-312); D is now the UPDATED D, G's graph from :195 is reused (:256)
    x_fake_aug2 = diff_augment(synth["depth"], rand["aug"][3], cfg.policy)  # :256
    if cfg.gan_mode in ("ragan", "rahinge", "ralsgan"):
        x_real_aug2 = diff_augment(x_real, rand["aug"][2], cfg.policy).detach()  # :255
        y_real2 = discriminator({k: v.detach() for k, v in D.items()}, x_real_aug2, cfg.ring)  # :259
    else:
        y_real2 = None  # nsgan/wgan/lsgan/hinge loss_G never reads pred_real (models/loss.py:66-75)
    y_fake2 = discriminator({k: v.detach() for k, v in D.items()}, x_fake_aug2, cfg.ring)  # :260
    loss_gan_g = gan_loss(cfg.gan_mode, y_real2, y_fake2, "G")
    loss_G = cfg.w_gan * loss_gan_g
    scalars["loss/G/adversarial"] = loss_gan_g.detach()
    if cfg.w_pl > 0:  # path-length regularisation :268-306 (no GradScaler: its scale / unscale pair cancels)
        pl = rand["pl"]  # {"z": [B//2,nz], "noise": Gumbel noise of that forward, "y": randn_like(x_pl), "pl_ema": 0-dim}
        z_pl = pl["z"].clone().requires_grad_()
        x_pl = generator(G, z_pl, cfg.arch, pl.get("noise"), cfg.tau, cfg.ring, training=True)["depth"]  # :276-277
        noise_pl = pl["y"] / math.sqrt(x_pl.shape[2] * x_pl.shape[3])  # :278-279
        outputs = (x_pl * noise_pl).sum()
        (grads,) = torch.autograd.grad(outputs=outputs, inputs=[z_pl], create_graph=True)  # :282-287
        pl_lengths = torch.sqrt(grads.pow(2).sum(dim=-1))  # :294-295
        pl_ema = pl["pl_ema"].lerp(pl_lengths.mean(), 0.01)  # :297 (the mean stays in the graph)
        extras["pl_ema"] = pl_ema.detach().clone()  # :298
        pl_penalty = (pl_lengths - pl_ema).pow(2).mean()  # :300
        scalars["loss/G/path_length/baseline"] = pl_ema.detach()
        scalars["loss/G/path_length"] = pl_penalty.detach()
        loss_G = loss_G + cfg.w_pl * pl_penalty + 0.0 * x_pl[0, 0, 0, 0]  # :305-306
        if return_grads:
            extras["pl_grads_z"] = grads.detach().clone()
    g_keys = [k for k in G if k not in PARAM_BUFFERS]
    g_params = [G[k] for k in g_keys]
    g_grads = torch.autograd.grad(loss_G, g_params, allow_unused=True)
    g_grads = [torch.zeros_like(p) if g is None else g for p, g in zip(g_params, g_grads)]
    if return_grads:
        extras["grad_G"] = {k: g.detach().clone() for k, g in zip(g_keys, g_grads)}
        extras["y_fake2"] = y_fake2.detach()
    with torch.no_grad():
        for k, p, g in zip(g_keys, g_params, g_grads):
            adam_update(p, g, opt_G[k]["m"], opt_G[k]["v"], step_no, cfg.lr_g, cfg.beta1, cfg.beta2)
        ema_update(G_ema, G, cfg.ema_decay)  # :316
    for d in (G, D):
        for k, p in d.items():
            p.requires_grad_(False)
    return {k: float(v) for k, v in scalars.items()}, extras


# --------------------------------------------------------------------------
# parameter construction in the reference's state_dict layout
# --------------------------------------------------------------------------
def ch_plan(ch_base, ch_max):
    return [min(ch_base << i, ch_max) for i in range(4)]


def init_G(arch, in_ch, ch_base, ch_max, shape, gen, drop_const=-1.0):
    """Shapes/keys of define_G's state_dict; weights ~ N(0,1), biases 0 (EqualLR init common.py:128-130)."""
    masker = arch.split("/")[0]
    pre = "" if masker == "none" else "backbone."
    ch = ch_plan(ch_base, ch_max)
    h0, w0 = shape[0] >> 4, shape[1] >> 4
    P = OrderedDict()
    if masker != "none":
        P["drop_const"] = torch.tensor(float(drop_const))
    P[pre + "0.0.module.weight"] = torch.randn(in_ch, ch[3], h0, w0, generator=gen)
    P[pre + "0.1.bias"] = torch.zeros(ch[3])
    for i, (ci, co) in enumerate(((ch[3], ch[2]), (ch[2], ch[1]), (ch[1], ch[0])), start=1):
        P[pre + f"{i}.1.module.weight"] = torch.randn(ci, co, 4, 4, generator=gen)
        P[pre + f"{i}.2.bias"] = torch.zeros(co)
    heads = {"none": {"depth": 1}, "dusty1": {"depth": 1, "confidence": 1}, "dusty2": {"depth": 1, "confidence": 2}}[masker]
    for name, k in heads.items():
        P[pre + f"4.heads.{name}.1.module.weight"] = torch.randn(ch[0], k, 4, 4, generator=gen)
        P[pre + f"4.heads.{name}.1.module.bias"] = torch.zeros(k)
    return P


def init_D(in_ch, ch_base, ch_max, shape, gen):
    ch = ch_plan(ch_base, ch_max)
    h0, w0 = shape[0] >> 4, shape[1] >> 4
    P = OrderedDict()
    cin = in_ch * 2
    for i in range(4):
        P[f"{i + 1}.1.module.weight"] = torch.randn(ch[i], cin, 4, 4, generator=gen)
        P[f"{i + 1}.2.bias"] = torch.zeros(ch[i])
        cin = ch[i]
    P["5.module.weight"] = torch.randn(1, ch[3], h0, w0, generator=gen)
    P["5.module.bias"] = torch.zeros(1)
    return P
