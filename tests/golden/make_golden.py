#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE's own modules.

Runs only in the build container, where /root/reference is mounted.  It imports the reference's
`models/*`, `models/loss.py` and `utils/diff_augment.py` unmodified, drives them with a restatement of
`Trainer.step` (trainers/dcgan_amp.py:162-325 minus DDP / torch.cuda.amp, which need a GPU -- SURVEY.md §0.5),
captures every random draw by replaying torch's global generator, and stores inputs + expected outputs as
.npz.  The vectors are DATA; no reference source travels.  The oracle (oracle/dusty_oracle.py) and the HIP path
are then both checked against these files without the reference present.

usage:  python tests/golden/make_golden.py            (writes tests/golden/*.npz)
        python tests/golden/make_golden.py gan_modes  (only the six step_*_<gan_mode>.npz files)
        python tests/golden/make_golden.py pl         (only the three step_*_pl.npz files)
        python tests/golden/make_golden.py lidar      (only lidar.npz)
        python tests/golden/make_golden.py metrics    (only metrics.npz)
        python tests/golden/make_golden.py full       (only full_dusty2.npz: the 64x1024 / 512-channel pin, digests)
        python tests/golden/make_golden.py covmmd     (only covmmd.npz: COV / MMD / 1-NNA from the reference's functions)
"""
import importlib.util
import math
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("DUSTY_REFERENCE", "/root/reference")
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)

import models  # noqa: E402  (reference package)
from models.loss import GANLoss  # noqa: E402


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


# utils/__init__.py imports cv2/omegaconf/kornia/numba (absent here): load the two files we need by path.
diff_augment = _load(os.path.join(REF, "utils", "diff_augment.py"), "ref_diff_augment")


def _load_lidar():
    """utils/lidar.py does `from . import render` (kornia/numba, visualisation only).  Give the import an empty
    placeholder so the file loads; only Coordinate.invert_depth (pure arithmetic, :31-36) is used."""
    pkg = types.ModuleType("ref_utils")
    pkg.__path__ = [os.path.join(REF, "utils")]
    sys.modules["ref_utils"] = pkg
    sys.modules["ref_utils.render"] = types.ModuleType("ref_utils.render")
    spec = importlib.util.spec_from_file_location("ref_utils.lidar", os.path.join(REF, "utils", "lidar.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["ref_utils.lidar"] = mod
    spec.loader.exec_module(mod)
    return mod


class Cfg(dict):
    __getattr__ = dict.__getitem__


def make_cfg(arch, in_ch, ch_base, ch_max, shape, ring):
    heads = {"none": {"depth": 1}, "dusty1": {"depth": 1, "confidence": 1}, "dusty2": {"depth": 1, "confidence": 2}}
    gen = Cfg(arch=f"{arch}/dcgan_eqlr", in_ch=in_ch, out_ch=heads[arch], ch_base=ch_base, ch_max=ch_max,
              drop_const=-1, shape=shape, tau=1)
    dis = Cfg(arch="dcgan_eqlr", in_ch=1, ch_base=ch_base, ch_max=ch_max, shape=shape)
    return Cfg(model=Cfg(gen=gen, dis=dis, ring=ring))


# ---------------------------------------------------------------- randomness capture
def capture_gumbel(arch, B, H, W):
    """Replay GumbelSigmoid.logistic_noise's draws (models/dusty.py:30-36) in module call order
    (dusty1: pixel; dusty2: pixel then image, models/dusty.py:116-118)."""
    eps = 1e-10
    out = {}
    shapes = {"none": [], "dusty1": [("pixel", (B, 1, H, W))],
              "dusty2": [("pixel", (B, 1, H, W)), ("image", (B, 1, 1, 1))]}[arch]
    for name, shp in shapes:
        u1 = torch.rand(*shp)
        u2 = torch.rand_like(u1)
        out[name] = -torch.log(torch.log(u1 + eps) / torch.log(u2 + eps) + eps)
    return out


def capture_aug(B, H, W, policy):
    """Replay the draws of one DiffAugment.forward (utils/diff_augment.py:27-28,36-38,46-48,59-60,77,86-87,99)."""
    rp = {}
    for p in policy:
        if p in ("brightness", "saturation", "contrast"):
            f = torch.empty((B, 1, 1, 1))
            f.bernoulli_(p=1.0)
            f.uniform_(-1, 1)
            rp[{"brightness": "u_b", "saturation": "u_s", "contrast": "u_c"}[p]] = f.flatten().clone()
        elif p == "translation":
            sh, sw = int(H * (1 / 8) / 2 + 0.5), int(W * (1 / 8) / 2 + 0.5)
            rp["t_h"] = torch.randint(-sh, sh + 1, size=[B, 1, 1]).flatten()
            rp["t_w"] = torch.randint(-sw, sw + 1, size=[B, 1, 1]).flatten()
            torch.empty(B).bernoulli_(p=1.0)
        elif p == "cutout":
            ch, cw = int(H * 0.5 + 0.5), int(W * 0.5 + 0.5)
            rp["o_x"] = torch.randint(0, H + (1 - ch % 2), size=[B, 1, 1]).flatten()
            rp["o_y"] = torch.randint(0, W + (1 - cw % 2), size=[B, 1, 1]).flatten()
            torch.empty(B).bernoulli_(p=1.0)
    return rp


def run_and_capture(fn, capture):
    """Run fn() on the global generator, then rewind and replay `capture()` to learn what it drew."""
    s0 = torch.get_rng_state()
    out = fn()
    s1 = torch.get_rng_state()
    torch.set_rng_state(s0)
    cap = capture()
    assert torch.equal(torch.get_rng_state(), s1), "replayed draws do not match the reference's RNG consumption"
    return out, cap


def ema_inplace(ema_model, new_model, decay):  # trainers/dcgan_amp.py:30-35
    ep, npar = dict(ema_model.named_parameters()), dict(new_model.named_parameters())
    with torch.no_grad():
        for k in ep:
            ep[k].copy_(decay * ep[k] + (1.0 - decay) * npar[k])


def sd_np(module, prefix):
    return {f"{prefix}/{k}": v.detach().numpy().copy() for k, v in module.state_dict().items()}


def make_step_golden(name, arch, ring, seed, in_ch=8, ch_base=4, ch_max=16, shape=(32, 64), B=2, steps=2,
                     gan_mode="nsgan", gp=1.0, pl=0.0):
    torch.manual_seed(seed)
    cfg = make_cfg(arch, in_ch, ch_base, ch_max, list(shape), ring)
    G = models.define_G(cfg)
    D = models.define_D(cfg)
    G_ema = models.define_G(cfg)
    G_ema.eval()
    ema_inplace(G_ema, G, 0.0)
    A = diff_augment.DiffAugment(policy=None)
    crit = GANLoss(gan_mode)
    lr, b1, b2 = 0.002, 0.0, 0.99
    optG = torch.optim.Adam(G.parameters(), lr=lr, betas=(b1, b2))
    optD = torch.optim.Adam(D.parameters(), lr=lr, betas=(b1, b2))
    decay = 0.5 ** (B / (10 * 1000))
    H, W = shape
    lidar = _load_lidar()

    data = {"meta/arch": np.array(arch), "meta/ring": np.array(ring), "meta/shape": np.array(shape),
            "meta/in_ch": np.array(in_ch), "meta/ch_base": np.array(ch_base), "meta/ch_max": np.array(ch_max),
            "meta/B": np.array(B), "meta/steps": np.array(steps), "meta/gan_mode": np.array(gan_mode),
            "meta/gp": np.array(gp), "meta/pl": np.array(pl), "meta/lr": np.array(lr), "meta/beta1": np.array(b1), "meta/beta2": np.array(b2),
            "meta/ema_decay": np.array(decay), "meta/torch": np.array(torch.__version__)}
    data.update(sd_np(G, "init/G"))
    data.update(sd_np(D, "init/D"))

    pl_ema_state = torch.tensor(0.0)  # trainers/dcgan_amp.py:112 register_buffer-like state
    for it in range(steps):
        pre = f"s{it}"
        # synthetic "dataset" batch (polar depth in [0,1] + validity mask), then fetch_reals (:154-160)
        pol = torch.rand(B, 1, H, W)
        mask_b = torch.rand(B, 1, H, W) > 0.15
        pol = pol * mask_b
        inv = lidar.Coordinate.invert_depth(Cfg(min_depth=0.9, max_depth=120.0,
                                                denormalize_minmax=lidar.Coordinate.denormalize_minmax,
                                                normalize_minmax=lidar.Coordinate.normalize_minmax), pol)
        inv = inv * 2.0 - 1.0  # sigmoid_to_tanh utils/__init__.py:70-73
        m = mask_b.float()
        x_real = m * inv + (1 - m) * (-1.0)
        data[f"{pre}/pol"], data[f"{pre}/mask"], data[f"{pre}/x_real"] = pol.numpy(), mask_b.numpy(), x_real.numpy()

        G.train()
        for p in D.parameters():
            p.requires_grad = True
        optD.zero_grad(set_to_none=True)
        z = torch.randn(B, in_ch)
        data[f"{pre}/z"] = z.numpy()
        synth, noise = run_and_capture(lambda: G(latent=z), lambda: capture_gumbel(arch, B, H, W))
        for k, v in noise.items():
            data[f"{pre}/noise/{k}"] = v.numpy()
        x_real_aug, rp0 = run_and_capture(lambda: A(x_real), lambda: capture_aug(B, H, W, A.policy))
        x_real_aug = x_real_aug.detach().requires_grad_()
        x_fake_aug, rp1 = run_and_capture(lambda: A(synth["depth"]), lambda: capture_aug(B, H, W, A.policy))
        x_fake_aug = x_fake_aug.detach()
        y_real, y_fake = D(x_real_aug), D(x_fake_aug)
        sc = {"loss/D/output/real": y_real.mean().item(), "loss/D/output/fake": y_fake.mean().item()}
        loss_gan = crit(y_real, y_fake, "D")
        loss_D = 1.0 * loss_gan
        sc["loss/D/adversarial"] = loss_gan.item()
        if gp > 0:
            (grads,) = torch.autograd.grad(outputs=y_real.sum(), inputs=[x_real_aug], create_graph=True,
                                           only_inputs=True)
            r1 = (grads ** 2).sum(dim=[1, 2, 3]).mean()
            sc["loss/D/gradient_penalty"] = r1.item()
            loss_D = loss_D + (gp / 2) * r1 + 0.0 * y_real.squeeze()[0]
            data[f"{pre}/r1_grads"] = grads.detach().numpy()
        loss_D.backward()
        for k, p in D.named_parameters():
            data[f"{pre}/grad_D/{k}"] = (torch.zeros_like(p) if p.grad is None else p.grad).numpy().copy()
        optD.step()
        for k, v in synth.items():
            data[f"{pre}/synth/{k}"] = v.detach().numpy()
        data[f"{pre}/x_real_aug"], data[f"{pre}/x_fake_aug"] = x_real_aug.detach().numpy(), x_fake_aug.numpy()
        data[f"{pre}/y_real"], data[f"{pre}/y_fake"] = y_real.detach().numpy(), y_fake.detach().numpy()

        for p in D.parameters():
            p.requires_grad = False
        optG.zero_grad(set_to_none=True)
        x_real_aug2, rp2 = run_and_capture(lambda: A(x_real), lambda: capture_aug(B, H, W, A.policy))
        x_fake_aug2, rp3 = run_and_capture(lambda: A(synth["depth"]), lambda: capture_aug(B, H, W, A.policy))
        y_real2, y_fake2 = D(x_real_aug2.detach()), D(x_fake_aug2)
        loss_gan_g = crit(y_real2, y_fake2, "G")
        sc["loss/G/adversarial"] = loss_gan_g.item()
        loss_G = 1.0 * loss_gan_g
        if pl > 0:  # path-length regularisation, trainers/dcgan_amp.py:268-306 (GradScaler's scale / unscale cancels)
            B_pl = B // 2
            z_pl = torch.randn(B_pl, in_ch)
            data[f"{pre}/pl/z"] = z_pl.numpy().copy()
            z_pl.requires_grad_()
            synth_pl, noise2 = run_and_capture(lambda: G(latent=z_pl), lambda: capture_gumbel(arch, B_pl, H, W))
            for k, v in noise2.items():
                data[f"{pre}/pl/noise/{k}"] = v.numpy()
            x_pl = synth_pl["depth"]
            y_pl = torch.randn_like(x_pl)
            data[f"{pre}/pl/y"] = y_pl.numpy().copy()
            data[f"{pre}/pl/pl_ema"] = pl_ema_state.numpy().copy()
            noise_pl = y_pl / np.sqrt(np.prod(x_pl.shape[2:]))
            outputs = (x_pl * noise_pl).sum()
            (grads_z,) = torch.autograd.grad(outputs=outputs, inputs=[z_pl], create_graph=True, only_inputs=True)
            pl_lengths = torch.sqrt(grads_z.pow(2).sum(dim=-1))
            pl_ema = pl_ema_state.lerp(pl_lengths.mean(), 0.01)
            pl_ema_state.copy_(pl_ema.detach())
            pl_penalty = (pl_lengths - pl_ema).pow(2).mean()
            sc["loss/G/path_length/baseline"] = pl_ema_state.item()
            sc["loss/G/path_length"] = pl_penalty.item()
            data[f"{pre}/pl/grads_z"] = grads_z.detach().numpy().copy()
            loss_G = loss_G + pl * pl_penalty + 0.0 * x_pl[0, 0, 0, 0]
        loss_G.backward()
        for k, p in G.named_parameters():
            data[f"{pre}/grad_G/{k}"] = (torch.zeros_like(p) if p.grad is None else p.grad).numpy().copy()
        optG.step()
        ema_inplace(G_ema, G, decay)
        data[f"{pre}/y_fake2"] = y_fake2.detach().numpy()
        for j, rp in enumerate((rp0, rp1, rp2, rp3)):
            for k, v in rp.items():
                data[f"{pre}/aug{j}/{k}"] = v.numpy()
        for k, v in sc.items():
            data[f"{pre}/scalar/{k}"] = np.array(v, dtype=np.float64)
        data.update(sd_np(G, f"{pre}/after/G"))
        data.update(sd_np(D, f"{pre}/after/D"))
        data.update(sd_np(G_ema, f"{pre}/after/G_ema"))

    # final Adam second-moment state (exp_avg_sq) keyed like the parameters
    for opt, mod, tag in ((optG, G, "G"), (optD, D, "D")):
        for (k, p) in mod.named_parameters():
            st = opt.state[p]
            data[f"final/optim_{tag}/{k}/exp_avg"] = st["exp_avg"].numpy().copy()
            data[f"final/optim_{tag}/{k}/exp_avg_sq"] = st["exp_avg_sq"].numpy().copy()
    path = os.path.join(HERE, f"step_{name}.npz")
    np.savez_compressed(path, **data)
    print("wrote", path, f"{os.path.getsize(path) / 1024:.0f} KiB")


def make_ops_golden():
    """Per-op vectors: Pad, BlurVH, Up/Down/Proj/Head modules (fwd + input/weight grads), each rand_* augment,
    GumbelSigmoid fwd/bwd, every GANLoss metric, invert_depth."""
    import models.ops.common as ops
    from models.gans import dcgan_eqlr as net
    from models import dusty

    torch.manual_seed(7)
    d = {"meta/torch": np.array(torch.__version__)}
    x = torch.randn(2, 3, 6, 8)
    for ring in (True, False):
        pad = ops.Pad(padding=1, horizontal="circular" if ring else "reflect", vertical="reflect")
        d[f"pad/ring{int(ring)}/x"], d[f"pad/ring{int(ring)}/y"] = x.numpy(), pad(x).numpy()
        xb = torch.randn(2, 1, 8, 16)
        d[f"blurvh/ring{int(ring)}/x"], d[f"blurvh/ring{int(ring)}/y"] = xb.numpy(), ops.BlurVH(ring)(xb).numpy()

    def module_case(tag, mod, xin):
        xin = xin.clone().requires_grad_()
        y = mod(xin)
        gy = torch.randn_like(y)
        grads = torch.autograd.grad(y, [xin] + list(mod.parameters()), gy)
        d[f"{tag}/x"], d[f"{tag}/y"], d[f"{tag}/gy"], d[f"{tag}/gx"] = xin.detach().numpy(), y.detach().numpy(), gy.numpy(), grads[0].numpy()
        for (k, p), g in zip(mod.named_parameters(), grads[1:]):
            d[f"{tag}/param/{k}"], d[f"{tag}/grad/{k}"] = p.detach().numpy(), g.numpy()

    for ring in (True, False):
        r = f"ring{int(ring)}"
        up = net.Up(6, 4, ring)
        up[2].bias.data.normal_()
        module_case(f"up/{r}", up, torch.randn(2, 6, 4, 8))
        dn = net.Down(4, 6, ring)
        dn[2].bias.data.normal_()
        module_case(f"down/{r}", dn, torch.randn(2, 4, 8, 16))
    pj = net.Proj(5, 6, (2, 4))
    pj[1].bias.data.normal_()
    module_case("proj", pj, torch.randn(3, 5))
    hd = net.Head(4, {"depth": 1, "confidence": 2}, True)
    for h in hd.heads.values():
        h[1].module.bias.data.normal_()
    xh = torch.randn(2, 4, 4, 8)
    yh = hd(xh)
    d["head/x"] = xh.numpy()
    for k, v in yh.items():
        d[f"head/y/{k}"] = v.detach().numpy()
    for k, v in hd.state_dict().items():
        d[f"head/param/{k}"] = v.numpy()

    # augment functions one by one (p=1)
    xa = torch.randn(3, 1, 16, 32)
    d["aug/x"] = xa.numpy()
    for fname, policy in (("brightness", ["brightness"]), ("saturation", ["saturation"]), ("contrast", ["contrast"]),
                          ("translation", ["translation"]), ("cutout", ["cutout"])):
        y, rp = run_and_capture(lambda: diff_augment.AUGMENT_FNS[fname](xa, p=1.0),
                                lambda: capture_aug(3, 16, 32, policy))
        d[f"aug/{fname}/y"] = y.numpy()
        for k, v in rp.items():
            d[f"aug/{fname}/{k}"] = v.numpy()

    # GumbelSigmoid fwd/bwd with captured noise
    gs = dusty.GumbelSigmoid(tau=1.0, hard=True, pixelwise=True)
    lg = torch.randn(2, 1, 8, 16, requires_grad=True)
    ym, noise = run_and_capture(lambda: gs(lg), lambda: capture_gumbel("dusty1", 2, 8, 16))
    gy = torch.randn_like(ym)
    (glg,) = torch.autograd.grad(ym, lg, gy)
    d["gumbel/logits"], d["gumbel/noise"], d["gumbel/y"], d["gumbel/gy"], d["gumbel/glogits"] = (
        lg.detach().numpy(), noise["pixel"].numpy(), ym.detach().numpy(), gy.numpy(), glg.numpy())

    # GANLoss, all metrics
    pr, pf = torch.randn(5, 1, 1, 1), torch.randn(5, 1, 1, 1)
    d["ganloss/pred_real"], d["ganloss/pred_fake"] = pr.numpy(), pf.numpy()
    for metric in ("nsgan", "wgan", "lsgan", "hinge", "ragan", "rahinge", "ralsgan"):
        c = GANLoss(metric)
        d[f"ganloss/{metric}/D"] = np.array(c(pr, pf, "D").item())
        d[f"ganloss/{metric}/G"] = np.array(c(pr, pf, "G").item())

    lidar = _load_lidar()
    pol = torch.rand(2, 1, 4, 8)
    inv = lidar.Coordinate.invert_depth(Cfg(min_depth=0.9, max_depth=120.0,
                                            denormalize_minmax=lidar.Coordinate.denormalize_minmax,
                                            normalize_minmax=lidar.Coordinate.normalize_minmax), pol)
    d["invert_depth/pol"], d["invert_depth/inv"] = pol.numpy(), inv.numpy()
    path = os.path.join(HERE, "ops.npz")
    np.savez_compressed(path, **d)
    print("wrote", path, f"{os.path.getsize(path) / 1024:.0f} KiB")


def make_lidar_golden():
    """utils/lidar.py vectors from the reference's own LiDAR class: the angle grid resize of
    init_coordmap, invert_depth / revert_depth, pol_to_xyz and inv_to_xyz.  The angle file is data made here (the
    reference ships none): elevation rows + azimuth columns with per-cell jitter, like process_kitti.py:150-197."""
    import tempfile
    lidar = _load_lidar()
    torch.manual_seed(31)
    Hs, Ws, H, W, B = 8, 64, 8, 32, 3
    pitch = torch.linspace(0.05, -0.42, Hs)[:, None].expand(Hs, Ws) + 0.002 * torch.randn(Hs, Ws)
    yaw = torch.linspace(math.pi, -math.pi, Ws)[None, :].expand(Hs, Ws) + 0.002 * torch.randn(Hs, Ws)
    angle_src = torch.stack([pitch, yaw]).contiguous()
    d = {"angle_src": angle_src.numpy(), "meta/shape": np.array([H, W]), "meta/min_depth": np.array(0.9),
         "meta/max_depth": np.array(120.0), "meta/torch": np.array(torch.__version__)}
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "angles.pt")
        torch.save(angle_src, path)
        L = lidar.LiDAR(num_ring=H, num_points=W, min_depth=0.9, max_depth=120.0, angle_file=path)
    d["angle"] = L.angle.numpy()
    inv = torch.rand(B, 1, H, W)
    inv[torch.rand(B, 1, H, W) < 0.2] = 0.0  # dropped points (drop_const of Coordinate = 0)
    d["inv"] = inv.numpy()
    d["revert_depth/norm"] = L.revert_depth(inv[inv > 0]).numpy()
    d["revert_depth/metric"] = L.revert_depth(inv[inv > 0], norm=False).numpy()
    d["invert_depth"] = L.invert_depth(L.revert_depth(inv[inv > 0])).numpy()
    d["points"] = L.inv_to_xyz(inv.clone()).numpy()
    d["pol_to_xyz"] = L.pol_to_xyz(inv).numpy()
    gen = torch.tanh(torch.randn(B, 1, H, W) * 1.5)
    gen[torch.rand(B, 1, H, W) < 0.2] = -1.0  # dusty drop_const in tanh space -> 0 after tanh_to_sigmoid
    d["gen_depth"] = gen.numpy()
    d["gen_points"] = L.inv_to_xyz(((gen + 1.0) / 2.0).clamp_(0, 1)).numpy()  # utils/__init__.py:168,176
    # surface-normal images of both point maps (utils/geometry.py is plain torch; xyz_to_normal of utils/__init__.py:215-219
    # restated around it: utils/__init__.py itself cannot be imported here, see the top of this file)
    geometry = _load(os.path.join(REF, "utils", "geometry.py"), "ref_geometry")
    for tag, pts in (("normals", d["points"]), ("gen_normals", d["gen_points"])):
        nrm = -geometry.estimate_surface_normal(torch.from_numpy(pts), mode="closest")
        nrm[nrm != nrm] = 0.0
        d[tag] = ((nrm + 1.0) / 2.0).clamp_(0.0, 1.0).numpy()
    path = os.path.join(HERE, "lidar.npz")
    np.savez_compressed(path, **d)
    print("wrote", path, f"{os.path.getsize(path) / 1024:.0f} KiB")


def make_kitti_golden():
    """datasets/kitti.py:54-67 - KITTIOdometry.preprocess itself (input side).  The file imports
    torchvision (absent here) for `transform`; `preprocess` is pure numpy, so the file is loaded by path with an empty
    torchvision placeholder in sys.modules and only `preprocess` is called (the NEAREST resize of `transform` is
    torchvision's, a third-party step the restatement replaces by F.interpolate(mode="nearest") - unpinned, as before).
    Scans: random points plus every threshold case the masks decide on - zeros, |.| just below / at / above min_depth and
    max_depth."""
    for name in ("torchvision", "torchvision.transforms", "torchvision.transforms.functional"):
        sys.modules.setdefault(name, types.ModuleType(name))
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["torchvision.transforms"].functional = sys.modules["torchvision.transforms.functional"]
    kitti = _load(os.path.join(REF, "datasets", "kitti.py"), "ref_kitti")
    import tempfile
    rng = np.random.default_rng(2026)
    d = {"meta/min_depth": np.array(0.9), "meta/max_depth": np.array(120.0), "meta/numpy": np.array(np.__version__)}
    with tempfile.TemporaryDirectory() as tmp:
        ds = kitti.KITTIOdometry(tmp, "val", shape=(8, 32), min_depth=0.9, max_depth=120.0)   # (no files: an empty datalist)
        for k, (Hs, Ws, C) in enumerate([(16, 100, 3), (8, 64, 5), (32, 128, 4)]):
            r = np.exp(rng.uniform(np.log(0.3), np.log(200.0), (Hs, Ws)))
            dirs = rng.normal(size=(Hs, Ws, 3))
            dirs /= np.linalg.norm(dirs, axis=2, keepdims=True)
            xyz = (dirs * r[..., None]).astype(np.float32)
            u = rng.random((Hs, Ws))
            xyz[u < 0.05] = 0.0
            xyz[(u > 0.05) & (u < 0.08)] = (0.9, 0.0, 0.0)            # |.| == min_depth -> invalid (strict >)
            xyz[(u > 0.08) & (u < 0.11)] = (0.0, np.float32(0.9000001), 0.0)
            xyz[(u > 0.11) & (u < 0.14)] = (120.0, 0.0, 0.0)          # |.| == max_depth -> invalid (strict <)
            xyz[(u > 0.14) & (u < 0.17)] = (0.0, 0.0, np.float32(119.99999))
            pts = np.concatenate([xyz, rng.random((Hs, Ws, C - 3)).astype(np.float32)], -1) if C > 3 else xyz
            pts = pts.astype(np.float32)
            out = ds.preprocess({"xyz": pts[..., :3].copy()})          # __getitem__ :79-85 up to transform
            d[f"s{k}/points"] = pts
            d[f"s{k}/depth"], d[f"s{k}/mask"], d[f"s{k}/xyz"] = out["depth"], out["mask"], out["xyz"]
    path = os.path.join(HERE, "kitti_pre.npz")
    np.savez_compressed(path, **d)
    print("wrote", path, f"{os.path.getsize(path) / 1024:.0f} KiB")


def make_metrics_golden():
    """utils/metrics/jsd.py (plain torch, importable) on seeded clouds: the occupancy-grid counters of both sets and the
    divergence.  Clouds live in the radius-0.5 ball like trainers/dcgan_amp.py:387 feeds them
    (unit-space points / 2), with a share of dropped points at the origin."""
    jsd = _load(os.path.join(REF, "utils", "metrics", "jsd.py"), "ref_jsd")
    torch.manual_seed(41)
    d = {"meta/resolution": np.array(28), "meta/torch": np.array(torch.__version__)}
    sets = {}
    for name, scale in (("gen", 0.22), ("ref", 0.3)):
        v = torch.randn(6, 200, 3)
        r = torch.rand(6, 200, 1) ** (1 / 3) * 0.5 * (0.4 + scale)
        pts = v / v.norm(dim=2, keepdim=True) * r.clamp(max=0.5)
        pts[torch.rand(6, 200) < 0.1] = 0.0
        sets[name] = pts
        d[f"pcs_{name}"] = pts.numpy()
        _, counters = jsd.entropy_of_occupancy_grid(pts, 28, True, 128, False)
        d[f"counters_{name}"] = counters.numpy()
    d["jsd"] = np.array(jsd.compute_jsd(sets["gen"], sets["ref"], verbose=False))
    d["grid"] = jsd.unit_cube_grid_point_cloud(28, True, "cpu")[0].numpy()
    # utils/metrics/swd.py (plain torch): scores for seeded images, with the reference's random draws captured by
    # replaying the generator in compute_swd's draw order (per minibatch: set 1 levels, set 2 levels; then per level
    # dir_repeats direction matrices)
    swd = _load(os.path.join(REF, "utils", "metrics", "swd.py"), "ref_swd")
    B, C, H, W, bs, npatch, reps, ndirs = 6, 1, 32, 64, 4, 128, 4, 128
    torch.manual_seed(43)
    img1 = torch.tanh(torch.randn(B, C, H, W))
    img2 = torch.tanh(torch.randn(B, C, H, W) * 0.7 + 0.1)
    d["swd/image1"], d["swd/image2"] = img1.numpy(), img2.numpy()
    state = torch.get_rng_state()
    scores = swd.compute_swd(img1.clone(), img2.clone(), batch_size=bs)  # (the reference modifies its inputs in place)
    torch.set_rng_state(state)
    L = int(np.log2(min(H, W) // 16) + 1)
    counts = [((H >> l) - 6) * ((W >> l) - 6) for l in range(L)]
    for mb in range(-(-B // bs)):
        for which in range(2):
            for l in range(L):
                d[f"swd/inds/{mb}/{which}/{l}"] = torch.randperm(counts[l])[:npatch].numpy()
    for l in range(L):
        for r in range(reps):
            d[f"swd/dirs/{l}/{r}"] = torch.randn(C * 49, ndirs).numpy()
    for k, v in scores.items():
        d[f"swd/score/{k}"] = np.array(v)
    d["swd/meta"] = np.array([bs, npatch, reps, ndirs])
    path = os.path.join(HERE, "metrics.npz")
    np.savez_compressed(path, **d)
    print("wrote", path, f"{os.path.getsize(path) / 1024:.0f} KiB")


def make_covmmd_golden():
    """utils/metrics/cov_mmd_1nna.py (plain torch) loaded by path.  Its `from .distance import chamfer_distance,
    earth_mover_distance` pulls in two CUDA extensions that are JIT-built at import (not buildable
    here), so the package slot `.distance` gets a placeholder whose `chamfer_distance` is the reference's OWN CPU search
    - `nnsearch` of utils/metrics/distance/cd/chamfer_distance.cpp:39-66, compiled from its source by oracle/Makefile.ref
    into oracle/_ref/libref_cd.so, the code path its extension takes for CPU tensors (chamfer_distance.py:33-36) - and
    whose `earth_mover_distance` is absent (CUDA only).  Stored: `_compute_cov_mmd` / `_compute_nna` on crafted matrices
    (ties, k = 1 and 3, sqrt on / off) and `compute_cov_mmd_1nna(..., ("cd",))` end to end on seeded clouds, with the
    three pairwise matrices."""
    import ctypes
    so = os.path.join(os.path.dirname(os.path.dirname(HERE)), "oracle", "_ref", "libref_cd.so")
    lib = ctypes.CDLL(so)
    vp = ctypes.c_void_p

    def chamfer_distance(x1, x2):  # (b,n,3), (b,m,3) -> dist1 (b,n), dist2 (b,m): ChamferDistanceFunction.forward, CPU branch
        a, b = np.ascontiguousarray(x1.numpy(), np.float32), np.ascontiguousarray(x2.numpy(), np.float32)
        bs, n, m = a.shape[0], a.shape[1], b.shape[1]
        d1, i1 = np.zeros((bs, n), np.float32), np.zeros((bs, n), np.int32)
        d2, i2 = np.zeros((bs, m), np.float32), np.zeros((bs, m), np.int32)
        lib.ref_cd_nnsearch(bs, n, m, a.ctypes.data_as(vp), b.ctypes.data_as(vp), d1.ctypes.data_as(vp), i1.ctypes.data_as(vp))
        lib.ref_cd_nnsearch(bs, m, n, b.ctypes.data_as(vp), a.ctypes.data_as(vp), d2.ctypes.data_as(vp), i2.ctypes.data_as(vp))
        return torch.from_numpy(d1), torch.from_numpy(d2)

    pkg = types.ModuleType("ref_metrics")
    pkg.__path__ = [os.path.join(REF, "utils", "metrics")]
    sys.modules["ref_metrics"] = pkg
    dist = types.ModuleType("ref_metrics.distance")
    dist.chamfer_distance, dist.earth_mover_distance = chamfer_distance, None
    sys.modules["ref_metrics.distance"] = dist
    spec = importlib.util.spec_from_file_location("ref_metrics.cov_mmd_1nna",
                                                  os.path.join(REF, "utils", "metrics", "cov_mmd_1nna.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["ref_metrics.cov_mmd_1nna"] = mod
    spec.loader.exec_module(mod)

    torch.manual_seed(51)
    d = {"meta/torch": np.array(torch.__version__)}
    # crafted matrices: random, then quantised so that minima tie
    for tag, (nr, ng, quant) in {"rand": (9, 7, None), "ties": (8, 8, 0.25), "wide": (5, 12, 0.5)}.items():
        def sym(n):
            M = torch.rand(n, n)
            M = (M + M.t()) / 2
            M.fill_diagonal_(0.0)
            return M
        M_rr, M_gg, M_rg = sym(nr), sym(ng), torch.rand(nr, ng)
        if quant:
            M_rr, M_gg, M_rg = [(M / quant).round() * quant for M in (M_rr, M_gg, M_rg)]
        d[f"mat/{tag}/M_rr"], d[f"mat/{tag}/M_rg"], d[f"mat/{tag}/M_gg"] = M_rr.numpy(), M_rg.numpy(), M_gg.numpy()
        for k, v in mod._compute_cov_mmd(M_rg).items():
            d[f"mat/{tag}/covmmd/{k}"] = np.array(v, np.float64)
        for kk, sq in ((1, False), (3, False), (1, True)):
            for k, v in mod._compute_nna(M_rr, M_rg, M_gg, k=kk, sqrt=sq).items():
                d[f"mat/{tag}/nna_k{kk}_sqrt{int(sq)}/{k}"] = np.array(v, np.float64)
    # end to end on clouds (two populations of different spread, a few duplicates across the sets)
    pcs_ref = torch.randn(12, 96, 3) * 0.30
    pcs_gen = torch.randn(10, 96, 3) * 0.22 + 0.05
    pcs_gen[0] = pcs_ref[3]
    d["pcs_ref"], d["pcs_gen"] = pcs_ref.numpy(), pcs_gen.numpy()
    res = mod.compute_cov_mmd_1nna(pcs_gen, pcs_ref, 5, ("cd",), verbose=False)  # (batch 5: a ragged last batch)
    for k, v in res.items():
        d[f"e2e/{k}"] = np.array(v, np.float64)
    for tag, (a, b) in {"M_rr": (pcs_ref, pcs_ref), "M_rg": (pcs_ref, pcs_gen), "M_gg": (pcs_gen, pcs_gen)}.items():
        d[f"e2e_mat/{tag}"] = mod._pairwise_distance(a, b, 5, ("cd",), verbose=False)["cd"].numpy()
    path = os.path.join(HERE, "covmmd.npz")
    np.savez_compressed(path, **d)
    print("wrote", path, f"{os.path.getsize(path) / 1024:.0f} KiB", {k: round(v, 4) for k, v in res.items()})


def make_full_golden(name="full_dusty2", arch="dusty2", seed=20261, B=2, shape=(64, 1024), in_ch=512, ch_base=64,
                     ch_max=512, autocast=False):
    """One step of the REFERENCE's modules at the benchmark's full width (BASELINE configs 2-4: 64x1024, 512 latent,
    channels 64..512; dusty2 = the superset of the three archs), B = 2.  Parameters and inputs come from torch's CPU
    generator (tests/golden_util.full_inputs - regenerated identically on the checking side) and are LOADED into the
    reference's modules; the Gumbel noise is injected through the reference's own `GumbelSigmoid.fixed_noise` attribute;
    DiffAugment's draws are captured by replay as in make_step_golden.  Stored:
    digests (golden_util.digest) of every output, logit, augmented image, R1 gradient, parameter gradient and updated
    parameter, plus the scalars.
    autocast=True (`make_golden.py autocast` -> full_dusty2_autocast.npz): the SAME step - same parameters, inputs, noise and
    (asserted by the replay) the same DiffAugment draws - with the reference trainer's `enable_amp` regions
    (trainers/dcgan_amp.py:194-211, :228-232, :253-264) under torch.autocast("cpu", dtype=torch.bfloat16): what bfloat16
    autocast costs the REFERENCE ITSELF, tensor by tensor - the yardstick the timed mode's tolerances are held to
    (tests/test_oracle_golden.py, tests/test_gpu_configs.py).  No GradScaler: bfloat16 needs none, and the
    reference's scale / unscale pair is an identity on the values."""
    import contextlib
    sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
    from tests.golden_util import digest, full_inputs
    amp = (lambda: torch.autocast("cpu", dtype=torch.bfloat16)) if autocast else contextlib.nullcontext
    H, W = shape
    Gp, Dp, pol, mask_b, z, u = full_inputs(arch, in_ch, ch_base, ch_max, shape, B, seed)
    cfg = make_cfg(arch, in_ch, ch_base, ch_max, list(shape), True)
    torch.manual_seed(seed)
    G, D, G_ema = models.define_G(cfg), models.define_D(cfg), models.define_G(cfg)
    G.load_state_dict(Gp)
    sdD = D.state_dict()
    sdD.update(Dp)
    D.load_state_dict(sdD)
    G_ema.eval()
    ema_inplace(G_ema, G, 0.0)
    A = diff_augment.DiffAugment(policy=None)
    crit = GANLoss("nsgan")
    lr, b1, b2 = 0.002, 0.0, 0.99
    optG = torch.optim.Adam(G.parameters(), lr=lr, betas=(b1, b2))
    optD = torch.optim.Adam(D.parameters(), lr=lr, betas=(b1, b2))
    decay = 0.5 ** (B / (10 * 1000))
    lidar = _load_lidar()
    data = {"meta/arch": np.array(arch), "meta/shape": np.array(shape), "meta/in_ch": np.array(in_ch),
            "meta/ch_base": np.array(ch_base), "meta/ch_max": np.array(ch_max), "meta/B": np.array(B),
            "meta/seed": np.array(seed), "meta/lr": np.array(lr), "meta/ema_decay": np.array(decay),
            "meta/torch": np.array(torch.__version__), "meta/autocast": np.array("bfloat16" if autocast else "off")}

    def put(key, t):
        data[f"{key}/stats"], data[f"{key}/sample"] = digest(t)

    inv = lidar.Coordinate.invert_depth(Cfg(min_depth=0.9, max_depth=120.0,
                                            denormalize_minmax=lidar.Coordinate.denormalize_minmax,
                                            normalize_minmax=lidar.Coordinate.normalize_minmax), pol)
    m = mask_b.float()
    x_real = m * (inv * 2.0 - 1.0) + (1 - m) * (-1.0)
    put("x_real", x_real)

    # Gumbel noise through the reference's own injection point: GumbelSigmoid.forward adds `fixed_noise` instead of drawing
    # when the attribute is set (models/dusty.py:45-50, the hook utils/__init__.py:141-149 uses for evaluation); the noise
    # is the module's own formula (:33-36) applied to the regenerable uniforms
    eps = 1e-10
    ln = lambda u1, u2: -torch.log(torch.log(u1 + eps) / torch.log(u2 + eps) + eps)
    G.gumbel_pixel.fixed_noise = ln(u[0], u[1])
    G.gumbel_image.fixed_noise = ln(u[2], u[3])
    G.train()
    with amp():                                                           # trainers/dcgan_amp.py:194-211
        synth = G(latent=z)
        for k, v in synth.items():
            put(f"synth/{k}", v)
        x_real_aug, rp0 = run_and_capture(lambda: A(x_real), lambda: capture_aug(B, H, W, A.policy))
        x_real_aug = x_real_aug.detach().requires_grad_()
        x_fake_aug, rp1 = run_and_capture(lambda: A(synth["depth"]), lambda: capture_aug(B, H, W, A.policy))
        x_fake_aug = x_fake_aug.detach()
        put("x_real_aug", x_real_aug)
        put("x_fake_aug", x_fake_aug)
        y_real, y_fake = D(x_real_aug), D(x_fake_aug)
        data["y_real"], data["y_fake"] = y_real.detach().float().numpy().copy(), y_fake.detach().float().numpy().copy()
        sc = {"loss/D/output/real": y_real.mean().item(), "loss/D/output/fake": y_fake.mean().item()}
        loss_gan = crit(y_real, y_fake, "D")
    sc["loss/D/adversarial"] = loss_gan.item()
    (grads,) = torch.autograd.grad(outputs=y_real.sum(), inputs=[x_real_aug], create_graph=True, only_inputs=True)   # :216-224
    with amp():                                                           # :228-232
        r1 = (grads ** 2).sum(dim=[1, 2, 3]).mean()
        sc["loss/D/gradient_penalty"] = r1.item()
        loss_D = loss_gan + 0.5 * r1 + 0.0 * y_real.squeeze()[0]
    put("r1_grads", grads)
    optD.zero_grad(set_to_none=True)
    loss_D.backward()
    for k, p in D.named_parameters():
        put(f"grad_D/{k}", p.grad)
    optD.step()
    for p in D.parameters():
        p.requires_grad = False
    optG.zero_grad(set_to_none=True)
    with amp():                                                           # :253-264
        _, rp2 = run_and_capture(lambda: A(x_real), lambda: capture_aug(B, H, W, A.policy))
        x_fake_aug2, rp3 = run_and_capture(lambda: A(synth["depth"]), lambda: capture_aug(B, H, W, A.policy))
        y_fake2 = D(x_fake_aug2)
        data["y_fake2"] = y_fake2.detach().float().numpy().copy()
        loss_gan_g = crit(None, y_fake2, "G")
    sc["loss/G/adversarial"] = loss_gan_g.item()
    loss_gan_g.backward()
    for k, p in G.named_parameters():
        put(f"grad_G/{k}", p.grad)
    optG.step()
    ema_inplace(G_ema, G, decay)
    for tag, mod in (("G", G), ("D", D), ("G_ema", G_ema)):
        for k, v in mod.state_dict().items():
            put(f"after/{tag}/{k}", v)
    for j, rp in enumerate((rp0, rp1, rp2, rp3)):
        for k, v in rp.items():
            data[f"aug{j}/{k}"] = v.numpy()
    for k, v in sc.items():
        data[f"scalar/{k}"] = np.array(v, dtype=np.float64)
    path = os.path.join(HERE, f"{name}.npz")
    np.savez_compressed(path, **data)
    print("wrote", path, f"{os.path.getsize(path) / 1024:.0f} KiB", sc)


def make_pl_goldens():
    """path-length regularisation on (solver.loss.pl = 2, the value commented in configs/solver/nsgan_eqlr.yaml:21), two
    steps so the running baseline pl_ema is exercised; B = 4 -> B_pl = 2"""
    make_step_golden("none_pl", "none", True, seed=31, steps=2, B=4, pl=2.0)
    make_step_golden("dusty1_pl", "dusty1", True, seed=32, steps=2, B=4, pl=2.0)
    make_step_golden("dusty2_pl", "dusty2", True, seed=33, steps=2, B=4, pl=2.0)


def make_gan_mode_goldens():
    """the six config-reachable `solver.gan_mode`s besides nsgan (models/loss.py:42-61,70-85), one step each; the
    relativistic ones are the only metrics whose G phase reads D(real) (trainers/dcgan_amp.py:255,259)"""
    make_step_golden("none_wgan", "none", True, seed=21, steps=1, B=3, gan_mode="wgan")
    make_step_golden("none_lsgan", "none", True, seed=22, steps=1, B=3, gan_mode="lsgan")
    make_step_golden("dusty1_hinge", "dusty1", True, seed=23, steps=1, B=3, gan_mode="hinge")
    make_step_golden("dusty2_ragan", "dusty2", True, seed=24, steps=1, B=3, gan_mode="ragan")
    make_step_golden("dusty1_rahinge", "dusty1", True, seed=25, steps=1, B=3, gan_mode="rahinge", gp=0.0)
    make_step_golden("dusty2_ralsgan", "dusty2", True, seed=26, steps=1, B=3, gan_mode="ralsgan")


if __name__ == "__main__":
    torch.set_num_threads(4)
    if sys.argv[1:] == ["pl"]:
        make_pl_goldens()
        sys.exit(0)
    if sys.argv[1:] == ["metrics"]:
        make_metrics_golden()
        sys.exit(0)
    if sys.argv[1:] == ["covmmd"]:
        make_covmmd_golden()
        sys.exit(0)
    if sys.argv[1:] == ["full"]:
        torch.set_num_threads(8)
        make_full_golden()
        sys.exit(0)
    if sys.argv[1:] == ["autocast"]:
        torch.set_num_threads(8)
        make_full_golden(name="full_dusty2_autocast", autocast=True)
        sys.exit(0)
    if sys.argv[1:] == ["lidar"]:
        make_lidar_golden()
        sys.exit(0)
    if sys.argv[1:] == ["kitti"]:
        make_kitti_golden()
        sys.exit(0)
    if sys.argv[1:] == ["gan_modes"]:  # only the fixtures added after the first set (the others stay byte-identical)
        make_gan_mode_goldens()
        sys.exit(0)
    make_ops_golden()
    make_step_golden("none_ring", "none", True, seed=11)
    make_step_golden("dusty1_ring", "dusty1", True, seed=12)
    make_step_golden("dusty2_ring", "dusty2", True, seed=13)
    make_step_golden("dusty2_noring", "dusty2", False, seed=14)
    make_step_golden("dusty2_nogp", "dusty2", True, seed=15, gp=0.0)
    # a mid-size case closer to the production channel plan (still CPU-seconds)
    make_step_golden("dusty2_mid", "dusty2", True, seed=16, in_ch=32, ch_base=16, ch_max=64, shape=(64, 128), B=3,
                     steps=1)
    make_gan_mode_goldens()
    make_pl_goldens()
    make_lidar_golden()
    make_metrics_golden()
    make_covmmd_golden()
    make_full_golden()
