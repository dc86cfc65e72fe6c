"""Whole `Trainer.step` on the GPU (HIP path through the C ABI) against
  (1) the golden vectors generated from the reference's own modules (tiny nets, fp32, every tensor), and
  (2) the CPU oracle at production channel widths / 64x1024 (fp32 parity mode <= 1e-3 rel as BASELINE.json asks;
      bf16 mode within its own stated tolerance).
Randomness (z, Gumbel noise, the four DiffAugment draws) is injected so both sides see identical batches.
"""
import copy

import numpy as np
import pytest
import torch

from oracle import dusty_oracle as O
from tests.golden_util import PL_CASES, STEP_CASES, load, meta_pl, rel_l2, step_rand, sub

pytestmark = pytest.mark.gpu
DEV = "cuda"


def make_trainer(arch, ring, shape, in_ch, ch_base, ch_max, B, gp=1.0, amp=False, n_acc=1, gan_mode="nsgan", pl=0.0, gpu=0):
    from dusty_gan_amd.trainers.dcgan_amp import Trainer
    from dusty_gan_amd.utils.config import load_config
    model = {"none": "dcgan_eqlr", "dusty1": "dusty1_dcgan_eqlr", "dusty2": "dusty2_dcgan_eqlr"}[arch]
    cfg = load_config([f"model={model}", "dataset=synthetic", f"dataset.shape=[{shape[0]},{shape[1]}]",
                       f"model.gen.in_ch={in_ch}", f"model.gen.ch_base={ch_base}", f"model.gen.ch_max={ch_max}",
                       f"model.dis.ch_base={ch_base}", f"model.dis.ch_max={ch_max}", f"model.ring={str(ring).lower()}",
                       f"solver.batch_size={B * n_acc}", f"solver.loss.gp={gp}", f"enable_amp={str(amp).lower()}",
                       f"solver.num_accumulation={n_acc}", "dataset.pool=1", f"solver.gan_mode={gan_mode}",
                       f"solver.loss.pl={pl}"])
    return Trainer(cfg, {"gpu": gpu, "ngpus": 1, "batch_size": B, "num_workers": 0})


def grads_by_name(optim):
    if getattr(optim, "regen_grad", None) is not None:
        optim.regen_grad()  # Proj.weight's gradient when the optimizer kernel formed it on the fly (bf16, one GPU)
    st = optim.store
    views = optim._param_views(st.grad)
    return {k: v.detach().cpu() for (k, _), v in zip(optim.net.named_parameters(), views)}


@pytest.mark.parametrize("case", STEP_CASES + PL_CASES)
def test_step_matches_reference_golden(case):
    g = load("step_" + case)
    arch, ring = str(g["meta/arch"]), bool(g["meta/ring"])
    B, steps = int(g["meta/B"]), int(g["meta/steps"])
    tr = make_trainer(arch, ring, tuple(int(v) for v in g["meta/shape"]), int(g["meta/in_ch"]),
                      int(g["meta/ch_base"]), int(g["meta/ch_max"]), B, gp=float(g["meta/gp"]),
                      gan_mode=str(g["meta/gan_mode"]), pl=meta_pl(g))
    tr.G.load_state_dict(sub(g, "init/G"))
    tr.D.load_state_dict(sub(g, "init/D"))
    tr.G_ema.load_state_dict(sub(g, "init/G"))
    assert abs(tr.ema_decay - float(g["meta/ema_decay"])) < 1e-12
    tol = 1e-3  # north_star: within 1e-3 rel of the reference (fp32 parity mode; observed ~1e-5)
    for it in range(steps):
        pre = f"s{it}"
        pol = torch.from_numpy(g[f"{pre}/pol"])
        mask = torch.from_numpy(g[f"{pre}/mask"])
        x_real, m_real = tr.fetch_reals({"depth": pol, "mask": mask})
        assert rel_l2(x_real.cpu(), g[f"{pre}/x_real"]) < 1e-5
        rand = step_rand(g, it)
        tr.optimize_D(reals=[(x_real, m_real)], rands=[rand])
        synth = {k: v.detach().cpu().clone() for k, v in tr._mb[0]["synth"].items()}
        gD = grads_by_name(tr.optim_D)
        scal = tr.optimize_G()
        gG = grads_by_name(tr.optim_G)
        sc = scal.cpu().tolist()
        got = {"loss/D/output/real": sc[0], "loss/D/output/fake": sc[1], "loss/D/adversarial": sc[2],
               "loss/D/gradient_penalty": sc[3], "loss/G/adversarial": sc[4], "loss/G/path_length/baseline": sc[5],
               "loss/G/path_length": sc[6]}
        for k, v in sub(g, f"{pre}/scalar").items():
            assert abs(got[k] - float(v)) <= tol * max(1.0, abs(float(v))), (k, got[k], float(v))
        for k, v in sub(g, f"{pre}/synth").items():
            if k == "mask":
                assert (synth[k] != v).float().mean() < 1e-3
            else:
                assert rel_l2(synth[k], v) < tol, k
        # relativistic-average losses: sum_i dLoss/dy_i == 0 identically, so the final conv's bias gradient is zero and
        # the reference's stored value is fp32 rounding noise (~1e-7) whose SIGN Adam then turns into a +-lr update
        noise = {"5.module.bias"} if str(g["meta/gan_mode"]).startswith("ra") else set()
        for k, v in sub(g, f"{pre}/grad_D").items():
            if k in noise:
                assert float(v.abs().max()) < 1e-6 and float(gD[k].abs().max()) < 1e-6
                continue
            assert rel_l2(gD[k], v) < tol, ("grad_D", k)
        if meta_pl(g) > 0:
            assert rel_l2(tr._pl_dz.cpu(), g[f"{pre}/pl/grads_z"]) < tol
        for k, v in sub(g, f"{pre}/grad_G").items():
            assert rel_l2(gG[k], v) < tol, ("grad_G", k)
        for tag, net in (("G", tr.G), ("D", tr.D), ("G_ema", tr.G_ema)):
            sd = net.state_dict()
            for k, v in sub(g, f"{pre}/after/{tag}").items():
                if tag == "D" and noise and f"{pre}/grad_D/{k}" in g.files:
                    # same story per channel: a bias whose units have the same slope for every sample gets
                    # (sum_i dy_i) * const == 0; compare where the gradient is not rounding noise, bound the rest
                    gref = torch.from_numpy(g[f"{pre}/grad_D/{k}"])
                    live = gref.abs() > 1e-4 * gref.abs().max() if k not in noise else torch.zeros_like(gref, dtype=torch.bool)
                    assert float((sd[k].cpu() - v).abs().max()) <= 2 * float(g["meta/lr"]) * 1.001, (tag, k)
                    if live.any():
                        assert rel_l2(sd[k].cpu()[live], v[live]) < tol, (tag, k)
                    continue
                # (path-length cases: a bias gradient element of the order of Adam's eps moves its step with the
                # rounding noise - accept 2.5 % of one lr step on single elements, as tests/test_oracle_golden.py does)
                assert rel_l2(sd[k].cpu(), v) < tol or (meta_pl(g) > 0 and float((sd[k].cpu() - v).abs().max()) < 5e-5), (tag, k)
    # Adam state round trip in torch.optim.Adam's format
    sdo = tr.optim_D.state_dict()
    names = [k for k, _ in tr.D.named_parameters()]
    for i, k in enumerate(names):
        if k in noise:
            continue
        assert rel_l2(sdo["state"][i]["exp_avg_sq"], g[f"final/optim_D/{k}/exp_avg_sq"]) < tol
        assert int(sdo["state"][i]["step"]) == steps


def oracle_state(tr):
    G = {k: v.detach().cpu().clone() for k, v in tr.G.state_dict().items()}
    D = {k: v.detach().cpu().clone() for k, v in tr.D.state_dict().items() if not k.endswith("kernel")}
    return G, D


def sync_D(tr, D):
    """give the engine the oracle's discriminator (parameters only; the BlurVH kernels are buffers of the module)"""
    sd = tr.D.state_dict()
    sd.update({k: v.detach().clone() for k, v in D.items()})
    tr.D.load_state_dict(sd)


def run_both(arch, shape, in_ch, ch_base, ch_max, B, amp, steps=1, seed=0, pl=0.0, emulate=False, sync=False):
    """One trainer and the oracle on the same parameters, batches and randomness.
    emulate: the oracle rounds to bf16 where the engine's bf16 mode stores bf16 (oracle `_Emu`).
    sync:    between the two phases the engine takes over the ORACLE's updated discriminator, so that both G phases
             differentiate the same D (the first Adam step at beta1 = 0 is lr * sign(g): D weights whose gradient is
             rounding noise otherwise land 2 lr apart in the two implementations and the G gradients inherit it)."""
    torch.manual_seed(4321 + seed)  # the nets draw their N(0,1) init from torch's global generator
    tr = make_trainer(arch, True, shape, in_ch, ch_base, ch_max, B, amp=amp, pl=pl)
    G, D = oracle_state(tr)
    G_ema = {k: v.clone() for k, v in G.items()}
    oG, oD = O.new_optim_state(G), O.new_optim_state(D)
    cfg = O.StepConfig(arch=arch, ema_decay=tr.ema_decay, w_pl=pl, emulate_bf16=emulate)
    pl_ema = torch.tensor(0.0)
    gen = torch.Generator().manual_seed(seed)
    H, W = shape
    res = []
    for it in range(steps):
        pol = torch.rand(B, 1, H, W, generator=gen)
        mask = torch.rand(B, 1, H, W, generator=gen) > 0.15
        pol = pol * mask
        rand = {"z": torch.randn(B, in_ch, generator=gen),
                "noise": {"pixel": O.logistic_noise(torch.rand(B, 1, H, W, generator=gen), torch.rand(B, 1, H, W, generator=gen)),
                          "image": O.logistic_noise(torch.rand(B, 1, 1, 1, generator=gen), torch.rand(B, 1, 1, 1, generator=gen))},
                "aug": [O.draw_augment_params(B, H, W, gen) for _ in range(4)]}
        if pl > 0:
            Bp = B // 2
            rand["pl"] = {"z": torch.randn(Bp, in_ch, generator=gen),
                          "noise": {"pixel": O.logistic_noise(torch.rand(Bp, 1, H, W, generator=gen), torch.rand(Bp, 1, H, W, generator=gen)),
                                    "image": O.logistic_noise(torch.rand(Bp, 1, 1, 1, generator=gen), torch.rand(Bp, 1, 1, 1, generator=gen))},
                          "y": torch.randn(Bp, 1, H, W, generator=gen), "pl_ema": pl_ema.clone()}
        x_real_cpu, _ = O.fetch_reals(pol, mask)
        sc_ref, ex = O.train_step(G, D, G_ema, oG, oD, it + 1, cfg, x_real_cpu, rand, return_grads=True)
        if pl > 0:
            pl_ema = ex["pl_ema"]
        x_real, m_real = tr.fetch_reals({"depth": pol, "mask": mask})
        tr.optimize_D(reals=[(x_real, m_real)], rands=[rand])
        synth = {k: v.detach().cpu().clone() for k, v in tr._mb[0]["synth"].items()}
        gD = grads_by_name(tr.optim_D)
        if sync:
            ex["D_engine_after"] = {k: v.detach().cpu().clone() for k, v in tr.D.state_dict().items()}
            sync_D(tr, D)
        scal = tr.optimize_G().cpu().tolist()
        gG = grads_by_name(tr.optim_G)
        res.append((sc_ref, ex, synth, gD, gG, scal))
    return tr, (G, D, G_ema), res


# fp32 parity mode, gradients at full width (see the docstring of test_step_fp32_vs_oracle_full_width_64x1024): measured on
# MI355X 2.2e-6 (D) / 1.7e-5 (G) with no flipped unit at B = 2, up to 1.5e-3 at B = 8 ... 32 where a few units flip
# (tests/test_gpu_configs.py prints them): held to 5e-3 rel-L2 AND cosine 0.99999 (round 4: 2e-2 / 0.9999, 20 x looser than
# anything measured - a 1 % defect of one layer would have passed)
FP32_GRAD_TOL, FP32_GRAD_COS = 5e-3, 0.99999


def _cos(a, b):
    a, b = a.flatten().double(), b.flatten().double()
    return float((a @ b) / (a.norm() * b.norm()))


def test_step_fp32_vs_oracle_full_width_64x1024():
    """config 2/3/4 shape and channel plan (64x1024, 512 latent, ch 64..512), B=2, fp32 parity mode.
    Outputs, logits and losses: <= 1e-4 rel (north_star asks 1e-3; measured ~1e-6).
    Gradients: every tensor is the SAME linear map of the same upstream given the same leaky-relu masks, and the
    per-op tests pin that to 1e-4; end to end a handful of the ~4M units whose pre-activation is within fp32
    rounding of zero (measured 3 of 3.9M, |pre| < 2e-8: scripts/diag_flips.py) take the other slope in the two
    implementations.  One flipped unit is an O(1) change of that unit's contribution, i.e. ~sqrt(#flips/#units)
    ~ 1e-3 rel-L2 on every gradient downstream.  So gradients are held to 5e-3 rel-L2 AND cosine >= 0.99999 (measured in
    this case: D 2e-6, G 2e-5 - no unit flipped; with flips up to 1.5e-3, tests/test_gpu_configs.py)."""
    tr, (G, D, G_ema), res = run_both("dusty2", (64, 1024), 512, 64, 512, 2, amp=False)
    sc_ref, ex, synth, gD, gG, scal = res[0]
    tol = 1e-4
    keys = ["loss/D/output/real", "loss/D/output/fake", "loss/D/adversarial", "loss/D/gradient_penalty",
            "loss/G/adversarial"]
    for k, v in zip(keys, scal):
        assert abs(v - sc_ref[k]) <= tol * max(1.0, abs(sc_ref[k])), (k, v, sc_ref[k])
    for k in ("depth", "depth_orig", "confidence"):
        assert rel_l2(synth[k], ex["synth"][k]) < tol, k
    assert (synth["mask"] != ex["synth"]["mask"]).float().mean() < 1e-5
    worst = {}
    for name, got, ref in (("grad_D", gD, ex["grad_D"]), ("grad_G", gG, ex["grad_G"])):
        for k, v in ref.items():
            r, c = rel_l2(got[k], v), _cos(got[k], v)
            worst[name] = max(worst.get(name, 0.0), r)
            worst[name + "_cos"] = min(worst.get(name + "_cos", 1.0), c)
    print("PARITY full-width fp32 B=2", {k: float(f"{v:.7g}") for k, v in worst.items()})
    for name in ("grad_D", "grad_G"):
        assert worst[name] < FP32_GRAD_TOL and worst[name + "_cos"] > FP32_GRAD_COS, (name, worst)
    tol = 1e-3
    sd = tr.G.state_dict()
    for k, v in G.items():
        if k != "drop_const":
            assert rel_l2(sd[k].cpu(), v) < tol, k
    sde = tr.G_ema.state_dict()
    for k, v in G_ema.items():
        if k != "drop_const":
            assert rel_l2(sde[k].cpu(), v) < tol, k


def test_step_bf16_vs_oracle_mid():
    """bf16 storage / fp32 accumulate mode against the fp32 oracle.  Stated tolerances:
      outputs / logits  rel-L2 <= 2e-2   (measured 6e-3)
      losses            <= 1e-2 absolute (measured 1e-3)
      gradients         rel-L2 <= 2.5e-1 and cosine >= 0.97 (measured: D 4-8e-2, G 1.4-2.0e-1).
    Reference point (scripts/autocast_gap.py, same nets / batch): stock PyTorch bf16 autocast -- what the
    reference's `enable_amp` does, with bf16 for fp16 -- is 6e-3 from its own fp32 on the outputs, 3-7e-2 on the D
    gradients and 1.0-1.7e-1 on the G gradients.  The gradient gap is NOT accumulation error: bf16 inputs move
    pre-activations by ~0.4 %, which flips the leaky-relu slope (1 <-> 0.2) of the ~0.4 % of units that sit at zero;
    each flipped unit is off by 80 %, i.e. ~5 % rel-L2 per layer, compounding over the 4-9 layers a gradient crosses.
    The fp32 mode is the <= 1e-3 parity mode."""
    tr, _, res = run_both("dusty2", (64, 256), 128, 64, 256, 4, amp=True)
    sc_ref, ex, synth, gD, gG, scal = res[0]
    for k in ("depth_orig", "confidence"):
        assert rel_l2(synth[k], ex["synth"][k]) < 2e-2, k
    assert (synth["mask"] != ex["synth"]["mask"]).float().mean() < 1e-2
    keys = ["loss/D/output/real", "loss/D/output/fake", "loss/D/adversarial", "loss/D/gradient_penalty",
            "loss/G/adversarial"]
    for k, v in zip(keys, scal):
        assert abs(v - sc_ref[k]) < 1e-2, (k, v, sc_ref[k])

    def cos(a, b):
        a, b = a.flatten().double(), b.flatten().double()
        return float((a @ b) / (a.norm() * b.norm()))
    for name, got, ref in (("D", gD, ex["grad_D"]), ("G", gG, ex["grad_G"])):
        for k, v in ref.items():
            if v.abs().max() > 0:
                assert rel_l2(got[k], v) < 2.5e-1, (name, k, rel_l2(got[k], v))
                assert cos(got[k], v) > 0.97, (name, k, cos(got[k], v))


def test_gradient_accumulation_equals_full_batch():
    """num_accumulation=2 with micro-batches of B/2 == one batch of B (reference: utils/context_manager.py:21-35
    + loss / num_accumulation, trainers/dcgan_amp.py:234,308)."""
    arch, shape, nz, cb, cm, B = "dusty1", (32, 64), 8, 4, 16, 4
    tr1 = make_trainer(arch, True, shape, nz, cb, cm, B)
    tr2 = make_trainer(arch, True, shape, nz, cb, cm, B // 2, n_acc=2)
    tr2.G.load_state_dict(tr1.G.state_dict())
    tr2.D.load_state_dict(tr1.D.state_dict())
    tr2.G_ema.load_state_dict(tr1.G_ema.state_dict())
    gen = torch.Generator().manual_seed(5)
    H, W = shape
    x = torch.rand(B, 1, H, W, generator=gen) * 2 - 1
    m = torch.ones(B, 1, H, W)
    rand = {"z": torch.randn(B, nz, generator=gen),
            "noise": {"pixel": O.logistic_noise(torch.rand(B, 1, H, W, generator=gen), torch.rand(B, 1, H, W, generator=gen))},
            "aug": [O.draw_augment_params(B, H, W, gen) for _ in range(4)]}

    def half(r, s):
        return {"z": r["z"][s], "noise": {k: v[s] for k, v in r["noise"].items()},
                "aug": [{k: v[s] for k, v in rp.items()} for rp in r["aug"]]}
    xd = x.to(DEV)
    md = m.to(DEV)
    tr1.optimize_D(reals=[(xd, md)], rands=[rand])
    h = B // 2
    tr2.optimize_D(reals=[(xd[:h].contiguous(), md[:h]), (xd[h:].contiguous(), md[h:])],
                   rands=[half(rand, slice(0, h)), half(rand, slice(h, B))])
    g1, g2 = grads_by_name(tr1.optim_D), grads_by_name(tr2.optim_D)
    for k in g1:
        assert rel_l2(g2[k], g1[k]) < 1e-4, k
    tr1.optimize_G()
    tr2.optimize_G()
    g1, g2 = grads_by_name(tr1.optim_G), grads_by_name(tr2.optim_G)
    for k in g1:
        assert rel_l2(g2[k], g1[k]) < 1e-4, k


def test_gradient_accumulation_matches_the_oracle_on_the_full_batch():
    """num_accumulation = 2 against the ORACLE (not only against the engine's own full-batch step): two micro-batches of
    B / 2 must give the gradients, losses and updated parameters `O.train_step` gives on the B samples in one batch - the
    reference's schedule (utils/context_manager.py:21-35 with loss / num_accumulation, trainers/dcgan_amp.py:234,308)."""
    arch, shape, nz, cb, cm, B = "dusty2", (32, 64), 8, 4, 16, 4
    torch.manual_seed(808)
    tr = make_trainer(arch, True, shape, nz, cb, cm, B // 2, n_acc=2)
    G, D = oracle_state(tr)
    G_ema = {k: v.clone() for k, v in G.items()}
    gen = torch.Generator().manual_seed(6)
    H, W = shape
    pol = torch.rand(B, 1, H, W, generator=gen)
    mask = torch.rand(B, 1, H, W, generator=gen) > 0.15
    pol = pol * mask
    rand = {"z": torch.randn(B, nz, generator=gen),
            "noise": {"pixel": O.logistic_noise(torch.rand(B, 1, H, W, generator=gen), torch.rand(B, 1, H, W, generator=gen)),
                      "image": O.logistic_noise(torch.rand(B, 1, 1, 1, generator=gen), torch.rand(B, 1, 1, 1, generator=gen))},
            "aug": [O.draw_augment_params(B, H, W, gen) for _ in range(4)]}
    x_cpu, _ = O.fetch_reals(pol, mask)
    cfg = O.StepConfig(arch=arch, ema_decay=tr.ema_decay)
    sc_ref, ex = O.train_step(G, D, G_ema, O.new_optim_state(G), O.new_optim_state(D), 1, cfg, x_cpu, rand, return_grads=True)
    h = B // 2

    def half(s):
        return {"z": rand["z"][s], "noise": {k: v[s] for k, v in rand["noise"].items()},
                "aug": [{k: v[s] for k, v in rp.items()} for rp in rand["aug"]]}
    reals = []
    for s in (slice(0, h), slice(h, B)):
        reals.append(tr.fetch_reals({"depth": pol[s], "mask": mask[s]}))
    scal = tr.step(1, reals=reals, rands=[half(slice(0, h)), half(slice(h, B))])
    for k, v in scal.items():
        assert abs(v - sc_ref[k]) <= 1e-3 * max(1.0, abs(sc_ref[k])), (k, v, sc_ref[k])
    gG = grads_by_name(tr.optim_G)
    for k, v in ex["grad_G"].items():
        assert rel_l2(gG[k], v) < 1e-3, ("grad_G", k, rel_l2(gG[k], v))
    for tag, net, ref in (("G", tr.G, G), ("D", tr.D, D), ("G_ema", tr.G_ema, G_ema)):
        sd = net.state_dict()
        for k, v in ref.items():
            if k.endswith("kernel"):
                continue
            assert rel_l2(sd[k].cpu(), v) < 1e-3, (tag, k, rel_l2(sd[k].cpu(), v))


def test_accumulated_step_replays_from_a_graph(monkeypatch):
    """num_accumulation = 2 on the device-resident synthetic pool: the whole accumulated step (both micro-batches' fetches,
    draws, D and G phases, one optimizer step each) is captured into ONE hipGraph and replayed - the same parameters after
    2 eager + 3 replayed steps as 5 eager steps from the same seeds.  (Round 3 launched such steps eagerly: ~5 ms of host
    time per step.)"""
    def run(graph):
        monkeypatch.setenv("DUSTY_GAN_GRAPH", "1" if graph else "0")
        torch.manual_seed(515)
        tr = make_trainer("dusty1", True, (64, 256), 128, 64, 256, 8, amp=True, n_acc=2)
        sc = [dict(tr.step(i).items()) for i in range(5)]
        assert (tr._graph is not None) == graph
        if graph:
            assert sum(isinstance(g, torch.cuda.CUDAGraph) for g in tr._graph) == 1
            assert tr.batches_drawn == 10     # the host loader moved on by num_accumulation batches per step
            # ... and so did the DEVICE-side pool index the captured fetches read (n_acc per replay: round-4 advice - an
            # off-by-one micro-batch in the captured fetches would pass every tolerance below)
            from dusty_gan_amd import _lib as L
            with L.Counters.bind(tr.counters):
                L.Counters.flush_if(tr._pool_ctr)
            assert int(tr._pool_ctr) == tr.batches_drawn == tr._pool_host, (int(tr._pool_ctr), tr.batches_drawn, tr._pool_host)
        return tr, sc
    a, sa = run(True)
    b, sb = run(False)
    for net in ("G", "D", "G_ema"):
        fa, fb = getattr(a, net).store.flat.cpu(), getattr(b, net).store.flat.cpu()
        # (this 64x256 net has layers too narrow for the ping-pong conv - coarse rows of 32 columns - whose kernels added their
        #  bias gradients with float atomics until round 6: the two launch forms ended 8e-4 / 4e-4 / 2e-6 apart, the run-to-run
        #  noise of such a step.  Now no sum on this path depends on the order of its terms: the same bits.)
        assert torch.equal(fa, fb), (net, rel_l2(fa, fb))
    for x, y in zip(sa, sb):
        for k in x:
            assert abs(x[k] - y[k]) <= 1e-6 * max(1.0, abs(y[k])), (k, x[k], y[k])


@pytest.mark.parametrize("arch,shape,nz,lo,hi,B,amp,x3,n_acc", [
    ("dusty2", (64, 1024), 128, 64, 512, 8, False, False, 1),
    ("dusty1", (64, 256), 128, 64, 256, 8, False, False, 2),
    ("dusty2", (32, 64), 8, 4, 16, 2, False, False, 1),
    ("dusty2", (32, 64), 8, 4, 16, 2, True, False, 1),
    ("none", (32, 128), 16, 8, 32, 4, False, False, 1),
    ("dusty2", (64, 256), 128, 64, 256, 8, False, True, 1),
], ids=["fp32-64x1024", "fp32-64x256-acc2", "fp32-narrow", "bf16-narrow", "fp32-narrow-none", "fp32x3-64x256"])
def test_two_runs_off_the_timed_shapes_are_bit_identical(monkeypatch, arch, shape, nz, lo, hi, B, amp, x3, n_acc):
    """SURVEY section 5 "determinism check by double-run" for what the timed configuration does not run (round 6): the exact
    fp32 mode (lock-step persistent / one-tile MFMA convs, register-staged MFMA weight gradients), narrow nets at the
    golden-vector sizes (direct kernels), a 64x256 net whose coarse maps are too narrow for the ping-pong conv, with and
    without accumulation - two trainers from one seed hold IDENTICAL bits in G and D after three steps.  Before round 6 every
    tensor of every case differed (scripts/probes/two_runs_bits.py with the round-5 library: 1e-9 ... 1e-2): split-K weight
    gradients and bias-gradient sums of these kernels were float atomics in arrival order."""
    monkeypatch.setenv("DUSTY_GAN_FP32_SPLIT", "1" if x3 else "0")

    def run():
        torch.manual_seed(99)
        tr = make_trainer(arch, True, shape, nz, lo, hi, B, amp=amp, n_acc=n_acc)
        for i in range(3):
            tr.step(i)
        torch.cuda.synchronize()
        return tr
    a, b = run(), run()
    for net in ("G", "D", "G_ema"):
        sa, sb = getattr(a, net).store, getattr(b, net).store
        bad = [k for k, sg in sa.seg.items()
               if not torch.equal(sa.flat[sg.off:sg.off + sg.numel], sb.flat[sg.off:sg.off + sg.numel])]
        assert not bad, (net, bad)


def test_checkpoint_roundtrip_and_generate(tmp_path):
    tr = make_trainer("dusty2", True, (32, 64), 8, 4, 16, 2)
    s = tr.step(1)
    assert set(s.keys()) == {"loss/D/output/real", "loss/D/output/fake", "loss/D/adversarial",
                             "loss/D/gradient_penalty", "loss/G/adversarial"}
    assert all(np.isfinite(v) for v in s.values())
    path = tr.save_models("0000000002", 2, directory=str(tmp_path))
    sd = torch.load(path, map_location="cpu")
    # the reference's keys (trainers/dcgan_amp.py:395-409) + the position record it omits
    assert set(sd.keys()) == {"step", "G", "D", "G_ema", "optim_G", "optim_D", "pl_ema", "resume_state"}
    assert sd["G"]["backbone.3.1.module.weight"].shape == (8, 4, 4, 4)  # ConvTranspose2d layout (Cin,Cout,4,4)
    assert sd["D"]["1.1.module.weight"].shape == (4, 2, 4, 4)            # Conv2d layout (Cout,Cin,4,4)
    tr2 = make_trainer("dusty2", True, (32, 64), 8, 4, 16, 2)
    tr2.G.load_state_dict(sd["G"]); tr2.D.load_state_dict(sd["D"]); tr2.G_ema.load_state_dict(sd["G_ema"])
    tr2.optim_G.load_state_dict(sd["optim_G"]); tr2.optim_D.load_state_dict(sd["optim_D"])
    assert torch.equal(tr2.G.store.flat.cpu(), tr.G.store.flat.cpu())
    assert torch.equal(tr2.D.store.v.cpu(), tr.D.store.v.cpu())
    out = tr.generate(ema=True)
    assert out["depth"].shape == (2, 1, 32, 64) and out["mask"].shape == (2, 2, 32, 64)
    assert torch.isfinite(out["depth"]).all()
    with pytest.raises(RuntimeError):
        tr.G(torch.zeros(2, 8))  # CPU input: no fallback
    # utils.setup (reference utils/__init__.py:116-160): the evaluation scripts' way back from a checkpoint
    from dusty_gan_amd.utils import setup
    from dusty_gan_amd.utils.config import dump_config
    from dusty_gan_amd.utils.lidar import postprocess
    cfg_path = str(tmp_path / "config.yaml")
    dump_config(tr.cfg, cfg_path)
    cfg, G, lidar, device = setup(path, cfg_path, ema=True, fix_noise=True)
    assert device.type == "cuda" and cfg.dataset.shape == [32, 64] and not G.training
    assert torch.equal(G.store.flat.cpu(), tr.G_ema.store.flat.cpu())
    z = torch.randn(3, 8, device=device)
    o1, o2 = G(z), None
    o1 = {k: v.clone() for k, v in o1.items()}
    o2 = G(z)
    for k in o1:  # fix_noise: the Gumbel noise of the first call is reused, so the same latent gives the same scan
        assert torch.equal(o1[k], o2[k]), k
    m = [mod for mod in G.modules() if type(mod).__name__ == "GumbelSigmoid"]
    assert len(m) == 2 and m[0].fixed_noise is not None and m[0].fixed_noise.shape == (1, 1, 32, 64)
    out = postprocess(o2, lidar)
    assert "points" in out and out["points"].shape == (3, 3, 32, 64)


def test_path_length_regulariser_vs_oracle():
    """solver.loss.pl > 0 (reference :268-306) beyond the golden fixtures: a wider net, two steps (the running baseline
    pl_ema carries over), fp32 mode against the oracle's autograd double backward; then bf16 mode, where the penalty
    and the baseline must stay within a few per cent of the fp32 oracle"""
    tr, (G, D, G_ema), res = run_both("dusty2", (64, 256), 64, 16, 64, 4, amp=False, steps=2, pl=2.0)
    for sc_ref, ex, synth, gD, gG, scal in res:
        for k, v in (("loss/G/path_length/baseline", scal[5]), ("loss/G/path_length", scal[6])):
            assert abs(v - sc_ref[k]) <= 1e-3 * max(1.0, abs(sc_ref[k])), (k, v, sc_ref[k])
        for k, v in ex["grad_G"].items():
            assert rel_l2(gG[k], v) < 2e-2 and _cos(gG[k], v) > 0.9999, (k, rel_l2(gG[k], v))
    assert abs(float(tr.pl_ema) - res[-1][0]["loss/G/path_length/baseline"]) < 1e-5
    tr16, _, res16 = run_both("dusty2", (64, 256), 64, 16, 64, 4, amp=True, steps=1, pl=2.0)
    sc_ref, ex, synth, gD, gG, scal = res16[0]
    assert abs(scal[5] - sc_ref["loss/G/path_length/baseline"]) <= 5e-2 * abs(sc_ref["loss/G/path_length/baseline"])
    assert abs(scal[6] - sc_ref["loss/G/path_length"]) <= 1e-1 * abs(sc_ref["loss/G/path_length"])
    for k, v in ex["grad_G"].items():
        if v.numel() >= 1024:  # (a 2-element head-bias gradient is all bf16 slope-flip noise)
            assert _cos(gG[k], v) > 0.9, (k, _cos(gG[k], v))
    # full-width latent grid (64x1024 -> Proj has 32 768 rows): d(sum x y)/dz goes through the split-K form of grad_z
    _, _, resw = run_both("dusty1", (64, 1024), 64, 16, 128, 2, amp=False, steps=1, pl=2.0)
    sc_ref, ex, synth, gD, gG, scal = resw[0]
    assert abs(scal[6] - sc_ref["loss/G/path_length"]) <= 1e-3 * max(1.0, abs(sc_ref["loss/G/path_length"]))
    for k, v in ex["grad_G"].items():
        assert rel_l2(gG[k], v) < 2e-2 and _cos(gG[k], v) > 0.9999, (k, rel_l2(gG[k], v))
    # checkpoint carries the baseline; unknown weight keys still raise
    assert float(tr.state(1)["pl_ema"]) == float(tr.pl_ema)


@pytest.mark.parametrize("gan_mode", ["nsgan", "rahinge", "nsgan+pl"])
def test_graph_replay_matches_eager_launches(monkeypatch, gan_mode):
    """the hipGraph-captured step (device-resident Philox / Adam counters) trains exactly like the eager launch
    sequence: same seeds -> same parameters after 5 iterations (fp32; atomics make the last bits differ)"""
    def run(graph):
        monkeypatch.setenv("DUSTY_GAN_GRAPH", "1" if graph else "0")
        torch.manual_seed(2024)
        tr = make_trainer("dusty2", True, (32, 64), 8, 4, 16, 4, gan_mode=gan_mode.split("+")[0],
                          pl=2.0 if gan_mode.endswith("+pl") else 0.0)
        sc = [dict(tr.step(i).items()) for i in range(5)]
        assert ("loss/G/path_length" in sc[0]) == gan_mode.endswith("+pl")
        assert (tr._graph is not None) == graph
        return tr, sc
    a, sa = run(True)
    b, sb = run(False)
    assert a.optim_G.step_count == b.optim_G.step_count == 5
    for net in ("G", "D", "G_ema"):
        fa, fb = getattr(a, net).store.flat.cpu(), getattr(b, net).store.flat.cpu()
        assert rel_l2(fa, fb) < 1e-5, net
    for x, y in zip(sa, sb):
        for k in x:
            assert abs(x[k] - y[k]) < 1e-4 * max(1.0, abs(y[k])), k


def test_eager_draw_between_graph_replays(monkeypatch):
    """An eager draw from the trainer's generator between two steps (validation() / generate() call sample_latents)
    queues its counter advance on the host; the replayed graph reads the DEVICE counter.  The advance must be applied
    before the capture and before every replay: otherwise the step re-draws the evaluation latents (stale counter) and,
    when the advance is pending at capture time, bakes it into the graph.  Same seeds, same interleaving -> the graph run
    draws exactly what the eager run draws."""
    def run(graph):
        monkeypatch.setenv("DUSTY_GAN_GRAPH", "1" if graph else "0")
        torch.manual_seed(77)
        tr = make_trainer("none", True, (32, 64), 8, 4, 16, 4)
        zs, evs = [], []
        for i in range(7):
            if i >= 2:  # i == 2: pending at capture time; i >= 3: pending before a replay
                evs.append(tr.sample_latents(4).clone())
            tr.step(i)
            zs.append(tr._g_engines()[0].zT.float().view(4, -1).clone())
            if i >= 2:
                assert not torch.equal(zs[-1], evs[-1]), i  # the step did not re-draw the evaluation latents
        assert (tr._graph is not None) == graph
        return tr, zs, evs
    a, za, ea = run(True)
    b, zb, eb = run(False)
    for i, (x, y) in enumerate(zip(za, zb)):
        assert torch.equal(x, y), i
    for x, y in zip(ea, eb):
        assert torch.equal(x, y)
    assert a.rng.offset == b.rng.offset and a.A._rng.offset == b.A._rng.offset


@pytest.mark.parametrize("graph", [True, False], ids=["graph", "eager"])
def test_scalar_ring_equals_device_readback(monkeypatch, graph):
    """Trainer.step's scalars filed by the step's last launch in the ring of pinned host memory (read behind an event,
    steps later, out of order) are the values the device copy + blocking read-back returns; a slot read after
    Trainer.SCALAR_RING further steps raises instead of returning another step's values."""
    def run(ring, steps, late=False):
        monkeypatch.setenv("DUSTY_GAN_SCALAR_RING", "1" if ring else "0")
        monkeypatch.setenv("DUSTY_GAN_GRAPH", "1" if graph else "0")
        torch.manual_seed(91)
        tr = make_trainer("dusty2", True, (32, 64), 8, 4, 16, 4)
        outs = [tr.step(i) for i in range(steps)]                     # nothing read until every step is launched
        assert (tr._snap_ring is not None) == ring and (tr._graph is not None) == graph
        if late:
            return tr, outs
        return [dict(o.items()) for o in reversed(outs)][::-1]
    a, b = run(True, 9), run(False, 9)
    for i, (x, y) in enumerate(zip(a, b)):
        assert list(x) == list(y)
        for k in x:
            # (two RUNS: the scalars are sums of atomics - last-bit differences at step 0 grow with the training steps)
            assert abs(x[k] - y[k]) <= 1e-4 * max(1.0, abs(y[k])), (i, k, x[k], y[k])
    tr, outs = run(True, tr_steps := 3, late=True)
    first = outs[0]
    for i in range(tr.SCALAR_RING):
        last = tr.step(tr_steps + i)
    assert len(dict(last.items())) == len(a[0])
    with pytest.raises(RuntimeError, match="overwritten"):
        first["loss/D/adversarial"]


@pytest.mark.parametrize("arch,pl,gan_mode", [("dusty2", 2.0, "nsgan"), ("none", 0.0, "ragan")])
def test_step_scalars_with_mid_step_draws_graph_equals_eager(monkeypatch, arch, pl, gan_mode):
    """The scalar snapshot rides on the step's LAST launch.  Configurations whose G phase draws from the trainer's generator
    (the path-length block: fresh latents and noise - a counter sync in the middle of the step) or reads D(real) again
    (relativistic losses) must still file COMPLETE scalars: replayed graph == eager launches == the device tensor read
    back without the ring, key by key, over several steps."""
    def run(graph, ring):
        monkeypatch.setenv("DUSTY_GAN_GRAPH", "1" if graph else "0")
        monkeypatch.setenv("DUSTY_GAN_SCALAR_RING", "1" if ring else "0")
        torch.manual_seed(17)
        tr = make_trainer(arch, True, (32, 64), 8, 4, 16, 4, pl=pl, gan_mode=gan_mode)
        outs = [dict(tr.step(i).items()) for i in range(6)]
        assert (tr._graph is not None) == graph
        return outs
    a, b, c = run(True, True), run(False, True), run(False, False)
    for i in range(6):
        assert list(a[i]) == list(c[i]) and len(a[i]) == (7 if pl > 0 else 5)
        for k in a[i]:
            for other in (b, c):
                # (separate RUNS: atomically summed bias gradients differ in the last bit and training amplifies it step by
                #  step; a stale or incomplete snapshot would be off by O(1))
                tol = 2e-4 if i < 2 else 1e-2
                assert abs(a[i][k] - other[i][k]) <= tol * max(1.0, abs(other[i][k])), (i, k, a[i][k], other[i][k])
            assert a[i][k] == a[i][k] and (i == 0 or a[i][k] != a[i - 1][k] or k.endswith("baseline")), (i, k)


def test_graph_replay_survives_host_sync():
    """A host-side stream synchronize between two replays of the captured step must not change what the next replay
    computes.  Round 1 / 2 finding: with hipMemsetAsync nodes in the graph (the 32-byte per-sample accumulators of
    DiffAugment's contrast mean, R1's |g|^2 sums, the logits) the replay after an explicit synchronize left garbage in
    those accumulators - 1e24-sized augmented images, garbage losses, a silently corrupted run ("garbage after a
    host-side synchronize" in the round-1 notes).  The library now zero-fills with a kernel (csrc/common.h)."""
    torch.manual_seed(300)
    tr = make_trainer("none", True, (64, 1024), 512, 64, 512, 8, amp=True)
    for i in range(8):
        s = tr.step(i)
        if i % 2 == 1:
            torch.cuda.current_stream().synchronize()
        if i == 5:
            torch.cuda.synchronize()
        vals = list(s.values())
        assert all(abs(v) < 50.0 for v in vals), (i, vals)
    assert tr._graph is not None


@pytest.mark.parametrize("arch", ["dusty2", "none"])
def test_resume_continues_like_the_uninterrupted_run(tmp_path, arch):
    """cfg.resume (reference :134-144) + the position record: 3 steps, checkpoint, a NEW trainer resumed from it, 3 more
    steps == 6 uninterrupted steps - same latents, Gumbel noise, augmentation draws and batches (Philox counters and
    loader position restored), same Adam step counts.  fp32.  Measured (scripts/resume_noise.py, relative L2 over all
    parameters of G, D and G_ema after the six steps): two uninterrupted runs differ by ~1e-8 (split-K atomics), the
    dusty archs in some runs by 1.4e-4 (a last-bit logit difference flips one pixel of the hard Gumbel threshold); the
    resumed run sits in the same two clusters; the same checkpoint WITHOUT the position record (the reference's format:
    randomness re-drawn, loader restarted) lands 3.5e-3 - 4.1e-3 away.  Bounds: resumed <= 5e-4, control >= 2e-3 - the
    control shows the bound can tell the two apart."""
    from dusty_gan_amd.trainers.dcgan_amp import Trainer
    from dusty_gan_amd.utils.config import load_config

    def cfg(resume=None):
        model = {"none": "dcgan_eqlr", "dusty2": "dusty2_dcgan_eqlr"}[arch]
        ov = [f"model={model}", "dataset=synthetic", "dataset.shape=[32,64]", "model.gen.in_ch=8", "model.gen.ch_base=4",
              "model.gen.ch_max=16", "model.dis.ch_base=4", "model.dis.ch_max=16", "solver.batch_size=4",
              "enable_amp=false", "dataset.pool=3"]
        c = load_config(ov)
        c.resume = resume
        return c
    lc = {"gpu": 0, "ngpus": 1, "batch_size": 4, "num_workers": 0}

    def params(t):
        return torch.cat([getattr(t, net).store.flat.cpu() for net in ("G", "D", "G_ema")])
    torch.manual_seed(11)
    a = Trainer(cfg(), lc)
    sa = [dict(a.step(i).items()) for i in range(6)]
    torch.manual_seed(11)
    b = Trainer(cfg(), lc)
    for i in range(3):
        b.step(i)
    path = b.save_models("mid", 3 * 4, directory=str(tmp_path))
    torch.manual_seed(999)  # the resumed process has another torch seed: everything must come from the checkpoint
    c = Trainer(cfg(resume=path), lc)
    assert c.start_iteration == 3 and c.optim_G.step_count == 3 and c.batches_drawn == 3
    assert torch.equal(c.fixed_noise.cpu(), a.fixed_noise.cpu())
    sc = [dict(c.step(i).items()) for i in range(3, 6)]
    assert rel_l2(params(c), params(a)) <= 5e-4, rel_l2(params(c), params(a))
    if arch == "none":  # (the dusty archs' scalars carry the flipped pixels of the hard threshold)
        for x, y in zip(sc, sa[3:]):
            for k in y:
                assert abs(x[k] - y[k]) <= 1e-4 * max(1.0, abs(y[k])), (k, x[k], y[k])
    # control: the same checkpoint without the position record
    sd = torch.load(path, weights_only=False)
    sd.pop("resume_state")
    path2 = str(tmp_path / "no_position.pth")
    torch.save(sd, path2)
    torch.manual_seed(999)
    d = Trainer(cfg(resume=path2), lc)
    for i in range(3, 6):
        d.step(i)
    assert rel_l2(params(d), params(a)) >= 2e-3, rel_l2(params(d), params(a))
    assert c.rng.offset == a.rng.offset and c.A._rng.offset == a.A._rng.offset


def test_two_trainers_interleaved_train_like_solo_runs(monkeypatch):
    """Two trainers of one process stepping alternately (graph replay, eager draws in between) end where each ends alone:
    the counter advances a trainer queues - applied by the last launch of ITS step, i.e. captured into ITS graph - are its
    own (`_lib.CounterQueue` per trainer), and so is its split-K workspace."""
    monkeypatch.setenv("DUSTY_GAN_GRAPH", "1")

    def make(seed):
        torch.manual_seed(seed)
        return make_trainer("none", True, (32, 64), 8, 4, 16, 4)   # (the dusty maskers seed their generator lazily from torch's
                                                                    #  GLOBAL seed: not a function of the trainer alone)
    def solo(seed, n):
        tr = make(seed)
        for i in range(n):
            tr.step(i)
            tr.generate()                       # an eager draw between replays (queued advance outside the graph)
        return tr
    ref_a, ref_b = solo(11, 4), solo(12, 4)
    a, b = make(11), make(12)
    for i in range(4):
        a.step(i); b.step(i)
        b.generate(); a.generate()
    assert a._graph is not None and b._graph is not None
    for tr, ref in ((a, ref_a), (b, ref_b)):
        for net in ("G", "D", "G_ema"):
            r = rel_l2(getattr(tr, net).store.flat.cpu(), getattr(ref, net).store.flat.cpu())
            assert r < 1e-5, (net, r)
        assert tr.optim_G.step_count == 4 and int(tr.optim_G._step_dev) == int(ref.optim_G._step_dev)


def test_pool_index_follows_the_loader_when_something_else_draws_a_batch(monkeypatch):
    """The synthetic pool's batch is picked ON THE DEVICE by a counter the replayed step advances.  Another consumer of the
    loader between two graph steps (an eager step with injected draws, a script) moves the host position only: the next
    step re-seeds the device index from `batches_drawn` instead of fetching a batch behind the loader's for ever."""
    from dusty_gan_amd import _lib as L
    monkeypatch.setenv("DUSTY_GAN_GRAPH", "1")
    torch.manual_seed(5)
    tr = make_trainer("none", True, (32, 64), 8, 4, 16, 4)
    for i in range(4):
        tr.step(i)
    assert tr._graph is not None and tr._pool_ctr is not None

    def device_index():
        with L.Counters.bind(tr.counters):
            L.Counters.flush_if(tr._pool_ctr)
        return int(tr._pool_ctr)
    assert device_index() == tr.batches_drawn == 4
    tr._next_batch()                       # somebody else takes a batch
    tr.step(4)
    assert device_index() == tr.batches_drawn == 6
    tr.step(5)
    assert device_index() == tr.batches_drawn == 7


def test_capture_survives_an_eager_garbage_collector():
    """A cyclic-garbage collection that starts while the step is being captured can finalize an EARLIER trainer - its hipGraph,
    the tensors of its private pool - and releasing those from inside a capturing thread aborts the process (seen once in this
    suite, at whichever test the allocation counters chose).  `Trainer._cap_open` collects before the capture opens and keeps
    the collector off until it closes.  Here: an old trainer with a captured graph is left in a reference cycle that survives
    until the new trainer's capturing step (the collector is off until then), and that step runs with the collector set to fire
    every few allocations.  (The abort itself is intermittent - one suite run in about eight without the guard, never in this small
    case alone, scripts/probes/gc_in_capture.py - so this test pins the guard's bookkeeping: the garbage is gone before the
    capture, the collector is back on after it, the step trains.)"""
    import gc
    thr, was_on = gc.get_threshold(), gc.isenabled()
    gc.collect()
    gc.disable()
    try:
        old = make_trainer("none", True, (32, 256), 64, 16, 64, 4, amp=True)
        for i in range(4):
            old.step(i)
        assert old._graph is not None
        cycle = [old]
        cycle.append(cycle)
        del old, cycle                   # unreachable now, but only the cyclic collector can free it
        tr = make_trainer("none", True, (32, 256), 64, 16, 64, 4, amp=True)
        out = [dict(tr.step(i).items()) for i in range(2)]     # the two eager warm-up steps
        gc.enable()
        gc.set_threshold(5, 1, 1)
        out += [dict(tr.step(i).items()) for i in range(2, 5)]  # capture + replays
        assert tr._graph is not None and gc.isenabled()
    finally:
        gc.set_threshold(*thr)
        if was_on:
            gc.enable()
    assert all(v == v for s in out for v in s.values())
