"""BASELINE.json's configurations as whole `Trainer.step`s under a checker (`configs_untested`).

  configs[0]  dcgan_eqlr baseline, 32x256 (the reference cannot build 16x256), batch 8, fp32
  configs[1]  dcgan_eqlr baseline, 64x1024, batch 32, bf16  - also tests/test_gpu_timed_path.py; here against the
              bf16-EMULATING oracle with a tight gradient bound
  configs[2]  dusty1_dcgan_eqlr, 64x1024, batch 32, bf16 and fp32, with the two-head thin matrix-core kernels asserted
  configs[3]  per-GPU share (dusty2, 64x1024, batch 32, bf16) against the emulating oracle
  full width  the HIP fp32 path against digests the REFERENCE's own modules produced at 64x1024 / 512 channels
              (tests/golden/full_dusty2.npz, made by tests/golden/make_golden.py full)
configs[4]'s shapes (128x2048) are in tests/test_gpu_timed_path.py::test_config5_shapes_whole_step.

Two oracle modes (oracle/dusty_oracle.py): fp32 (pinned to the reference by tests/golden) and `emulate_bf16`, which rounds
to bfloat16 exactly where the engine's bf16 mode stores bfloat16.  In both modes the engine takes over the oracle's updated
discriminator between the phases (`sync=True`), so the G-phase gradients are compared on identical discriminators.

Measured at 64x1024, batch 32 (scripts/emu_gap.py, arch none), engine bf16 against the three checkers, worst tensor:
                         fp32 oracle, unsynced   fp32 oracle, synced   emulating oracle, synced   bound here
  depth (rel-L2)                5.3e-3                 5.3e-3                  1.1e-3               5e-3
  losses (rel)                  1.3e-3                 1.3e-3                  8e-5                 2e-3
  D gradients (rel-L2)          2.4e-2                 2.4e-2                  1.3e-2 (weights 5e-3) 2.5e-2
  G gradients (rel-L2)          1.3e-1                 1.1e-1                  5.3e-2               8e-2
Why the emulating oracle does not agree to 1e-3 either: two fp32 accumulation orders differ by ~5e-6 relative, which
moves 0.3-1 % of the stored values across a bf16 rounding boundary (one ulp = 4e-3 relative); every following layer
re-rounds, so after three layers the two activations sit one bf16 rounding noise (~1e-3) apart whatever the start, and a
1e-3 shift of the pre-activations flips the leaky-relu slope of ~4e-4 of the units per layer - each an 80 % change of that
unit's gradient, ~1.6e-2 relative L2 per layer, ~5e-2 over the 8-9 layers a generator gradient crosses.  The bounds below
are 1.5x the measured worst case: 3x (G) and 10x (D) tighter than the 0.25 of tests/test_gpu_timed_path.py, and below
what a layer mis-scaled by 10 % would show.
"""
import pytest
import torch

from oracle import dusty_oracle as O
from tests.golden_util import check_digest, digest, full_case, load, rel_l2, sub
from tests.test_gpu_step import _cos, grads_by_name, make_trainer, run_both, sync_D
from tests.test_gpu_timed_path import _persist

pytestmark = pytest.mark.gpu
KEYS = ["loss/D/output/real", "loss/D/output/fake", "loss/D/adversarial", "loss/D/gradient_penalty",
        "loss/G/adversarial"]


@pytest.fixture
def trace():
    from dusty_gan_amd import engine as E
    E.TRACE = []
    yield E.TRACE
    E.TRACE = None


# Gradient bounds = what these tests measure on MI355X (round 5, `pytest -s` prints every run's worst values as "PARITY ...")
# times 1.25 - 3.5, not a round number that a 1 % defect would pass under (the round-4 review's point):
#   fp32 parity mode, whole step at full width against the fp32 oracle: D 2.7e-4 ... 1.0e-3, G 1.4e-3 ... 1.5e-3 rel-L2 per tensor,
#     cosine >= 0.999998 (config 1 at B = 8, config 3 at B = 32; the B = 2 full-width case of tests/test_gpu_step.py: 2e-6 / 2e-5).
#     The spread is leaky-relu slope flips of units within rounding of zero (tests/test_gpu_step.py docstring): 5e-3 / 0.99999.
#   bf16 timed mode against the bf16-EMULATING oracle: D 1.0e-2 ... 1.7e-2, G 4.8e-2 ... 5.3e-2, cosine 0.9986 ... 0.9999:
#     D 2.1e-2, G 6.6e-2, cosine 0.9978 (1.25 x the worst seen).
FP32 = dict(out_tol=1e-3, loss_tol=1e-3, grad_tol={"grad_D": 5e-3, "grad_G": 5e-3}, cos_min=0.99999, mask_tol=1e-4)
BF16_EMU = dict(out_tol=5e-3, loss_tol=2e-3, grad_tol={"grad_D": 2.1e-2, "grad_G": 6.6e-2}, cos_min=0.9978, mask_tol=2e-3)


def _check(res, tr, state, out_tol, loss_tol, grad_tol, cos_min, mask_tol, lr=0.002):
    """outputs / losses / every gradient tensor / post-Adam parameters of one synced step"""
    G, D, G_ema = state
    sc_ref, ex, synth, gD, gG, scal = res
    for k, v in zip(KEYS, scal):
        assert abs(v - sc_ref[k]) <= loss_tol * max(1.0, abs(sc_ref[k])), (k, v, sc_ref[k])
    for k in synth:
        if k == "mask":
            assert (synth[k] != ex["synth"][k]).float().mean() <= mask_tol
        else:
            assert rel_l2(synth[k], ex["synth"][k]) <= out_tol, (k, rel_l2(synth[k], ex["synth"][k]))
    worst = {}
    fails = []
    for name, got, ref in (("grad_D", gD, ex["grad_D"]), ("grad_G", gG, ex["grad_G"])):
        for k, v in ref.items():
            if v.abs().max() > 0:
                r, c = rel_l2(got[k], v), _cos(got[k], v)
                worst[name] = max(worst.get(name, 0.0), r)
                worst[name + "_cos"] = min(worst.get(name + "_cos", 1.0), c)
                if r > grad_tol[name] or c < cos_min:
                    fails.append((name, k, f"rel-L2 {r:.3e} (bound {grad_tol[name]:.1e})", f"cosine {c:.6f} (bound {cos_min})"))
    # (measured values in the record: `pytest -s` shows how far inside its bounds a run sits)
    print("PARITY", {k: float(f"{v:.7g}") for k, v in worst.items()}, "bounds", grad_tol, cos_min)
    assert not fails, (fails, worst)
    # post-Adam parameters: the first step at beta1 = 0 moves every element by ~lr * sign(g), so every element is within
    # 2 lr of the oracle's; D before the sync (the engine's own update), G and G_ema after
    for tag, got_sd, ref in (("G", tr.G.state_dict(), G), ("D", ex["D_engine_after"], D), ("G_ema", tr.G_ema.state_dict(), G_ema)):
        for k, v in ref.items():
            if k == "drop_const" or k.endswith("kernel"):
                continue
            got = got_sd[k].cpu()
            assert float((got - v).abs().max()) <= 2.001 * lr, (tag, k)
            if float(v.abs().mean()) > 0.1:
                assert rel_l2(got, v) < 1e-3, (tag, k)
    return worst


def test_config1_plumbing_32x256_batch8_fp32(trace):
    """BASELINE configs[0] as SURVEY §8d defines it: dcgan_eqlr baseline, 32x256, batch 8, fp32, one whole step against
    the fp32 oracle at the north-star tolerance (<= 1e-3; gradients where a unit within rounding of zero takes the other
    slope: 2e-2 / cosine 0.9999, see tests/test_gpu_step.py)."""
    tr, state, res = run_both("none", (32, 256), 512, 64, 512, 8, amp=False, sync=True)
    _check(res[0], tr, state, **FP32)


@pytest.mark.parametrize("amp", [True, False], ids=["bf16", "fp32"])
def test_config3_dusty1_64x1024_batch32(trace, amp):
    """BASELINE configs[2]: dusty1_dcgan_eqlr (Gumbel point-drop), 64x1024, 512 / 64..512, batch 32 - bf16 (the timed
    mode, against the emulating oracle) and fp32 (the parity mode, against the fp32 oracle).  In bf16 the two-head Head
    runs on the thin matrix-core kernels: forward `thin_up_mfma` (N = 2), backward-data `thin_s2_mfma` (2-channel
    pixel-major gradient), weight gradient `thin_wgrad_up_mfma` - asserted through the launch introspection."""
    tr, state, res = run_both("dusty1", (64, 1024), 512, 64, 512, 32, amp=amp, emulate=amp, sync=True)
    assert len(_persist(trace)) >= 10
    if amp:
        thin = [t for t in trace if t[0] == "conv" and t[1] == 3]
        assert any(t[8] == 2 and " N2" in t[7] for t in thin), thin            # Head forward, two heads, thin_up_mfma
        assert any(t[8] == 1 and " K2 " in t[7] for t in thin), thin           # Head backward-data, thin_s2_mfma<2>
        assert any(t[8] == 1 and "mode0adj0" in t[7] for t in thin), thin      # Down1 forward
        assert any(t[0] == "wgrad" and t[1] == 7 and "Co2" in t[2] for t in trace), trace  # Head wgrad on the matrix cores
        assert any(t[0] == "wgrad" and t[1] == 7 and "Ci2" in t[2] for t in trace), trace  # Down1 wgrad
        _check(res[0], tr, state, **BF16_EMU)
    else:
        _check(res[0], tr, state, **FP32)


@pytest.mark.parametrize("arch", ["none", "dusty2"])
def test_timed_mode_gradients_tight_against_emulating_oracle(trace, arch):
    """BASELINE configs[1] (arch none) and configs[3]'s per-GPU share (dusty2) exactly as bench.py runs them - 64x1024,
    batch 32, bf16 - with every gradient tensor held to the bounds of the module docstring against the bf16-emulating
    oracle (tests/test_gpu_timed_path.py holds the same step to the fp32 oracle at 0.25: that bound cannot see a
    mis-scaled layer, these can)."""
    tr, state, res = run_both(arch, (64, 1024), 512, 64, 512, 32, amp=True, emulate=True, sync=True)
    assert any(t[0] == "wgrad" and t[1] == 5 for t in trace)
    _check(res[0], tr, state, **BF16_EMU)


def test_timed_mode_within_the_reference_autocast_yardstick():
    """The bf16 mode bench.py times, held to numbers the REFERENCE produced (round-5 review, item 4): the full-width step of
    tests/golden/full_dusty2.npz run by the engine in bf16, every output and gradient compared with the reference's fp32
    digests - its distance may not exceed 1.5 x the distance of the reference's OWN bfloat16-autocast run of the same step
    (tests/golden/full_dusty2_autocast.npz: `make_golden.py autocast`, the reference's modules under torch.autocast("cpu",
    bfloat16) in the trainer's enable_amp regions, trainers/dcgan_amp.py:194-211,228-232,253-264), tensor by tensor.
    One-number tensors (head bias gradients: sums that cancel to 1e-3 of their absolute sum) are held to 1.0 instead."""
    from tests.golden_util import sample_dev
    g, ga = load("full_dusty2"), load("full_dusty2_autocast")
    if str(g["meta/torch"]) != torch.__version__:
        pytest.skip(f"fixture made with torch {g['meta/torch']}: the regenerated inputs need the same CPU generator")
    ref_dev = {k[:-7]: rel_l2(ga[k], g[k]) for k in g.files if k.endswith("/sample") and k in ga.files}
    G, D, pol, mask, rand = full_case(g)
    shape = tuple(int(v) for v in g["meta/shape"])
    tr = make_trainer(str(g["meta/arch"]), True, shape, int(g["meta/in_ch"]), int(g["meta/ch_base"]),
                      int(g["meta/ch_max"]), int(g["meta/B"]), amp=True)
    assert tr.dtype == torch.bfloat16
    tr.G.load_state_dict(G)
    tr.G_ema.load_state_dict(G)
    sync_D(tr, D)
    x_real, m_real = tr.fetch_reals({"depth": pol, "mask": mask})
    tr.optimize_D(reals=[(x_real, m_real)], rands=[rand])
    synth = {k: v.detach().cpu().clone() for k, v in tr._mb[0]["synth"].items()}
    gD = grads_by_name(tr.optim_D)
    scal = tr.optimize_G().cpu().tolist()
    gG = grads_by_name(tr.optim_G)
    report = {}
    for key, t in ([(f"synth/{k}", synth[k]) for k in ("depth_orig", "confidence")]
                   + [(f"grad_D/{k}", v) for k, v in gD.items()] + [(f"grad_G/{k}", v) for k, v in gG.items()]):
        dev = sample_dev(g, key, t.cpu())
        if t.numel() <= 4:
            assert dev <= 1.0, (key, dev)
            continue
        report[key] = (dev, ref_dev[key])
        assert dev <= 1.5 * ref_dev[key], (key, dev, ref_dev[key])
    _, msample = digest(synth["mask"])
    assert (msample != g["synth/mask/sample"]).mean() <= 5e-3          # hard Gumbel thresholds: a logit within bf16 of zero may flip
    for k, v in zip(KEYS, scal):
        d_ref = abs(float(ga[f"scalar/{k}"]) - float(g[f"scalar/{k}"]))
        assert abs(v - float(g[f"scalar/{k}"])) <= 1.5 * d_ref + 2e-3 * max(1.0, abs(float(g[f"scalar/{k}"]))), (k, v, d_ref)
    worst = max(report.items(), key=lambda kv: kv[1][0] / kv[1][1])
    print("bf16 engine vs reference autocast (distance from the reference's fp32 digests), worst ratio:", worst[0],
          "%.3e / %.3e" % worst[1])


@pytest.mark.parametrize("x3,pairs", [(False, False), (True, True), (True, False)], ids=["fp32", "fp32x3", "fp32x3-register-split"])
def test_hip_path_matches_reference_at_full_width(monkeypatch, x3, pairs):
    """The HIP fp32 path against what the REFERENCE's own modules computed at 64x1024 / 512 channels (dusty2, B = 2, one
    step): digests from tests/golden/full_dusty2.npz, inputs regenerated from its seed.  Outputs, logits, losses 1e-4;
    gradients 5e-3 and cosine >= 0.99999 on the digests (a unit within fp32 rounding of zero takes the other slope in the
    two implementations - tests/test_gpu_step.py::test_step_fp32_vs_oracle_full_width_64x1024: measured worst 1.5e-3 /
    0.999998); updated parameters 1e-3.
    fp32x3: the same bounds with the fat layers' contractions on the bf16 matrix instructions (operands split into bf16
    hi + lo, DG_FORCE_FP32X3 on every launch of this trainer's engines) - the fast parity mode of `bench.py --precision fp32x3`:
    with the fat feature maps STORED as split-bf16 pairs (DG_BF16X2, round 5: the bf16 ping-pong conv and LDS-DMA weight
    gradient contract the halves in three K steps; what the mode runs) and with fp32 storage and the split made in registers
    by the one-tile kernels (round 4's form, `Trainer.fp32_pairs_default = False`)."""
    monkeypatch.setenv("DUSTY_GAN_FP32_SPLIT", "1" if x3 else "0")
    from dusty_gan_amd.trainers.dcgan_amp import Trainer
    monkeypatch.setattr(Trainer, "fp32_pairs_default", pairs)
    g = load("full_dusty2")
    if str(g["meta/torch"]) != torch.__version__:
        pytest.skip(f"fixture made with torch {g['meta/torch']}: the regenerated inputs need the same CPU generator")
    G, D, pol, mask, rand = full_case(g)
    shape = tuple(int(v) for v in g["meta/shape"])
    tr = make_trainer(str(g["meta/arch"]), True, shape, int(g["meta/in_ch"]), int(g["meta/ch_base"]),
                      int(g["meta/ch_max"]), int(g["meta/B"]), amp=False)
    assert abs(tr.ema_decay - float(g["meta/ema_decay"])) < 1e-12
    assert tr.fp32_split == x3
    assert tr.D.engine().ops.x3 == x3 and tr._g_engines()[0].ops.x3 == x3   # per engine: nothing process-wide to reset
    assert tr.fp32_pairs == pairs and tr.D.engine().x2 == pairs and tr._g_engines()[0].x2 == pairs
    tr.G.load_state_dict(G)
    tr.G_ema.load_state_dict(G)
    sync_D(tr, D)
    x_real, m_real = tr.fetch_reals({"depth": pol, "mask": mask})
    check_digest(g, "x_real", x_real.cpu(), 1e-5)
    tr.optimize_D(reals=[(x_real, m_real)], rands=[rand])
    synth = {k: v.detach().cpu().clone() for k, v in tr._mb[0]["synth"].items()}
    gD = grads_by_name(tr.optim_D)
    scal = tr.optimize_G().cpu().tolist()
    gG = grads_by_name(tr.optim_G)
    for k, v in zip(KEYS, scal):
        ref = float(g[f"scalar/{k}"])
        assert abs(v - ref) <= 1e-4 * max(1.0, abs(ref)), (k, v, ref)
    for k, v in synth.items():
        if k == "mask":
            stats, sample = digest(v)
            assert abs(stats[0] - g["synth/mask/stats"][0]) <= 4
            assert (sample != g["synth/mask/sample"]).mean() <= 1e-3
        else:
            check_digest(g, f"synth/{k}", v, 1e-4)
    # gradients against the REFERENCE's digests (round 6; the 2e-2 of rounds 3-5 would have passed a 1 % defect in one layer):
    #   fp32            5e-3 and cosine >= 0.99999 (measured worst 1.5e-3 / 0.999998)
    #   fp32x3          weights 1.3e-2 / 0.9999 (measured worst 1.01e-2, 16 mantissa bits per operand); bias gradients - sums over 10^4..10^5 signed terms, where the split products'
    #                   2^-16 relative error shows first - 1.5e-2 / 0.9999 (measured 1.07e-2 on Proj's bias); a head bias
    #                   gradient is ONE number, the sum of 131 072 per-pixel gradients that cancel to ~1e-3 of their absolute
    #                   sum: 3.7e-2 measured, held to 1.25 x that
    from tests.golden_util import sample_dev
    measured, bad = {}, []
    for pre, grads in (("grad_D", gD), ("grad_G", gG)):
        for k, v in grads.items():
            key = f"{pre}/{k}"
            measured[key] = sample_dev(g, key, v)
            one, bias = v.numel() <= 4, v.dim() == 1
            if x3 and one:
                tol, cos = 4.6e-2, None
            elif x3 and bias:
                tol, cos = 1.5e-2, 0.9999
            elif x3:
                tol, cos = 1.3e-2, 0.9999    # (measured worst 1.01e-2: Up2's weight gradient)
            else:
                tol, cos = 5e-3, (None if one else 0.99999)
            try:
                check_digest(g, key, v, tol, pre, cos=cos)
            except AssertionError as e:   # (all of them in one report)
                bad.append((key, measured[key], tol, str(e)[:120]))
    assert not bad, bad
    print("gradient distances from the reference digests, worst five:",
          sorted(measured.items(), key=lambda kv: -kv[1])[:5])
    for tag, net in (("G", tr.G), ("D", tr.D), ("G_ema", tr.G_ema)):
        for k, v in net.state_dict().items():
            if k == "drop_const" or k.endswith("kernel"):
                continue
            ref = g[f"after/{tag}/{k}/sample"]
            if float(abs(ref).mean()) > 0.1:      # N(0,1)-initialised weights: the whole digest
                check_digest(g, f"after/{tag}/{k}", v.cpu(), 1e-3, tag)
            else:                                 # zero-initialised biases after one lr * sign(g) step: element bound
                _, sample = digest(v.cpu())
                assert float(abs(sample - ref).max()) <= 2.001 * float(g["meta/lr"]), (tag, k)
