"""Philox4x32-10 stream on the device (csrc/optim.hip).  The reference draws z, Gumbel uniforms and DiffAugment
parameters from torch's unseeded device generator (trainers/dcgan_amp.py:151-152; models/dusty.py:33-34;
utils/diff_augment.py:27-28,59-60,86-87); here every draw is a counter-based Philox call so a (seed, rank) pair
reproduces a run and parity tests can inject the draws instead."""
import torch

from .. import _lib as L


class Philox:
    """The counter (Philox offset) lives in DEVICE memory and is advanced by a tiny kernel after each draw, so a
    training step captured in a hipGraph replays with fresh draws (a host-side offset would be frozen in the
    captured kernel arguments)."""

    def __init__(self, seed, device, stream_id=0):
        self.seed, self.device, self.stream_id = int(seed) & (2**64 - 1), device, int(stream_id)
        self.ctr = torch.zeros(1, dtype=torch.int64, device=device)
        # the queue of counter advances this generator belongs to: its creator's (a Trainer builds its generators with its own
        # queue current).  Draws made from anywhere - a test, a script calling sample_latents between two steps - queue
        # THERE, where the owner's next step looks before it launches or replays anything.
        self.queue = L.Counters.current()

    @property
    def offset(self):
        L.Counters.flush_if(self.ctr)
        return int(self.ctr.item())

    def advance(self, n):
        """queued (L.Counters): applied with the step's other counters, or before this counter is read again"""
        with L.Counters.bind(self.queue):
            L.Counters.add(self.ctr, n)

    def sync(self):
        """the device counter is current (call before a kernel reads it)"""
        L.Counters.flush_if(self.ctr)

    def _fill(self, kind, n, lo=0.0, hi=1.0, ilo=0, ihi=1):
        out = torch.empty(n, dtype=torch.int32 if kind == 3 else torch.float32, device=self.device)
        self.sync()
        L.check(L.lib().dg_philox_fill_dev(self.seed, self.stream_id, L.ptr(self.ctr), kind, lo, hi, ilo, ihi, n,
                                           L.ptr(out), L.stream_ptr()), "dg_philox_fill_dev")
        self.advance((n + 3) // 4)
        return out

    def job(self, kind, base, **kw):
        """a DgDraw for dg_step_prologue reading this generator's device counter at +`base` (what the earlier jobs of the
        same launch take); the caller syncs before the launch and advances afterwards, as the single draws do"""
        d = L.DgDraw()
        d.kind, d.seed, d.stream_id, d.offset_dev, d.base = kind, self.seed, self.stream_id, L.ptr(self.ctr), int(base)
        for k, v in kw.items():
            setattr(d, k, v)
        return d

    def uniform(self, n, lo=0.0, hi=1.0):
        return self._fill(2 if (lo != 0.0 or hi != 1.0) else 0, n, lo, hi)

    def normal(self, n):
        return self._fill(1, n)

    def randint(self, lo, hi, n):
        """integers in [lo, hi)"""
        return self._fill(3, n, ilo=lo, ihi=hi)

    def logistic_noise(self, shape, eps=1e-10):
        """GumbelSigmoid.logistic_noise (reference: models/dusty.py:30-36): U1 and U2 drawn and combined in ONE launch
        (dg_philox_logistic_dev) - the same numbers, and the same counter advance, as two `uniform` fills followed by
        dg_logistic_noise (tests/test_gpu_ops.py)"""
        n = 1
        for s in shape:
            n *= s
        out = torch.empty(n, dtype=torch.float32, device=self.device)
        self.sync()
        L.check(L.lib().dg_philox_logistic_dev(self.seed, self.stream_id, L.ptr(self.ctr), eps, n, L.ptr(out),
                                               L.stream_ptr()), "dg_philox_logistic_dev")
        self.advance(2 * ((n + 3) // 4))
        return out.view(*shape)
