"""DiffAugment on [B,1,H,W] range images -- reference: utils/diff_augment.py:114-132 (p = 1.0).

One fused gather kernel per call (csrc/pointwise.hip) instead of the reference's ~40 small ops, plus the hand-derived
backward (the G phase differentiates through A(fake), trainers/dcgan_amp.py:256).  Every random draw of one call is
an explicit parameter set `rp` so parity tests can inject the reference's draws:
    u_b,u_s,u_c float [B] (the uniform_(-1,1) draws; the applied factor is u*u),
    t_h,t_w,o_x,o_y int [B].
"""
import torch
from torch import nn

from .. import _lib as L
from .rng import Philox

DEFAULT_POLICY = ["brightness", "saturation", "contrast", "translation", "cutout"]


class DiffAugment(nn.Module):
    def __init__(self, policy=None, p=1.0, seed=None):
        super().__init__()
        self.policy = list(DEFAULT_POLICY if policy is None else policy)
        for k in self.policy:
            if k not in L.POLICY_BITS:
                raise KeyError(k)  # same failure mode as AUGMENT_FNS[p] in the reference
        # the fused kernel applies the stages in the reference's shipped order (solver/nsgan_eqlr.yaml:34-39); the
        # reference iterates the list as given (diff_augment.py:124-131) and contrast's per-sample mean does not commute
        # with translation / cutout, so another order would silently compute something else
        order = [k for k in DEFAULT_POLICY if k in self.policy]
        if [k for k in self.policy if k in DEFAULT_POLICY] != order or len(set(self.policy)) != len(self.policy):
            raise NotImplementedError(f"DiffAugment policy order {self.policy}: the kernel applies {order}")
        if p != 1.0:
            raise NotImplementedError("DiffAugment probability p != 1 (the reference trainer always uses p=1)")
        self.p = p
        self.mask = L.policy_mask(self.policy)
        self._seed = seed
        self._rng = None

    # ---- parameter draws
    def rng(self, device):
        if self._rng is None or self._rng.device != device:
            seed = torch.initial_seed() + 101 if self._seed is None else self._seed
            self._rng = Philox(seed, device, stream_id=5)
        return self._rng

    def draw(self, B, H, W, device):
        return self.draw_sets(1, B, H, W, device)[0]

    def draw_sets(self, n, B, H, W, device):
        """`n` independent parameter sets in ONE launch.  Set k, sample b uses Philox counter offset 2 (k B + b), i.e.
        exactly the numbers `n` consecutive draw() calls produce."""
        r = self.rng(device)
        uf = torch.empty(3, n * B, dtype=torch.float32, device=device)
        qi = torch.empty(4, n * B, dtype=torch.int32, device=device)
        r.sync()
        L.check(L.lib().dg_aug_draw_dev(r.seed, r.stream_id, L.ptr(r.ctr), n * B, H, W, L.ptr(uf), L.ptr(qi),
                                        L.stream_ptr()), "dg_aug_draw_dev")
        r.advance(2 * n * B)
        return self.sets_of(uf, qi, n, B)

    @staticmethod
    def sets_of(uf, qi, n, B):
        """the `n` parameter sets held by one draw's output buffers uf [3][n B], qi [4][n B]"""
        sets = []
        for k in range(n):
            sl = slice(k * B, (k + 1) * B)
            sets.append({"u_b": uf[0, sl], "u_s": uf[1, sl], "u_c": uf[2, sl], "t_h": qi[0, sl], "t_w": qi[1, sl],
                         "o_x": qi[2, sl], "o_y": qi[3, sl]})
        return sets

    @staticmethod
    def params_to_device(rp, device):
        """Accept CPU / int64 parameter sets (golden vectors) and put them in the kernel's dtypes."""
        out = {}
        for k, v in rp.items():
            v = torch.as_tensor(v)
            out[k] = v.to(device=device, dtype=torch.float32 if k.startswith("u_") else torch.int32).contiguous()
        return out

    def _args(self, rp, B, device):
        # stand-ins for parameters a policy does not draw: zeros, allocated once per (B, device) - two fill kernels per
        # call were ~10 of the step's ~90 tiny launches
        key = (B, str(device))
        if getattr(self, "_zeros_key", None) != key:
            self._zeros = (torch.zeros(B, dtype=torch.float32, device=device),
                           torch.zeros(B, dtype=torch.int32, device=device))
            self._zeros_key = key
        z_f, z_i = self._zeros
        g = lambda k, z: L.ptr(rp[k]) if k in rp else L.ptr(z)
        keep = (z_f, z_i)
        return (g("u_b", z_f), g("u_c", z_f), g("t_h", z_i), g("t_w", z_i), g("o_x", z_i), g("o_y", z_i)), keep

    def aug_set(self, x, rp):
        """(DgAugSet, keep-alive) for the fused DiffAugment + BlurVH pass of the discriminator's input, or None when the
        per-sample sums of x were not made by its producer in this step (the caller then runs `apply` + BlurVH)"""
        tag = L.tagged_sums(x, with_parts=True)
        if tag is None or not x.is_contiguous() or x.dtype != torch.float32 or x.shape[1] != 1:
            return None
        pre, parts = tag
        args, keep = self._args(rp, x.shape[0], x.device)
        q = L.DgAugSet()
        q.x, q.xsum, q.xsum_parts = L.ptr(x), L.ptr(pre), parts
        q.u_b, q.u_c, q.t_h, q.t_w, q.o_x, q.o_y = args
        return q, (keep, pre, x)

    # ---- forward / backward
    def apply(self, x, rp, out=None):
        """y = A(x) with the given draws.  x [B,1,H,W] fp32 (cuda)."""
        if not x.is_cuda:
            raise RuntimeError("DiffAugment runs on the GPU only")
        B, Cc, H, W = x.shape
        if Cc != 1:
            raise NotImplementedError("DiffAugment kernel handles single-channel range images")
        x = x.contiguous()
        if out is None:
            out = torch.empty_like(x)
        pre = L.tagged_sums(x)                # the producer of x already summed it per sample (fetch_reals / head kernels)
        if pre is not None:
            args, keep = self._args(rp, B, x.device)
            L.check(L.lib().dg_diffaug_fwd_pre(L.ptr(x), *args, self.mask, B, H, W, L.ptr(pre), L.ptr(out), L.stream_ptr()),
                    "dg_diffaug_fwd_pre")
            return out
        ws = L.AccArena.take(B, x.device)     # per-sample sums: a pre-zeroed slice of the step's arena, if a step is running
        fn = L.lib().dg_diffaug_fwd_acc if ws is not None else L.lib().dg_diffaug_fwd
        if ws is None:
            ws = torch.empty(B, dtype=torch.float32, device=x.device)
        args, keep = self._args(rp, B, x.device)
        L.check(fn(L.ptr(x), *args, self.mask, B, H, W, L.ptr(ws), L.ptr(out), L.stream_ptr()), "dg_diffaug_fwd")
        return out

    def backward(self, gy, rp, out=None):
        """gx = dL/dx from gy = dL/dA(x) for the same draws."""
        B, _, H, W = gy.shape
        gy = gy.contiguous()
        if out is None:
            out = torch.empty_like(gy)
        ws = L.AccArena.take(B, gy.device)
        fn = L.lib().dg_diffaug_bwd_acc if ws is not None else L.lib().dg_diffaug_bwd
        if ws is None:
            ws = torch.empty(B, dtype=torch.float32, device=gy.device)
        args, keep = self._args(rp, B, gy.device)
        L.check(fn(L.ptr(gy), *args, self.mask, B, H, W, L.ptr(ws), L.ptr(out), L.stream_ptr()), "dg_diffaug_bwd")
        return out

    def backward_pre(self, gy, rp, gsum, out=None):
        """`backward` with gsum[b] = the sum of gy[b] over the window that reaches the source (made by the BlurVH
        adjoint that produced gy: dg_blur_bwd_augsum)"""
        B, _, H, W = gy.shape
        if out is None:
            out = torch.empty_like(gy)
        args, keep = self._args(rp, B, gy.device)
        L.check(L.lib().dg_diffaug_bwd_pre(L.ptr(gy), *args, self.mask, B, H, W, L.ptr(gsum), L.ptr(out), L.stream_ptr()),
                "dg_diffaug_bwd_pre")
        return out

    def forward(self, x):
        B, _, H, W = x.shape
        return self.apply(x, self.draw(B, H, W, x.device))
