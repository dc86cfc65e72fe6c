#!/usr/bin/env python3
"""Proj.weight optimizer step at the benchmark size (Np = 131072, K = 512) for per-GPU batch 32 and all-gathered
batches 64..256: gradient GEMM + plain Adam/EMA (unfused) against dg_adam_proj_fused (VALU kernel for nb <= 62, MFMA
epilogue above).  usage: python scripts/bench_proj_adam.py"""
import json
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dusty_gan_amd import _lib as L  # noqa: E402
import ctypes as C  # noqa: E402


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps


def main():
    lib = L.lib()
    Np, K = 131072, 512
    n = Np * K
    p, v, ema, grad = (torch.randn(n, device="cuda") for _ in range(4))
    v.abs_()
    sh = torch.empty(n, device="cuda", dtype=torch.bfloat16)
    step = torch.zeros(1, dtype=torch.int64, device="cuda")
    res = {}
    for nb in (32, 64, 128, 256):
        dp0 = torch.randn(nb, Np, device="cuda").bfloat16()
        z = torch.randn(nb, K, device="cuda").bfloat16()
        w = L.DgWgrad()
        w.wmode, w.ring, w.B, w.Hc, w.Wc, w.Ci, w.Co = 2, 1, 1, 1, nb, Np, K
        w.a, w.a_sb, w.a_sp, w.a_sc = dp0.data_ptr(), 0, Np, 1
        w.g, w.g_sb, w.g_sp, w.g_sc = z.data_ptr(), 0, K, 1
        w.dw, w.scale, w.rowscale, w.a_dtype, w.g_dtype = grad.data_ptr(), 1.0 / math.sqrt(Np), None, L.DG_BF16, L.DG_BF16

        def unfused():
            L.check(lib.dg_wgrad(C.byref(w), 0, 0, None))
            L.check(lib.dg_adam_ema_step_dev(p.data_ptr(), grad.data_ptr(), None, v.data_ptr(), ema.data_ptr(),
                                             sh.data_ptr(), L.DG_BF16, n, 1.0, 0.002, 0.0, 0.99, 1e-8, step.data_ptr(),
                                             0.999, None))

        def fused():
            L.check(lib.dg_adam_proj_fused(p.data_ptr(), v.data_ptr(), ema.data_ptr(), sh.data_ptr(), L.DG_BF16,
                                           dp0.data_ptr(), z.data_ptr(), L.DG_BF16, nb, Np, K, 1.0 / math.sqrt(Np), 1.0,
                                           0.002, 0.99, 1e-8, step.data_ptr(), 0.999, None))
        res[nb] = {"unfused_us": round(timed(unfused), 1), "fused_us": round(timed(fused), 1)}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
