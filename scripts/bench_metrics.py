#!/usr/bin/env python3
"""Timing of the validation-metric kernels at the KITTI validation size: N clouds (sequence 08
has 4071 scans) of 64x1024 points, FPS to 512 points, all-pairs Chamfer, JSD voting.
usage: python scripts/bench_metrics.py [--clouds 4071] [--fps-clouds 256]
Roofline for the Chamfer kernel: VALU fp32.  One point pair = 3 sub + 1 mul + 2 fma + 1 min = 9 flops (8 without the
min); peak = 157.3 TFLOP/s packed fp32 (MI355X_MICROARCH.md)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--clouds", type=int, default=4071)
    ap.add_argument("--points", type=int, default=512)
    ap.add_argument("--fps-clouds", type=int, default=256)
    ap.add_argument("--validation", action="store_true", help="also time a full Trainer.validation() at 64x1024")
    args = ap.parse_args()
    from dusty_gan_amd.utils.metrics import chamfer_dir, compute_cov_mmd_1nna, compute_jsd
    from dusty_gan_amd.utils.sampling import downsample_point_clouds
    from oracle import metrics_oracle as MO
    g = torch.Generator(device="cuda").manual_seed(0)
    N, n = args.clouds, args.points
    ref = torch.randn(N, n, 3, device="cuda", generator=g) * 0.2
    gen = torch.randn(N, n, 3, device="cuda", generator=g) * 0.25
    ms = timed(lambda: chamfer_dir(ref, gen), reps=2)
    pairs = N * N * n * n
    res = {"chamfer_dir_ms": round(ms, 2), "clouds": N, "points": n,
           "point_pairs_per_s": round(pairs / ms * 1e3, 0), "TFLOPs_9_per_pair": round(9 * pairs / ms / 1e9, 2),
           "frac_of_157_TFLOPs": round(9 * pairs / ms / 1e9 / 157.3, 4)}
    t0 = time.perf_counter()
    compute_cov_mmd_1nna(gen, ref, 512, ("cd",), verbose=False)
    torch.cuda.synchronize()
    res["cov_mmd_1nna_s"] = round(time.perf_counter() - t0, 3)
    t0 = time.perf_counter()
    compute_jsd(gen / 2, ref / 2)
    torch.cuda.synchronize()
    res["jsd_s"] = round(time.perf_counter() - t0, 3)
    full = torch.randn(args.fps_clouds, 65536, 3, device="cuda", generator=g) * 0.3
    res["fps_ms_per_cloud_65536_to_512"] = round(timed(lambda: downsample_point_clouds(full, n), reps=1)
                                                 / args.fps_clouds, 4)
    res["fps_clouds"] = args.fps_clouds
    # CPU restatement on a bounded sample (the reference's own CPU path, nnsearch, is the same O(n*m) loop)
    a, b = ref[:8].cpu().numpy(), gen[:64].cpu().numpy()
    t0 = time.perf_counter()
    MO.chamfer_dir(a, b)
    dt = time.perf_counter() - t0
    res["cpu_oracle_point_pairs_per_s"] = round(8 * 64 * n * n / dt, 0)
    res["cpu_threads"] = torch.get_num_threads()
    t0 = time.perf_counter()
    MO.fps(full[0].cpu().numpy(), n)
    res["cpu_oracle_fps_ms_per_cloud"] = round((time.perf_counter() - t0) * 1e3, 1)
    if args.validation:
        # the whole reference validation pass (trainers/dcgan_amp.py:342-393) at the KITTI size: N real + N generated
        # 64x1024 scans -> point maps -> FPS to 512 -> SWD + JSD + COV/MMD/1-NNA(CD)
        from dusty_gan_amd.trainers.dcgan_amp import Trainer
        from dusty_gan_amd.utils.config import load_config
        del ref, gen, full
        torch.cuda.empty_cache()
        pool = -(-args.clouds // 32)
        cfg = load_config(["model=dusty2_dcgan_eqlr", "dataset=synthetic", "dataset.shape=[64,1024]",
                           "solver.batch_size=32", f"dataset.pool={pool}", "enable_amp=true"])
        tr = Trainer(cfg, {"gpu": 0, "ngpus": 1, "batch_size": 32, "num_workers": 0})
        tr.validation()  # warm-up (workspaces, sort plans)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        sc = tr.validation()
        torch.cuda.synchronize()
        res["validation_s"] = round(time.perf_counter() - t0, 2)
        res["validation_N"] = pool * 32
        res["validation_scores"] = {k: round(v, 5) for k, v in sc.items() if k in ("swd-mean", "jsd", "mmd-cd", "cov-cd",
                                                                                    "1-nn-accuracy-cd")}
    print(json.dumps(res))


if __name__ == "__main__":
    main()
