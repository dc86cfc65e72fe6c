#!/usr/bin/env python3
"""Throughput of the file input path on one MI355X box.

Writes N projected scans (64 x 2048 x 4 fp32, 2 MB each, like process_kitti.py's output) into a scratch directory, then
measures
  * the whole pipeline: host threads np.load -> pinned slot -> H2D -> dg_scan_to_polar, images/s;
  * the kernel alone on device-resident scans (HIP events), against its HBM roofline: algorithmic bytes per image =
    one 16-byte cell per OUTPUT pixel read + pol and mask written (H*W*(16+8) bytes);
  * the CPU restatement of datasets/kitti.py (oracle/lidar_oracle.py, one process) on the same files.
usage: python scripts/bench_loader.py [--n 256] [--batch 32] [--width 1024] [--workers 8]
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=256)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--workers", type=int, default=8)
    ap.add_argument("--epochs", type=int, default=4)
    args = ap.parse_args()
    from dusty_gan_amd.datasets import KITTIOdometry, ScanLoader
    from dusty_gan_amd.datasets.scans import scan_to_polar
    from oracle import lidar_oracle as LO
    Hs, Ws, H, W = 64, 2048, 64, args.width
    rng = np.random.default_rng(0)
    with tempfile.TemporaryDirectory() as root:
        d = os.path.join(root, "sequences", "00", "velodyne")
        os.makedirs(d)
        for i in range(args.n):
            np.save(os.path.join(d, f"{i:06d}.npy"), rng.normal(0, 20, (Hs, Ws, 4)).astype(np.float32))
        ds = KITTIOdometry(root, "train", shape=(H, W))
        loader = ScanLoader(ds, args.batch, "cuda", num_workers=args.workers, prefetch=3)
        for _ in loader:  # warm-up epoch (page cache, pinned buffers)
            pass
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        nimg = 0
        for _ in range(args.epochs):
            for out in loader:
                nimg += out["depth"].shape[0]
        torch.cuda.synchronize()
        pipe = nimg / (time.perf_counter() - t0)
        # kernel alone
        scans = torch.from_numpy(np.stack([np.load(p) for p in ds.datalist[:args.batch]])).cuda()
        for _ in range(3):
            scan_to_polar(scans, (H, W), 0.9, 120.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 50
        e0.record()
        for _ in range(reps):
            scan_to_polar(scans, (H, W), 0.9, 120.0)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        alg = args.batch * H * W * (16 + 8)
        # CPU restatement, one process
        t0 = time.perf_counter()
        ncpu = 0
        while time.perf_counter() - t0 < 5.0:
            LO.scan_to_polar(np.load(ds.datalist[ncpu % len(ds)]), (H, W))
            ncpu += 1
        cpu = ncpu / (time.perf_counter() - t0)
    print(json.dumps({"pipeline_images_per_s": round(pipe, 1), "raw_MB_per_s": round(pipe * Hs * Ws * 16 / 1e6, 1),
                      "kernel_us_per_batch": round(us, 2), "kernel_GBps_algorithmic": round(alg / us / 1e3, 1),
                      "kernel_frac_of_8TBps": round(alg / us / 1e3 / 8000, 4),
                      "cpu_oracle_images_per_s_1proc": round(cpu, 1), "batch": args.batch, "shape": [H, W],
                      "workers": args.workers, "files": args.n}))


if __name__ == "__main__":
    main()
