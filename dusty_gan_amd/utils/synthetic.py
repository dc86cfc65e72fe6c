"""Device-resident synthetic LiDAR batches (SURVEY.md §8d "Synthetic inputs"): depth ~ exp(U(ln 1, ln 110)) metres,
validity ~ Bernoulli(0.85), normalised like datasets/kitti.py:54-67 (polar depth in [0,1], invalid -> 0).
A fixed pool is cycled so data loading stays off the measured path."""
import math

import torch

from .rng import Philox


class SyntheticLiDAR:
    graph_safe = True  # fixed-shape device batches (see Trainer._graph_eligible)

    def __init__(self, batch, H, W, device, seed=1234, pool=4, min_depth=0.9, max_depth=120.0):
        rng = Philox(seed, device, stream_id=9)
        self.batches = []
        n = batch * H * W
        # one tensor per field holds the whole pool ([pool, B, 1, H, W]); the batches are views of it, so a kernel can
        # pick the batch by a device-resident index (dg_fetch_reals_pool_sum: the captured step needs no copy per replay)
        self.pool_depth = torch.empty(pool, batch, 1, H, W, dtype=torch.float32, device=device)
        self.pool_mask = torch.empty(pool, batch, 1, H, W, dtype=torch.float32, device=device)
        for i in range(pool):
            u = rng.uniform(n)
            depth_m = torch.exp(u * (math.log(110.0) - math.log(1.0)) + math.log(1.0))
            mask = rng.uniform(n) < 0.85
            pol = (depth_m - min_depth) / (max_depth - min_depth)
            pol = torch.where(mask, pol, torch.zeros_like(pol))
            # (mask kept as float: fetch_reals' `.float()` is then a no-op instead of a conversion kernel per step)
            self.pool_depth[i].copy_(pol.view(batch, 1, H, W))
            self.pool_mask[i].copy_(mask.view(batch, 1, H, W).float())
            self.batches.append({"depth": self.pool_depth[i], "mask": self.pool_mask[i]})

    def __iter__(self):
        return iter(self.batches)

    def __len__(self):
        return len(self.batches)
