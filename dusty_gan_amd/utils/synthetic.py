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
        for _ in range(pool):
            u = rng.uniform(n)
            depth_m = torch.exp(u * (math.log(110.0) - math.log(1.0)) + math.log(1.0))
            mask = rng.uniform(n) < 0.85
            pol = (depth_m - min_depth) / (max_depth - min_depth)
            pol = torch.where(mask, pol, torch.zeros_like(pol))
            # (mask kept as float: fetch_reals' `.float()` is then a no-op instead of a conversion kernel per step)
            self.batches.append({"depth": pol.view(batch, 1, H, W), "mask": mask.view(batch, 1, H, W).float()})

    def __iter__(self):
        return iter(self.batches)

    def __len__(self):
        return len(self.batches)
