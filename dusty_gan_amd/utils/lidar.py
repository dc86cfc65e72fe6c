"""The one LiDAR coordinate helper on the hot path -- reference: utils/lidar.py:31-36 (Coordinate.invert_depth),
used by Trainer.fetch_reals (trainers/dcgan_amp.py:154-160).  The spherical projection / point-cloud parts of the
reference's LiDAR class are post-processing and out of scope."""
import torch

from .. import _lib as L


class LiDAR:
    def __init__(self, num_ring, num_points, min_depth, max_depth, angle_file=None):
        self.H, self.W = num_ring, num_points
        self.min_depth, self.max_depth = float(min_depth), float(max_depth)
        self.angle_file = angle_file  # only needed for xyz post-processing, which this engine does not do

    def fetch_reals(self, pol, mask, drop_const):
        """pol [B,1,H,W] in [0,1], mask [B,1,H,W] {0,1} float -> inverse depth in [-1,1], dropped pixels = drop_const"""
        pol = pol.contiguous().float()
        mask = mask.contiguous().float()
        out = torch.empty_like(pol)
        L.check(L.lib().dg_fetch_reals(L.ptr(pol), L.ptr(mask), self.min_depth, self.max_depth, float(drop_const),
                                       pol.numel(), L.ptr(out), L.stream_ptr()), "dg_fetch_reals")
        return out, mask
