"""LiDAR coordinate helper -- reference: utils/lidar.py:11-68,111-130 (Coordinate / LiDAR).

On the training path it provides `fetch_reals` (Coordinate.invert_depth :31-36 fused with sigmoid_to_tanh and the
drop-constant fill of Trainer.fetch_reals, trainers/dcgan_amp.py:154-160).  On the output side it turns generated
inverse depth into the unit-space point map (`inv_to_xyz` :58-65 with revert_depth / pol_to_xyz) on the sensor's
angle grid (`angles.pt`, resized like LiDAR.init_coordmap :127-130).  All arithmetic is in csrc/ (pointwise.hip,
lidar_io.hip, incl. the surface-normal image of utils/geometry.py); `points_to_depth` (:67-108, used by the reconstruction demo only) is not built.
"""
import math
import os

import torch
import torch.nn.functional as F

from .. import _lib as L


class LiDAR:
    def __init__(self, num_ring, num_points, min_depth, max_depth, angle_file=None, drop_const=0.0):
        self.H, self.W = num_ring, num_points
        self.min_depth, self.max_depth = float(min_depth), float(max_depth)
        self.drop_const = float(drop_const)  # Coordinate's own default (utils/lidar.py:12), NOT the generator's -1
        self.angle_file = angle_file
        self.angle = None  # [1,2,H,W] (elevation, azimuth); the synthetic dataset has no angle file
        if angle_file is not None and os.path.exists(angle_file):
            self.angle = self.init_coordmap(self.H, self.W)

    def use_nominal_angles(self, fov_up_deg=2.0, fov_down_deg=-24.8):
        """An analytic angle grid (evenly spaced rings over the HDL-64E field of view, azimuth pi .. -pi like
        process_kitti.py:101-103) for the synthetic dataset, which has no angles.pt."""
        pitch = torch.linspace(math.radians(fov_up_deg), math.radians(fov_down_deg), self.H)[:, None].expand(self.H, self.W)
        yaw = -(torch.arange(self.W).float() + 0.5) / self.W * 2 * math.pi + math.pi
        self.angle = torch.stack([pitch, yaw[None, :].expand(self.H, self.W)])[None].contiguous()
        return self

    def init_coordmap(self, H, W):
        """utils/lidar.py:127-130 (one-time setup: a bilinear resize of the 2 x 64 x 2048 average-angle table)"""
        angle = torch.load(self.angle_file, map_location="cpu")[None].float()
        return F.interpolate(angle, size=(H, W), mode="bilinear")

    def to(self, device):
        if self.angle is not None:
            self.angle = self.angle.to(device).contiguous()
        return self

    def fetch_reals_pool(self, pool_pol, pool_mask, pool_ctr, drop_const):
        """fetch_reals of batch (*pool_ctr % pool size) of a device-resident pool [P,B,1,H,W]: the index is read on the
        device (a hipGraph replay then moves on to the next batch by itself).  None unless the step's accumulator arena
        is open and H W % 256 == 0 (the per-sample sums are made in the same pass)."""
        P, B = pool_pol.shape[0], pool_pol.shape[1]
        HW = pool_pol[0, 0].numel()
        sums = L.AccArena.take(B, pool_pol.device) if HW % 256 == 0 else None
        if sums is None:
            return None
        out = torch.empty_like(pool_pol[0])
        L.check(L.lib().dg_fetch_reals_pool_sum(L.ptr(pool_pol), L.ptr(pool_mask), L.ptr(pool_ctr), P, self.min_depth,
                                                self.max_depth, float(drop_const), B, HW, L.ptr(out), L.ptr(sums),
                                                L.stream_ptr()), "dg_fetch_reals_pool_sum")
        return L.tag_sums(out, sums)

    def fetch_job(self, pool_pol, pool_mask, pool_ctr, drop_const):
        """fetch_reals as a job of the step's FIRST launch (dg_step_prologue_fetch): (DgFetch, out, parts) - `out` the batch's
        inverse-depth images, `parts` [B, XSUM_PARTS] the partial sums the launch stores - or None where that form does not apply
        (H W % (1024 XSUM_PARTS) != 0).  pool_ctr given: pool_pol / pool_mask are [P,B,1,H,W] pools and the batch is
        (*pool_ctr % P), read on the device; None: they ARE the batch [B,1,H,W].  The caller passes the job to
        `_lib.AccArena.begin` (or `_lib.step_prologue`) and tags out with the sums."""
        pooled = pool_ctr is not None
        P = pool_pol.shape[0] if pooled else 1
        first = pool_pol[0] if pooled else pool_pol
        B, HW = first.shape[0], first[0].numel()
        if (HW % (1024 * L.XSUM_PARTS) != 0 or pool_pol.dtype != torch.float32 or pool_mask.dtype != torch.float32
                or not pool_pol.is_contiguous() or not pool_mask.is_contiguous()):
            return None
        out = torch.empty_like(first)
        parts = torch.empty(B, L.XSUM_PARTS, dtype=torch.float32, device=pool_pol.device)
        f = L.DgFetch()
        f.pol, f.mask, f.pool_ctr, f.npool = L.ptr(pool_pol), L.ptr(pool_mask), L.ptr(pool_ctr), P
        f.min_depth, f.max_depth, f.drop_const = self.min_depth, self.max_depth, float(drop_const)
        f.B, f.HW, f.out, f.parts = B, HW, L.ptr(out), L.ptr(parts)
        return f, out, parts

    def fetch_reals(self, pol, mask, drop_const):
        """pol [B,1,H,W] in [0,1], mask [B,1,H,W] {0,1} float -> inverse depth in [-1,1], dropped pixels = drop_const"""
        pol = pol.contiguous().float()
        mask = mask.contiguous().float()
        B, HW = pol.shape[0], pol[0].numel()
        if L.AccArena.buf is not None and pol.is_cuda:
            # the kernel the graph-replayed step runs as part of its first launch, here as a launch of its own: the image AND
            # the per-sample sums (XSUM_PARTS stored partials) are then the same bits on both paths (round 6)
            job = self.fetch_job(pol, mask, None, drop_const)
            if job is not None:
                L.step_prologue([], [], fetch=job[0])
                return L.tag_sums(job[1], job[2], parts=L.XSUM_PARTS), mask
        out = torch.empty_like(pol)
        sums = L.AccArena.take(B, pol.device) if HW % 256 == 0 else None
        if sums is not None:  # per-sample sums of the result in the same pass (DiffAugment's contrast reads them)
            L.check(L.lib().dg_fetch_reals_sum(L.ptr(pol), L.ptr(mask), self.min_depth, self.max_depth, float(drop_const),
                                               B, HW, L.ptr(out), L.ptr(sums), L.stream_ptr()), "dg_fetch_reals_sum")
            return L.tag_sums(out, sums), mask
        L.check(L.lib().dg_fetch_reals(L.ptr(pol), L.ptr(mask), self.min_depth, self.max_depth, float(drop_const),
                                       pol.numel(), L.ptr(out), L.stream_ptr()), "dg_fetch_reals")
        return out, mask

    def inv_to_xyz(self, inv_depth, tol=1e-8, from_tanh=False, return_depth=False):
        """utils/lidar.py:58-65.  inv_depth [B,1,H,W] in [0,1] (or the generator's [-1,1] output with from_tanh=True,
        which applies utils/__init__.py:168 first) -> points [B,3,H,W] in unit space (metres / max_depth)."""
        if self.angle is None:
            raise RuntimeError(f"no angle grid: {self.angle_file!r} does not exist (utils/lidar.py:120)")
        if not inv_depth.is_cuda:
            raise RuntimeError("inv_to_xyz runs on the GPU only (no CPU fallback)")
        x = inv_depth.contiguous().float()
        B, _, H, W = x.shape
        assert (H, W) == (self.H, self.W) and x.shape[1] == 1
        if self.angle.device != x.device:
            self.to(x.device)
        pts = torch.empty(B, 3, H, W, dtype=torch.float32, device=x.device)
        d01 = torch.empty_like(x) if return_depth else None
        L.check(L.lib().dg_inv_to_xyz(L.ptr(x), L.ptr(self.angle), B, H, W, int(from_tanh), self.min_depth,
                                      self.max_depth, self.drop_const, float(tol), L.ptr(d01), L.ptr(pts),
                                      L.stream_ptr()), "dg_inv_to_xyz")
        return (pts, d01) if return_depth else pts


def unit_map(x, mode):
    """utils.postprocess's elementwise branches (utils/__init__.py:169-172): mode 0 tanh_to_sigmoid+clamp, 1 sigmoid"""
    if not x.is_cuda:
        raise RuntimeError("postprocess runs on the GPU only (no CPU fallback)")
    x = x.contiguous().float()
    y = torch.empty_like(x)
    L.check(L.lib().dg_unit_map(L.ptr(x), x.numel(), mode, L.ptr(y), L.stream_ptr()), "dg_unit_map")
    return y


def xyz_to_normal(xyz, mode="closest"):
    """utils/__init__.py:215-219 -> estimate_surface_normal utils/geometry.py:38-127 (d = 2): point map [B,3,H,W] ->
    normal image [B,3,H,W] in [0,1]"""
    if mode != "closest":
        raise NotImplementedError(mode)
    if not xyz.is_cuda:
        raise RuntimeError("xyz_to_normal runs on the GPU only (no CPU fallback)")
    xyz = xyz.contiguous().float()
    B, _, H, W = xyz.shape
    out = torch.empty_like(xyz)
    L.check(L.lib().dg_normals(L.ptr(xyz), B, H, W, 2, L.ptr(out), L.stream_ptr()), "dg_normals")
    return out


def postprocess(synth, lidar, tol=1e-8, normal_mode="closest"):
    """utils.postprocess utils/__init__.py:163-178: depth / depth_orig -> [0,1], confidence -> sigmoid, + "points" and
    "normals".  Without an angle grid there is no point map and both are omitted."""
    out = {}
    for key, value in synth.items():
        if key == "depth":
            if lidar.angle is not None:
                out["points"], out["depth"] = lidar.inv_to_xyz(value, tol, from_tanh=True, return_depth=True)
            else:
                out["depth"] = unit_map(value, 0)
        elif key == "depth_orig":
            out["depth_orig"] = unit_map(value, 0)
        elif key == "confidence":
            out["confidence"] = unit_map(value, 1)
        else:
            out[key] = value
    if "points" in out:
        out["normals"] = xyz_to_normal(out["points"], mode=normal_mode)
    return out
