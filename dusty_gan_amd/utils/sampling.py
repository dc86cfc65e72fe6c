"""Furthest point sampling -- reference: utils/sampling/fps/furthest_point_sampling.py:21-93 (CUDA extension
furthest_point_sampling.cu).  One kernel (csrc/metrics.hip fps_kernel) selects and gathers."""
import torch

from .. import _lib as L


def _check(xyz):
    assert xyz.ndim == 3, "expected 3-dim, but got {}-dim tensor".format(xyz.ndim)
    assert xyz.size(2) == 3, "expected (B,N,3), but got {}".format(xyz.shape)
    assert xyz.is_cuda  # same assertion as the reference (:87): there is no CPU path
    return xyz.contiguous().float()


def _run(xyz, k, gather):
    xyz = _check(xyz)
    B, N, _ = xyz.shape
    idx = torch.empty(B, k, dtype=torch.int32, device=xyz.device)
    temp = torch.empty(B, N, dtype=torch.float32, device=xyz.device)
    out = torch.empty(B, k, 3, dtype=torch.float32, device=xyz.device) if gather else None
    L.check(L.lib().dg_fps(L.ptr(xyz), B, N, int(k), L.ptr(temp), L.ptr(idx), L.ptr(out), L.stream_ptr()), "dg_fps")
    return idx, out


def furthest_point_sampling(xyz, npoint):
    """(B,N,3) -> (B,npoint) int32 indices (:21-43)"""
    return _run(xyz, npoint, False)[0]


def downsample_point_clouds(xyz, k):
    """(B,N,3) -> (B,k,3) (:84-93)"""
    return _run(xyz, k, True)[1]
