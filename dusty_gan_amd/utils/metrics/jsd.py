"""Jensen-Shannon divergence between occupancy histograms -- reference: utils/metrics/jsd.py:11-116.
Voting (nearest in-sphere grid node per point) and the divergence are kernels (csrc/metrics.hip grid_vote_kernel,
jsd_kernel); the grid itself is a constant table built once per resolution."""
import torch

from ... import _lib as L

_GRIDS = {}


def unit_cube_grid_point_cloud(resolution, clip_sphere, device):
    """:11-21"""
    spacing = 1.0 / float(resolution - 1)
    steps = torch.arange(resolution, device=device)
    grid = torch.stack(torch.meshgrid(steps, steps, steps, indexing="ij"), dim=-1) * spacing - 0.5
    if clip_sphere:
        grid = grid.reshape(-1, 3)
        grid = grid[torch.norm(grid, dim=1) <= 0.5]
    return grid, spacing


def grid_counters(pcs, resolution=28):
    """the `grid_counters` output of entropy_of_occupancy_grid(pcs, resolution, in_sphere=True) (:24-79)"""
    if not pcs.is_cuda:
        raise RuntimeError("JSD runs on the GPU only (no CPU fallback)")
    key = (resolution, pcs.device)
    if key not in _GRIDS:
        # built on the host with the reference's own expression so the node coordinates (and ties) are identical
        _GRIDS[key] = unit_cube_grid_point_cloud(resolution, True, "cpu")[0].reshape(-1, 3).float().contiguous().to(
            pcs.device)
    grid = _GRIDS[key]
    pts = pcs.reshape(-1, 3).contiguous().float()
    counters = torch.zeros(grid.size(0), dtype=torch.float32, device=pcs.device)
    L.check(L.lib().dg_grid_vote(L.ptr(pts), pts.size(0), L.ptr(grid), grid.size(0), L.ptr(counters), L.stream_ptr()),
            "dg_grid_vote")
    return counters


@torch.no_grad()
def compute_jsd(pcs_gen, pcs_ref, resolution=28, batchsize=128, verbose=True):
    """:110-116 (`batchsize` / `verbose` only shaped the reference's chunked loop)"""
    P, Q = grid_counters(pcs_gen, resolution), grid_counters(pcs_ref, resolution)
    out = torch.empty(1, dtype=torch.float32, device=P.device)
    L.check(L.lib().dg_jsd(L.ptr(P), L.ptr(Q), P.numel(), L.ptr(out), L.stream_ptr()), "dg_jsd")
    return out.item()
