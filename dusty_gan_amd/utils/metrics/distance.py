"""Chamfer distance between sets of point clouds -- reference: utils/metrics/distance/cd/ (CUDA extension) as used by
utils/metrics/cov_mmd_1nna.py:20-52.  The reference evaluates one row of the distance matrix per Python iteration;
here a whole [Na,Nb] matrix of directed means is one launch (csrc/metrics.hip chamfer_dir_kernel)."""
import torch

from ... import _lib as L


def _prep(pcs):
    assert pcs.ndim == 3 and pcs.size(2) == 3, "expected (B,N,3), but got {}".format(tuple(pcs.shape))
    if not pcs.is_cuda:
        raise RuntimeError("chamfer distance runs on the GPU only (no CPU fallback)")
    return pcs.contiguous().float()


def chamfer_dir(pcs_1, pcs_2):
    """L[i,j] = mean_{p in pcs_1[i]} min_{q in pcs_2[j]} |p - q|^2  -> [B_1,B_2]"""
    a, b = _prep(pcs_1), _prep(pcs_2)
    out = torch.empty(a.size(0), b.size(0), dtype=torch.float32, device=a.device)
    L.check(L.lib().dg_chamfer_dir(L.ptr(a), a.size(0), a.size(1), L.ptr(b), b.size(0), b.size(1), L.ptr(out),
                                   L.stream_ptr()), "dg_chamfer_dir")
    return out


def chamfer_distance_matrix(pcs_1, pcs_2):
    """M[i,j] = compute_cd(pcs_1[i], pcs_2[j]) = dl.mean + dr.mean (cov_mmd_1nna.py:20-22), all pairs"""
    if pcs_1 is pcs_2:
        d = chamfer_dir(pcs_1, pcs_1)
        return d + d.t()
    return chamfer_dir(pcs_1, pcs_2) + chamfer_dir(pcs_2, pcs_1).t()


def earth_mover_distance(xyz1, xyz2):
    """cost[b] of the approximate matching between xyz1[b] and xyz2[b] (reference: utils/metrics/distance/emd/,
    EarthMoverDistanceFunction.forward): approxmatch + matchcost in one kernel (csrc/metrics.hip emd_kernel)"""
    a, b = _prep(xyz1), _prep(xyz2)
    assert a.size(0) == b.size(0)
    out = torch.empty(a.size(0), dtype=torch.float32, device=a.device)
    L.check(L.lib().dg_emd(L.ptr(a), a.size(0), a.size(1), L.ptr(b), b.size(0), b.size(1), 1, L.ptr(out),
                           L.stream_ptr()), "dg_emd")
    return out


def emd_distance_matrix(pcs_1, pcs_2):
    """M[i,j] = compute_emd(pcs_1[i], pcs_2[j]) = cost / N (cov_mmd_1nna.py:12-17), all pairs in one launch"""
    a, b = _prep(pcs_1), _prep(pcs_2)
    assert a.size(1) == b.size(1)
    out = torch.empty(a.size(0), b.size(0), dtype=torch.float32, device=a.device)
    L.check(L.lib().dg_emd(L.ptr(a), a.size(0), a.size(1), L.ptr(b), b.size(0), b.size(1), 0, L.ptr(out),
                           L.stream_ptr()), "dg_emd")
    return out / float(a.size(1))
