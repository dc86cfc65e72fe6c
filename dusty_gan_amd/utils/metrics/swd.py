"""Sliced Wasserstein distance between Laplacian-pyramid patch descriptors -- reference: utils/metrics/swd.py:16-151.

Same functions, arguments and result keys.  The pyramid (5x5 gaussian down / zero-insert up + subtract, reflect
padding) and the patch gathering are kernels (csrc/metrics.hip pyr_down_kernel, pyr_up_sub_kernel, patches_kernel);
the projections are one library GEMM and one library sort per repeat on device tensors, as in the reference.
`rand` lets a caller inject the random draws (patch positions and directions) in the reference's draw order -- the
parity tests replay the reference's own draws that way; by default they come from torch's device generator.
"""
from collections import defaultdict

import numpy as np
import torch
from torch.nn.modules.utils import _pair

from ... import _lib as L


def _dev(x):
    if not x.is_cuda:
        raise RuntimeError("SWD runs on the GPU only (no CPU fallback)")
    return x.contiguous().float()


def pyramid_down(image):
    """:24-30"""
    image = _dev(image)
    B, C, H, W = image.shape
    out = torch.empty(B, C, H // 2, W // 2, dtype=torch.float32, device=image.device)
    L.check(L.lib().dg_pyr_down(L.ptr(image), B * C, H, W, L.ptr(out), L.stream_ptr()), "dg_pyr_down")
    return out


def laplacian_pyramid(images, num_levels):
    """:45-50 (works on a copy; the reference subtracts in place in its caller's batch)"""
    pyramid = [_dev(images).clone()]
    for _ in range(1, num_levels):
        pyramid.append(pyramid_down(pyramid[-1]))
        fine, coarse = pyramid[-2], pyramid[-1]
        B, C, H, W = fine.shape
        L.check(L.lib().dg_pyr_up_sub(L.ptr(fine), L.ptr(coarse), B * C, H, W, L.stream_ptr()), "dg_pyr_up_sub")
    return pyramid


def extract_patches(minibatch, patch_size, num_patches, inds=None):
    """:53-62; inds = randperm(nH * nW)[:num_patches] unless given"""
    pH, pW = patch_size
    minibatch = _dev(minibatch)
    B, C, H, W = minibatch.shape
    N = (H - pH + 1) * (W - pW + 1)
    if inds is None:
        inds = torch.randperm(N, device=minibatch.device)[:num_patches]
    inds = inds.to(device=minibatch.device, dtype=torch.int64).contiguous()
    out = torch.empty(B, inds.numel(), C, pH, pW, dtype=torch.float32, device=minibatch.device)
    L.check(L.lib().dg_extract_patches(L.ptr(minibatch), B, C, H, W, pH, pW, L.ptr(inds), inds.numel(), L.ptr(out),
                                       L.stream_ptr()), "dg_extract_patches")
    return out


def make_descriptors(minibatch, num_levels, patch_size, num_patches, inds=None):
    """:65-70"""
    pyramids = laplacian_pyramid(minibatch, num_levels)
    return {i: extract_patches(pyramids[i], patch_size, num_patches, None if inds is None else inds[i])
            for i in range(num_levels)}


def finalize_descriptors(desc):
    """:73-80"""
    if isinstance(desc, list):
        desc = torch.cat(desc, dim=0)
    B, N, C, H, W = desc.shape
    C_std, C_mean = torch.std_mean(desc, dim=(0, 1, 3, 4), keepdim=True)
    desc = (desc - C_mean) / (C_std + 1e-8)
    return desc.reshape(-1, C * H * W)


def sliced_wasserstein_distance(desc1, desc2, dir_repeats, dirs_per_repeat, dirs=None):
    """:83-96"""
    D = desc1.shape[1]
    distances = []
    for r in range(dir_repeats):
        d = torch.randn(D, dirs_per_repeat, device=desc1.device) if dirs is None else dirs[r].to(desc1.device).float()
        d = d / torch.std(d, dim=0, keepdim=True)
        proj1, _ = torch.sort(torch.matmul(desc1, d), dim=0)
        proj2, _ = torch.sort(torch.matmul(desc2, d), dim=0)
        distances.append(torch.mean(torch.abs(proj1 - proj2)))
    return torch.mean(torch.stack(distances))


@torch.no_grad()
def compute_swd(image1, image2, num_levels=None, patch_size=7, num_patches=128, dir_repeats=4, dirs_per_repeat=128,
                batch_size=128, rand=None):
    """:99-151"""
    assert image1.ndim == image2.ndim == 4, "(B,C,H,W) shape is required"
    assert image1.shape == image2.shape
    B, C, H, W = image1.shape
    patch_size = _pair(patch_size)
    if num_levels is None:
        num_levels = int(np.log2(min(H, W) // 16) + 1)
    desc1, desc2 = defaultdict(list), defaultdict(list)
    for mb, i in enumerate(range(0, B, batch_size)):
        inds = (None, None) if rand is None else rand["inds"][mb]
        batch1 = make_descriptors(image1[i:i + batch_size], num_levels, patch_size, num_patches, inds[0])
        batch2 = make_descriptors(image2[i:i + batch_size], num_levels, patch_size, num_patches, inds[1])
        for level in batch1.keys():
            desc1[level].append(batch1[level])
            desc2[level].append(batch2[level])
    result = {}
    for level in desc1.keys():
        result["swd-" + str(16 << level)] = sliced_wasserstein_distance(
            finalize_descriptors(desc1[level]), finalize_descriptors(desc2[level]), dir_repeats, dirs_per_repeat,
            None if rand is None else rand["dirs"][level])
    result["swd-mean"] = sum(result.values()) / len(result)
    for key, value in result.items():
        result[key] = value.item()
    return result
