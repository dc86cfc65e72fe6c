"""CPU oracle for the validation metrics.  TEST INFRASTRUCTURE ONLY -- same rules as
oracle/dusty_oracle.py: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import it.

Restates, with numpy / stock torch CPU ops:
  * furthest point sampling + gather     utils/sampling/fps/furthest_point_sampling.cu:97-207, .py:84-93
  * Chamfer nearest-neighbour search      utils/metrics/distance/cd/chamfer_distance.cpp:41-66 (the extension's own CPU
                                          path `nnsearch`; the CUDA kernel computes the same quantity)
  * approximate earth mover's distance    utils/metrics/distance/emd/earth_mover_distance.cu:28-190,218-262
  * COV / MMD / 1-NNA                     utils/metrics/cov_mmd_1nna.py:12-148
  * JSD on the occupancy grid             utils/metrics/jsd.py:11-116
  * sliced Wasserstein distance           utils/metrics/swd.py:16-151 (random draws injected)

Parity pin:
  * JSD and SWD: utils/metrics/jsd.py and swd.py are plain torch and importable here; tests/golden/metrics.npz holds
    their outputs for seeded inputs (JSD: grid counters and the divergence; SWD: every score, with the reference's
    randperm / randn draws captured by replaying the generator) -> pinned.
  * FPS, Chamfer, EMD, COV/MMD/1-NNA: the reference runs them through CUDA extensions that are JIT-compiled at import
    (`torch.utils.cpp_extension.load` of .cu files, nvcc absent here; cov_mmd_1nna.py imports them at module level), so
    none of it can be executed in this image -> PARITY UNPINNED against the reference; restated from the sources
    cited above, FPS including the launcher's tie-breaking order, and cross-checked against brute force in the tests.
"""
import numpy as np
import torch


# --------------------------------------------------------------------------
# utils/sampling/fps
# --------------------------------------------------------------------------
def opt_n_threads(n):
    """furthest_point_sampling.cu `opt_n_threads`: largest power of two <= min(n, 512)"""
    t = 1
    while t * 2 <= n and t * 2 <= 512:
        t *= 2
    return t


def fps(xyz, m):
    """furthest_point_sampling_kernel (:97-207) for one cloud xyz [n,3] float32 -> indices [m] (int32).
    float32 arithmetic without fused multiply-add; ties resolved as the kernel does: thread t scans k = t, t+T, ...
    keeping the first strictly greater value; the tree reduction (:148-201) folds slot t+w into slot t for w = T/2 ..
    1 and keeps the lower SLOT on ties, so two threads meet at the lowest bit in which their ids differ and the one
    with that bit clear wins: among equal maxima the winner minimises (bit-reversed (k mod T), k)."""
    xyz = np.asarray(xyz, np.float32)
    n = xyz.shape[0]
    T = opt_n_threads(n)
    temp = np.full(n, 1e10, np.float32)
    mag = (xyz[:, 0] * xyz[:, 0]) + (xyz[:, 1] * xyz[:, 1]) + (xyz[:, 2] * xyz[:, 2])
    cand = mag > np.float32(1e-3)
    bits = max(T.bit_length() - 1, 0)
    rev = np.array([int(format(t, "0{}b".format(bits))[::-1], 2) if bits else 0 for t in range(T)])
    order = np.lexsort((np.arange(n), rev[np.arange(n) % T]))  # candidates in tie-priority order
    idx = np.zeros(m, np.int32)
    old = 0
    for j in range(1, m):
        d = xyz - xyz[old]
        d = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        temp[cand] = np.minimum(d[cand], temp[cand])
        if not cand.any():
            old = 0
        else:
            v = np.where(cand, temp, np.float32(-1.0))[order]
            old = int(order[int(np.argmax(v))])  # argmax returns the first maximum in priority order
        idx[j] = old
    return idx


def downsample_point_clouds(xyz, k):
    """furthest_point_sampling.py:84-93: xyz [B,N,3] -> [B,k,3]"""
    xyz = np.asarray(xyz, np.float32)
    return np.stack([c[fps(c, k)] for c in xyz])


# --------------------------------------------------------------------------
# utils/metrics/distance/cd + utils/metrics/cov_mmd_1nna.py
# --------------------------------------------------------------------------
def chamfer_dir(A, B):
    """L[i,j] = mean_p min_q |p - q|^2 over p in A[i], q in B[j]  (nnsearch chamfer_distance.cpp:41-66, `dist` averaged
    as compute_cd cov_mmd_1nna.py:20-22 does).  A [Na,n,3], B [Nb,m,3] -> [Na,Nb] float32."""
    A, B = torch.as_tensor(A, dtype=torch.float32), torch.as_tensor(B, dtype=torch.float32)
    out = torch.empty(A.shape[0], B.shape[0])
    for i in range(A.shape[0]):
        d = A[i][None, :, None, :] - B[:, None, :, :]          # [Nb,n,m,3]
        d = (d[..., 0] * d[..., 0] + d[..., 1] * d[..., 1]) + d[..., 2] * d[..., 2]
        out[i] = d.min(dim=2).values.mean(dim=1)
    return out


def pairwise_cd(pcs_1, pcs_2):
    """_pairwise_distance(..., metrics=("cd",)) cov_mmd_1nna.py:25-52: M[i,j] = dl.mean + dr.mean"""
    return chamfer_dir(pcs_1, pcs_2) + chamfer_dir(pcs_2, pcs_1).t()


def emd_cost(x1, x2):
    """approxmatch + matchcost of utils/metrics/distance/emd/earth_mover_distance.cu:28-190,218-262 for ONE pair
    x1 [n,3], x2 [m,3] (float32): ten annealing levels exp(-4^j d^2), j = 7 .. -1, then 0; the cost is
    sum_{k,l} match[l][k] * d^2(k,l), accumulated level by level (matchcost is linear in match)."""
    x1, x2 = np.asarray(x1, np.float32), np.asarray(x2, np.float32)
    n, m = x1.shape[0], x2.shape[0]
    multiL, multiR = (1.0, float(n // m)) if n >= m else (float(m // n), 1.0)
    remainL, remainR = np.full(n, multiL, np.float32), np.full(m, multiR, np.float32)
    diff = x2[None, :, :] - x1[:, None, :]
    d2 = (diff[..., 0] * diff[..., 0] + diff[..., 1] * diff[..., 1] + diff[..., 2] * diff[..., 2]).astype(np.float32)
    cost = 0.0
    for j in range(7, -3, -1):
        level = np.float32(0.0) if j == -2 else np.float32(-(4.0 ** j))
        E = np.exp(level * d2).astype(np.float32)                       # [n,m]
        suml = np.float32(1e-9) + E @ remainR
        ratioL = remainL / suml
        sumr = (E.T @ ratioL) * remainR
        consumption = np.minimum(remainR / (sumr + np.float32(1e-9)), np.float32(1.0))
        ratioR = consumption * remainR
        remainR = np.maximum(np.float32(0.0), remainR - sumr)
        Wm = E * ratioL[:, None] * ratioR[None, :]
        cost += float((Wm * d2).sum(dtype=np.float64))
        remainL = np.maximum(np.float32(0.0), remainL - Wm.sum(axis=1))
    return cost


def pairwise_emd(pcs_1, pcs_2):
    """_pairwise_distance(..., metrics=("emd",)) cov_mmd_1nna.py:25-52 with compute_emd :12-17 (cost / N)"""
    out = torch.empty(len(pcs_1), len(pcs_2))
    for i, a in enumerate(pcs_1):
        for j, b in enumerate(pcs_2):
            out[i, j] = emd_cost(a, b) / float(a.shape[0])
    return out


def cov_mmd(M_rg):
    """_compute_cov_mmd cov_mmd_1nna.py:55-66"""
    N_ref, N_gen = M_rg.shape
    mmd_gen, min_idx_gen = M_rg.min(dim=0)
    mmd_ref, _ = M_rg.min(dim=1)
    return {"mmd": mmd_ref.mean().item(), "mmd-sample": mmd_gen.mean().item(),
            "cov": float(len(torch.unique(min_idx_gen))) / float(N_ref)}


def nna(M_rr, M_rg, M_gg, k=1, sqrt=False):
    """_compute_nna cov_mmd_1nna.py:69-110"""
    N_ref, N_gen = M_rg.shape
    label = torch.cat([torch.ones(N_ref), torch.zeros(N_gen)], dim=0)
    M_ref = torch.cat((M_rr, M_rg), dim=1)
    M_gen = torch.cat((M_rg.t(), M_gg), dim=1)
    M = torch.cat([M_ref, M_gen], dim=0)
    M = M.abs().sqrt() if sqrt else M
    M = M + torch.diag(float("inf") * torch.ones_like(label))
    _, idx = M.topk(k=k, dim=0, largest=False)
    count = torch.zeros_like(label)
    for i in range(0, k):
        count = count + label.index_select(0, idx[i])
    pred = (count / k >= 0.5).float()
    s = {"tp": (pred * label).sum().item(), "fp": (pred * (1 - label)).sum().item(),
         "fn": ((1 - pred) * label).sum().item(), "tn": ((1 - pred) * (1 - label)).sum().item()}
    s.update({"precision": s["tp"] / (s["tp"] + s["fp"] + 1e-10), "recall": s["tp"] / (s["tp"] + s["fn"] + 1e-10),
              "accuracy_t": s["tp"] / (s["tp"] + s["fn"] + 1e-10), "accuracy_f": s["tn"] / (s["tn"] + s["fp"] + 1e-10),
              "accuracy": torch.eq(label, pred).float().mean().item()})
    return s


def compute_cov_mmd_1nna(pcs_gen, pcs_ref, metrics=("cd",)):
    """compute_cov_mmd_1nna(pcs_gen, pcs_ref, batch, metrics) cov_mmd_1nna.py:113-148 (result key names included)"""
    res = {}
    for metric in metrics:
        pw = {"cd": pairwise_cd, "emd": pairwise_emd}[metric]
        M_rr, M_rg, M_gg = pw(pcs_ref, pcs_ref), pw(pcs_ref, pcs_gen), pw(pcs_gen, pcs_gen)
        res.update({"{}-{}".format(k, metric): v for k, v in cov_mmd(M_rg).items()})
        res.update({"1-nn-{}-{}".format(k, metric): v for k, v in nna(M_rr, M_rg, M_gg, k=1).items()})
    return res


# --------------------------------------------------------------------------
# utils/metrics/jsd.py
# --------------------------------------------------------------------------
def unit_cube_grid(resolution, clip_sphere=True):
    """unit_cube_grid_point_cloud jsd.py:11-21"""
    spacing = 1.0 / float(resolution - 1)
    steps = torch.arange(resolution)
    grid = torch.stack(torch.meshgrid(steps, steps, steps, indexing="ij"), dim=-1) * spacing - 0.5
    if clip_sphere:
        grid = grid.reshape(-1, 3)
        grid = grid[torch.norm(grid, dim=1) <= 0.5]
    return grid.reshape(-1, 3)


def grid_counters(pcs, resolution=28):
    """the histogram half of entropy_of_occupancy_grid jsd.py:24-79 (the entropy half is discarded by compute_jsd):
    every point votes for its nearest in-sphere grid node (first index on ties, as argmin)"""
    pcs = torch.as_tensor(pcs, dtype=torch.float32)
    grid = unit_cube_grid(resolution)
    flat = pcs.reshape(-1, 3)
    inds = torch.empty(flat.shape[0], dtype=torch.long)
    for s in range(0, flat.shape[0], 4096):
        d = (flat[s:s + 4096, None] - grid[None]).pow(2).sum(dim=-1)
        inds[s:s + 4096] = d.argmin(dim=1)
    counters = torch.zeros(len(grid))
    counters.scatter_add_(0, inds, torch.ones_like(inds).float())
    return counters


def _entropy(p, base=2, eps=1e-8):
    p = p + eps
    return (-p * torch.log2(p)).sum(dim=-1)


def jensen_shannon_divergence(P, Q):
    """_jensen_shannon_divergence jsd.py:96-107 (note: `_entropy` adds eps IN PLACE, so P_ and Q_ carry +1e-8 into the
    mixture term, reproduced here)"""
    P_ = P / P.sum()
    Q_ = Q / Q.sum()
    P_ = P_ + 1e-8
    Q_ = Q_ + 1e-8
    e1 = (-P_ * torch.log2(P_)).sum()
    e2 = (-Q_ * torch.log2(Q_)).sum()
    mix = (P_ + Q_) / 2.0 + 1e-8
    e_sum = (-mix * torch.log2(mix)).sum()
    return e_sum - ((e1 + e2) / 2.0)


def compute_jsd(pcs_gen, pcs_ref, resolution=28):
    """compute_jsd jsd.py:110-116"""
    return jensen_shannon_divergence(grid_counters(pcs_gen, resolution), grid_counters(pcs_ref, resolution)).item()


# --------------------------------------------------------------------------
# utils/metrics/swd.py  (plain torch, importable here -> pinned by tests/golden/metrics.npz "swd/*")
# --------------------------------------------------------------------------
def _gauss_kernel(weight):
    """get_kernel swd.py:16-21"""
    k = torch.tensor(weight).float()
    k = torch.outer(k, k)
    k /= k.sum()
    return k[None, None]


def pyramid_down(image):
    """swd.py:24-30"""
    C = image.shape[1]
    g = _gauss_kernel([1, 4, 6, 4, 1]).repeat(C, 1, 1, 1)
    return torch.nn.functional.conv2d(torch.nn.functional.pad(image, (2, 2, 2, 2), mode="reflect"), g, stride=2,
                                      padding=0, groups=C)


def pyramid_up(image):
    """swd.py:33-42"""
    C = image.shape[1]
    dil = _gauss_kernel([0, 1, 0]).repeat(C, 1, 1, 1)
    dilated = torch.nn.functional.conv_transpose2d(image, dil, stride=2, padding=0, groups=C)
    padded = torch.nn.functional.pad(dilated[..., :-1, :-1], (2, 2, 2, 2), mode="reflect")
    g = _gauss_kernel([1, 4, 6, 4, 1]).repeat(C, 1, 1, 1) * 4
    return torch.nn.functional.conv2d(padded, g, stride=1, padding=0, groups=C)


def laplacian_pyramid(images, num_levels):
    """swd.py:45-50 (the reference subtracts in place, i.e. it also modifies the caller's batch slice)"""
    pyramid = [images.clone()]
    for _ in range(1, num_levels):
        pyramid.append(pyramid_down(pyramid[-1]))
        pyramid[-2] = pyramid[-2] - pyramid_up(pyramid[-1])
    return pyramid


def extract_patches(minibatch, patch_size, inds):
    """swd.py:53-62 with the randperm draw injected: inds = randperm(nH*nW)[:num_patches]"""
    pH, pW = patch_size
    patches = minibatch.unfold(2, pH, 1).unfold(3, pW, 1)
    B, C, nH, nW, pH, pW = patches.shape
    patches = patches.reshape(B, C, nH * nW, pH, pW).transpose(1, 2)
    return patches.index_select(dim=1, index=inds)


def swd_num_levels(H, W):
    """swd.py:131-132"""
    return int(np.log2(min(H, W) // 16) + 1)


def swd_patch_counts(H, W, num_levels, patch_size=(7, 7)):
    """number of patch positions per pyramid level (the N of randperm(N), swd.py:60)"""
    return [((H >> l) - patch_size[0] + 1) * ((W >> l) - patch_size[1] + 1) for l in range(num_levels)]


def finalize_descriptors(desc):
    """swd.py:73-80"""
    desc = torch.cat(desc, dim=0)
    B, N, C, H, W = desc.shape
    C_std, C_mean = torch.std_mean(desc, dim=(0, 1, 3, 4), keepdim=True)
    desc = (desc - C_mean) / (C_std + 1e-8)
    return desc.reshape(-1, C * H * W)


def sliced_wasserstein_distance(desc1, desc2, dirs_list):
    """swd.py:83-96 with the direction draws injected: dirs_list[r] = randn(D, dirs_per_repeat) BEFORE normalisation"""
    distances = []
    for dirs in dirs_list:
        dirs = dirs / torch.std(dirs, dim=0, keepdim=True)
        proj1, _ = torch.sort(torch.matmul(desc1, dirs), dim=0)
        proj2, _ = torch.sort(torch.matmul(desc2, dirs), dim=0)
        distances.append(torch.mean(torch.abs(proj1 - proj2)))
    return torch.mean(torch.stack(distances))


def compute_swd(image1, image2, rand, patch_size=(7, 7), batch_size=128):
    """compute_swd swd.py:99-151.  rand = {"inds": [minibatch][set][level] -> LongTensor[num_patches],
    "dirs": [level][repeat] -> Tensor[D, dirs_per_repeat]} in the reference's draw order."""
    image1, image2 = torch.as_tensor(image1).float(), torch.as_tensor(image2).float()
    B, C, H, W = image1.shape
    L = swd_num_levels(H, W)
    desc1, desc2 = [[] for _ in range(L)], [[] for _ in range(L)]
    for mb, i in enumerate(range(0, B, batch_size)):
        for which, (img, desc) in enumerate(((image1, desc1), (image2, desc2))):
            pyr = laplacian_pyramid(img[i:i + batch_size], L)
            for l in range(L):
                desc[l].append(extract_patches(pyr[l], patch_size, rand["inds"][mb][which][l]))
    result = {}
    for l in range(L):
        result["swd-" + str(16 << l)] = sliced_wasserstein_distance(finalize_descriptors(desc1[l]),
                                                                    finalize_descriptors(desc2[l]), rand["dirs"][l])
    result["swd-mean"] = sum(result.values()) / len(result)
    return {k: v.item() for k, v in result.items()}
