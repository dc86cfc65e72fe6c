"""GPU parity tests for the validation-metric kernels: csrc/metrics.hip through the C ABI and the
host glue of dusty_gan_amd/utils/metrics against oracle/metrics_oracle.py and tests/golden/metrics.npz."""
import math

import numpy as np
import pytest
import torch

from oracle import metrics_oracle as MO
from tests.golden_util import load, rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"


def lidar_like_clouds(B, n, seed, drop=0.15):
    """points on a jittered spherical grid in unit space with dropped returns at the origin and a duplicate"""
    rng = np.random.default_rng(seed)
    pitch = rng.uniform(-0.43, 0.06, (B, n))
    yaw = rng.uniform(-math.pi, math.pi, (B, n))
    r = np.exp(rng.uniform(math.log(0.01), math.log(0.9), (B, n)))
    pts = np.stack([r * np.cos(pitch) * np.cos(yaw), r * np.cos(pitch) * np.sin(yaw), r * np.sin(pitch)], -1)
    pts[rng.random((B, n)) < drop] = 0.0
    pts[:, n // 2] = pts[:, n // 3]  # exact duplicates -> exact ties in the FPS arg-max
    return pts.astype(np.float32)


@pytest.mark.parametrize("n,m", [(70, 20), (512, 64), (1000, 128), (4096, 512), (8192, 100)])
def test_fps_matches_oracle_indices(n, m):
    """index-exact furthest point sampling incl. the reference launcher's tie order and the origin-skip rule"""
    from dusty_gan_amd.utils.sampling import downsample_point_clouds, furthest_point_sampling
    B = 3
    pts = lidar_like_clouds(B, n, seed=n + m)
    idx = furthest_point_sampling(torch.from_numpy(pts).to(DEV), m).cpu().numpy()
    sub = downsample_point_clouds(torch.from_numpy(pts).to(DEV), m).cpu().numpy()
    for b in range(B):
        want = MO.fps(pts[b], m)
        assert idx[b].tolist() == want.tolist(), b
        assert np.array_equal(sub[b], pts[b][want])
    with pytest.raises(AssertionError):
        furthest_point_sampling(torch.from_numpy(pts), m)  # the reference asserts is_cuda too


def test_fps_degenerate_clouds():
    from dusty_gan_amd.utils.sampling import furthest_point_sampling
    z = torch.zeros(2, 300, 3, device=DEV)
    assert furthest_point_sampling(z, 5).cpu().tolist() == [[0] * 5] * 2  # nothing is a candidate
    one = z.clone()
    one[0, 17] = torch.tensor([0.3, 0.1, 0.0])
    assert furthest_point_sampling(one, 3).cpu()[0].tolist() == [0, 17, 17]


@pytest.mark.parametrize("Na,n,Nb,m", [(5, 512, 7, 512), (9, 100, 4, 333), (3, 1024, 6, 700), (2, 3000, 5, 2048),
                                       (70, 512, 33, 512)])
def test_chamfer_dir_matches_oracle(Na, n, Nb, m):
    """all-pairs directed Chamfer means (every wave-per-cloud layout of the kernel: 1, 2 and whole workgroups, ragged
    cloud sizes, partially filled last workgroup) against the restated nnsearch; tolerance 1e-5 relative"""
    from dusty_gan_amd.utils.metrics import chamfer_dir, chamfer_distance_matrix
    A = lidar_like_clouds(Na, n, seed=Na + n)
    B = lidar_like_clouds(Nb, m, seed=Nb + m + 1)
    L = chamfer_dir(torch.from_numpy(A).to(DEV), torch.from_numpy(B).to(DEV)).cpu()
    want = MO.chamfer_dir(A, B)
    assert L.shape == (Na, Nb)
    assert float(((L - want).abs() / want.abs().clamp_min(1e-12)).max()) < 1e-5
    M = chamfer_distance_matrix(torch.from_numpy(A).to(DEV), torch.from_numpy(B).to(DEV)).cpu()
    assert rel_l2(M, MO.pairwise_cd(A, B)) < 1e-5
    At = torch.from_numpy(A).to(DEV)
    S = chamfer_distance_matrix(At, At).cpu()
    assert torch.equal(S, S.t()) and float(S.diag().abs().max()) == 0.0
    with pytest.raises(RuntimeError):
        chamfer_dir(torch.from_numpy(A), torch.from_numpy(B))


def test_cov_mmd_1nna_matches_oracle():
    from dusty_gan_amd.utils.metrics import compute_cov_mmd_1nna
    gen = lidar_like_clouds(24, 256, seed=1) * 0.8
    ref = lidar_like_clouds(31, 256, seed=2)
    got = compute_cov_mmd_1nna(torch.from_numpy(gen).to(DEV), torch.from_numpy(ref).to(DEV), 512, ("cd",), verbose=False)
    want = MO.compute_cov_mmd_1nna(gen, ref)
    assert set(got) == set(want) == {"mmd-cd", "mmd-sample-cd", "cov-cd", "1-nn-tp-cd", "1-nn-fp-cd", "1-nn-fn-cd",
                                     "1-nn-tn-cd", "1-nn-precision-cd", "1-nn-recall-cd", "1-nn-accuracy_t-cd",
                                     "1-nn-accuracy_f-cd", "1-nn-accuracy-cd"}
    for k in want:
        assert abs(got[k] - want[k]) <= 1e-5 * max(1.0, abs(want[k])), (k, got[k], want[k])
    with pytest.raises(NotImplementedError):
        compute_cov_mmd_1nna(torch.from_numpy(gen).to(DEV), torch.from_numpy(ref).to(DEV), 512, ("emd",))


def test_jsd_matches_reference_golden():
    """utils/metrics/jsd.py: counters bit-exact, divergence to 1e-5, against the reference's own outputs"""
    from dusty_gan_amd.utils.metrics import compute_jsd
    from dusty_gan_amd.utils.metrics.jsd import grid_counters
    g = load("metrics")
    for name in ("gen", "ref"):
        c = grid_counters(torch.from_numpy(g[f"pcs_{name}"]).to(DEV)).cpu()
        assert torch.equal(c, torch.from_numpy(g[f"counters_{name}"])), name
    got = compute_jsd(torch.from_numpy(g["pcs_gen"]).to(DEV), torch.from_numpy(g["pcs_ref"]).to(DEV), verbose=False)
    assert abs(got - float(g["jsd"])) < 1e-5
    # a larger seeded case against the oracle (pinned to the same vectors on the CPU)
    a = lidar_like_clouds(40, 512, seed=5) / 2.0
    b = lidar_like_clouds(40, 512, seed=6) / 2.2
    ca = grid_counters(torch.from_numpy(a).to(DEV)).cpu()
    assert torch.equal(ca, MO.grid_counters(a))
    assert abs(compute_jsd(torch.from_numpy(a).to(DEV), torch.from_numpy(b).to(DEV)) - MO.compute_jsd(a, b)) < 1e-5
