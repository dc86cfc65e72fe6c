"""GPU parity tests for the data formats either side of the step: csrc/lidar_io.hip through the
C ABI and the host pipeline of dusty_gan_amd/datasets against oracle/lidar_oracle.py and tests/golden/lidar.npz."""
import math
import os

import numpy as np
import pytest
import torch

from oracle import dusty_oracle as O
from oracle import lidar_oracle as LO
from tests.golden_util import load, rel_l2

pytestmark = pytest.mark.gpu
DEV = "cuda"


def make_scans(B, Hs, Ws, C, seed):
    """range images with holes, out-of-range returns and cells sitting exactly on the min / max thresholds"""
    rng = np.random.default_rng(seed)
    d = np.exp(rng.uniform(math.log(0.3), math.log(200.0), (B, Hs, Ws))).astype(np.float32)
    pitch = rng.uniform(-0.45, 0.1, (B, Hs, Ws))
    yaw = rng.uniform(-math.pi, math.pi, (B, Hs, Ws))
    xyz = np.stack([d * np.cos(pitch) * np.cos(yaw), d * np.cos(pitch) * np.sin(yaw), d * np.sin(pitch)], -1)
    xyz[rng.random((B, Hs, Ws)) < 0.15] = 0.0            # no return
    edge = rng.random((B, Hs, Ws))
    xyz[edge < 0.02] = (0.0, np.float32(0.9), 0.0)       # |.| == min_depth -> invalid
    xyz[(edge > 0.02) & (edge < 0.04)] = (120.0, 0.0, 0.0)  # |.| == max_depth -> invalid
    pts = np.concatenate([xyz, rng.random((B, Hs, Ws, C - 3))], -1) if C > 3 else xyz
    return pts.astype(np.float32)


@pytest.mark.parametrize("Hs,Ws,C,H,W", [(64, 2048, 4, 64, 1024), (64, 2048, 4, 64, 256), (32, 1024, 4, 32, 256),
                                         (16, 100, 3, 7, 33), (8, 64, 5, 8, 64)])
def test_scan_to_polar_matches_oracle(Hs, Ws, C, H, W):
    """bit-exact polar depth / mask / xyz against the numpy restatement of datasets/kitti.py:54-77, with and without
    the horizontal flip; the fused network input against fetch_reals of the oracle (1e-5)"""
    from dusty_gan_amd.datasets.scans import scan_to_polar
    B = 3
    pts = make_scans(B, Hs, Ws, C, seed=Hs + W)
    flips = [False, True, True]
    out = scan_to_polar(torch.from_numpy(pts).to(DEV), (H, W), 0.9, 120.0, flip=torch.tensor(flips), want_xyz=True,
                        drop_const=-1.0)
    assert out["depth"].shape == (B, 1, H, W) and out["xyz"].shape == (B, 3, H, W)
    for b in range(B):
        ref = LO.scan_to_polar(pts[b], (H, W), flip=flips[b])
        assert torch.equal(out["mask"][b].cpu() > 0, ref["mask"]), b
        assert torch.equal(out["depth"][b].cpu(), ref["depth"]), b
        assert torch.equal(out["xyz"][b].cpu(), ref["xyz"]), b
        x_ref, _ = O.fetch_reals(ref["depth"], ref["mask"])
        assert (out["x_real"][b].cpu() - x_ref).abs().max() < 1e-5
        assert 0.05 < float(ref["mask"].float().mean()) < 0.95  # the case exercises both branches
    nofl = scan_to_polar(torch.from_numpy(pts).to(DEV), (H, W), 0.9, 120.0)
    assert set(nofl) == {"depth", "mask"} and torch.equal(nofl["depth"][0], out["depth"][0])
    with pytest.raises(RuntimeError):
        scan_to_polar(torch.from_numpy(pts), (H, W), 0.9, 120.0)  # host tensor: no CPU fallback


def test_scan_to_polar_matches_reference_kitti_preprocess():
    """the HIP kernel (dg_scan_to_polar) against what the reference's own KITTIOdometry.preprocess returned
    (tests/golden/kitti_pre.npz, datasets/kitti.py:54-67), at the scans' own size (the NEAREST resize is then the
    identity): depth, mask and xyz bit-exact, including points exactly on the range limits."""
    from dusty_gan_amd.datasets.scans import scan_to_polar
    from tests.golden_util import load
    g = load("kitti_pre")
    for k in range(3):
        pts = torch.from_numpy(g[f"s{k}/points"])[None].to(DEV)
        Hs, Ws = pts.shape[1:3]
        out = scan_to_polar(pts, (Hs, Ws), float(g["meta/min_depth"]), float(g["meta/max_depth"]), want_xyz=True)
        assert torch.equal(out["mask"][0, 0].cpu() > 0, torch.from_numpy(g[f"s{k}/mask"])), k
        assert torch.equal(out["depth"][0, 0].cpu(), torch.from_numpy(g[f"s{k}/depth"])), k
        assert torch.equal(out["xyz"][0].cpu(), torch.from_numpy(g[f"s{k}/xyz"]).permute(2, 0, 1)), k


def test_inv_to_xyz_and_postprocess_match_reference_golden(tmp_path):
    """Coordinate.inv_to_xyz / utils.postprocess against vectors from the reference's LiDAR class"""
    from dusty_gan_amd.utils.lidar import LiDAR, postprocess
    g = load("lidar")
    H, W = (int(v) for v in g["meta/shape"])
    path = os.path.join(tmp_path, "angles.pt")
    torch.save(torch.from_numpy(g["angle_src"]), path)
    lidar = LiDAR(H, W, float(g["meta/min_depth"]), float(g["meta/max_depth"]), angle_file=path).to(DEV)
    assert rel_l2(lidar.angle.cpu(), g["angle"]) < 1e-6
    inv = torch.from_numpy(g["inv"])
    pts = lidar.inv_to_xyz(inv.to(DEV)).cpu()
    assert rel_l2(pts, g["points"]) < 1e-5
    dropped = (inv == 0).expand(-1, 3, -1, -1)
    assert torch.equal(pts[dropped], torch.zeros_like(pts[dropped]))
    gen = torch.from_numpy(g["gen_depth"])
    conf = torch.randn(3, 2, H, W)
    out = postprocess({"depth": gen.to(DEV), "depth_orig": gen.to(DEV), "confidence": conf.to(DEV),
                       "mask": torch.ones(1, device=DEV)}, lidar)
    assert set(out) == {"depth", "depth_orig", "confidence", "mask", "points", "normals"}
    assert rel_l2(out["points"].cpu(), g["gen_points"]) < 1e-5
    # normal images: from the reference's own point maps (argmin ties / near-ties would make a pixel jump, so the
    # reference points are the input here), then end to end
    from dusty_gan_amd.utils.lidar import xyz_to_normal
    for a, b in (("points", "normals"), ("gen_points", "gen_normals")):
        nrm = xyz_to_normal(torch.from_numpy(g[a]).to(DEV)).cpu()
        # n = c / (|c| + 1e-8) with c a cross product of short, sometimes nearly collinear difference vectors: the
        # rounding of c is amplified by |v1||v2| / |c|, so single pixels move by up to ~1e-3 while the image agrees to 1e-5
        err = (nrm - torch.from_numpy(g[b])).abs()
        assert float(err.mean()) < 1e-5 and float(err.max()) < 5e-3, (a, float(err.mean()), float(err.max()))
    assert float((out["normals"].cpu() - torch.from_numpy(g["gen_normals"])).abs().mean()) < 1e-3
    assert torch.allclose(out["depth"].cpu(), ((gen + 1) / 2).clamp(0, 1), atol=1e-7)
    assert torch.equal(out["depth"], out["depth_orig"])
    assert torch.allclose(out["confidence"].cpu(), torch.sigmoid(conf), atol=1e-6)
    # same through the oracle restatement (pinned to the same vectors on the CPU)
    ref = LO.postprocess({"depth": gen}, torch.from_numpy(g["angle"]))
    assert rel_l2(out["points"].cpu(), ref["points"]) < 1e-5
    bare = LiDAR(H, W, 0.9, 120.0, angle_file=None)
    with pytest.raises(RuntimeError):
        bare.inv_to_xyz(inv.to(DEV))
    assert set(postprocess({"depth": gen.to(DEV)}, bare)) == {"depth"}


def write_kitti_tree(root, Hs, Ws, counts, seed=0):
    """<root>/sequences/<seq>/velodyne/<frame>.npy like process_kitti.py:77,116-118 writes them"""
    files = {}
    for seq, n in counts.items():
        d = os.path.join(root, "sequences", str(seq).zfill(2), "velodyne")
        os.makedirs(d, exist_ok=True)
        scans = make_scans(n, Hs, Ws, 4, seed + seq)
        for i in range(n):
            p = os.path.join(d, f"{i:06d}.npy")
            # one file per sequence is stored as float64: the loader's generic path (`.astype(np.float32)`,
            # datasets/kitti.py:81) instead of the read-into-pinned fast path
            np.save(p, scans[i].astype(np.float64) if i == 1 else scans[i])
            files[p] = scans[i]
    torch.save(torch.stack([torch.linspace(0.05, -0.42, Hs)[:, None].expand(Hs, Ws),
                            torch.linspace(math.pi, -math.pi, Ws)[None, :].expand(Hs, Ws)]).contiguous(),
               os.path.join(root, "angles.pt"))
    return files


def test_scan_loader_order_content_and_sharding(tmp_path):
    from dusty_gan_amd.datasets import KITTIOdometry, ScanLoader, define_dataset
    from dusty_gan_amd.utils.config import load_config
    files = write_kitti_tree(str(tmp_path), 8, 64, {0: 5, 1: 4, 8: 3, 11: 2})
    cfg = load_config(["dataset=kitti_odometry", f"dataset.root={tmp_path}", "dataset.shape=[8,32]"]).dataset
    ds = define_dataset(cfg, "train")
    assert isinstance(ds, KITTIOdometry) and len(ds) == 9  # sequences 00 and 01 are in the train split, 08 / 11 not
    assert len(define_dataset(cfg, "val")) == 3 and len(define_dataset(cfg, "test")) == 2
    assert ds.datalist == sorted(p for p in files if "/00/" in p or "/01/" in p)
    item = ds[4]
    ref = LO.scan_to_polar(files[ds.datalist[4]], (8, 32))
    assert item["mask"].dtype == torch.bool
    for k in ("xyz", "depth", "mask"):
        assert torch.equal(item[k].cpu(), ref[k]), k
    with pytest.raises(NotImplementedError):
        cfg.name = "nuscenes"
        define_dataset(cfg, "train")
    for world in (1, 2):
        seen = []
        for rank in range(world):
            loader = ScanLoader(ds, 2, DEV, world=world, rank=rank, num_workers=3, prefetch=2, want_xyz=True)
            want = LO.batches(LO.sampler_indices(len(ds), world, rank), 2)
            assert len(loader) == len(want)
            for epoch in range(2):  # the reference never calls set_epoch: both epochs see the same order
                got = list(loader)
                assert len(got) == len(want)
                for out, idxs in zip(got, want):
                    for j, i in enumerate(idxs):
                        ref = LO.scan_to_polar(files[ds.datalist[i]], (8, 32))
                        assert torch.equal(out["depth"][j].cpu(), ref["depth"])
                        assert torch.equal(out["mask"][j].cpu() > 0, ref["mask"])
                        assert torch.equal(out["xyz"][j].cpu(), ref["xyz"])
            seen += [i for b in want for i in b]
        assert len(set(seen)) >= len(ds) - (2 * world - 1)  # drop_last loses at most one short batch per rank
    # flip=True: every sample is the flipped or the unflipped item, both occur, and the draw is seeded
    dsf = KITTIOdometry(str(tmp_path), "train", shape=(8, 32), flip=True)
    runs = []
    for _ in range(2):
        loader = ScanLoader(dsf, 3, DEV, num_workers=2)
        kinds = []
        for out, idxs in zip(loader, LO.batches(LO.sampler_indices(len(dsf), 1, 0), 3)):
            for j, i in enumerate(idxs):
                a = LO.scan_to_polar(files[dsf.datalist[i]], (8, 32), flip=False)["depth"]
                b = LO.scan_to_polar(files[dsf.datalist[i]], (8, 32), flip=True)["depth"]
                got = out["depth"][j].cpu()
                assert torch.equal(got, a) or torch.equal(got, b)
                kinds.append(torch.equal(got, b))
        runs.append(kinds)
    assert runs[0] == runs[1] and any(runs[0]) and not all(runs[0])
    with pytest.raises(FileNotFoundError):
        ScanLoader(KITTIOdometry(str(tmp_path), "custom", shape=(8, 32)), 2, DEV)  # sequence 16: no files


def test_trainer_trains_from_scan_files(tmp_path):
    """dataset=kitti_odometry end to end: file loader -> fetch_reals -> eager warm-up steps -> hipGraph replays;
    generate() adds the point map from angles.pt"""
    from dusty_gan_amd.trainers.dcgan_amp import Trainer
    from dusty_gan_amd.utils.config import load_config
    write_kitti_tree(str(tmp_path), 32, 128, {0: 7, 3: 6})
    cfg = load_config(["model=dusty2_dcgan_eqlr", "dataset=kitti_odometry", f"dataset.root={tmp_path}",
                       "dataset.shape=[32,64]", "model.gen.in_ch=8", "model.gen.ch_base=4", "model.gen.ch_max=16",
                       "model.dis.ch_base=4", "model.dis.ch_max=16", "solver.batch_size=4", "enable_amp=false"])
    tr = Trainer(cfg, {"gpu": 0, "ngpus": 1, "batch_size": 4, "num_workers": 2})
    assert len(tr.dataset) == 13
    x, m = tr.fetch_reals(next(tr.loader))
    assert x.shape == (4, 1, 32, 64) and float(x.min()) >= -1.0 and float(x.max()) <= 1.0
    assert torch.equal(x[m == 0], torch.full_like(x[m == 0], -1.0))
    for i in range(6):  # 3 batches per epoch: crosses an epoch boundary and reaches the graph replay
        s = tr.step(i)
        assert all(np.isfinite(v) for v in s.values()), (i, dict(s.items()))
    assert tr._graph is not None
    # G_ema keeps changing under graph replays (the fused Adam+EMA kernel writes its master through raw pointers): an
    # evaluation after further replays must see the new weights, and match an fp32 evaluation of the master
    first = tr.generate(ema=True)["depth_orig"].clone()
    for i in range(6, 9):
        tr.step(i)
    out = tr.generate(ema=True)
    assert not torch.equal(first, out["depth_orig"])
    Ge = {k: v.detach().cpu().clone() for k, v in tr.G_ema.state_dict().items()}
    ref = O.generator_backbone(Ge, tr.fixed_noise.cpu())  # depth_orig = tanh(depth head): no Gumbel noise involved
    assert float((out["depth_orig"].cpu() - ((ref["depth"] + 1) / 2).clamp(0, 1)).abs().max()) < 1e-4
    assert out["points"].shape == (4, 3, 32, 64) and torch.isfinite(out["points"]).all()
    assert out["normals"].shape == (4, 3, 32, 64) and 0.0 <= float(out["normals"].min()) <= float(out["normals"].max()) <= 1.0
    assert float(out["depth"].min()) >= 0.0 and float(out["depth"].max()) <= 1.0
    r = out["points"].norm(dim=1, keepdim=True)  # unit space: |p| = metric depth / max_depth wherever a point exists
    assert float(r.max()) <= 1.0 + 1e-5


def test_validation_end_to_end(tmp_path):
    """Trainer.validation() (reference :342-393) on a file dataset: the real branch (val split -> fetch_reals ->
    inv_to_xyz -> FPS) against the oracle chain sample by sample, the scores against the oracle metrics evaluated on
    the same sets, and the reference's result keys"""
    from oracle import metrics_oracle as MO
    from dusty_gan_amd.trainers.dcgan_amp import Trainer
    from dusty_gan_amd.utils.config import load_config
    files = write_kitti_tree(str(tmp_path), 32, 128, {0: 4, 8: 7})
    cfg = load_config(["model=dusty1_dcgan_eqlr", "dataset=kitti_odometry", f"dataset.root={tmp_path}",
                       "dataset.shape=[32,64]", "model.gen.in_ch=8", "model.gen.ch_base=4", "model.gen.ch_max=16",
                       "model.dis.ch_base=4", "model.dis.ch_max=16", "solver.batch_size=4", "enable_amp=false",
                       "solver.validation.num_points=96"])
    tr = Trainer(cfg, {"gpu": 0, "ngpus": 1, "batch_size": 4, "num_workers": 2})
    tr.step(0)
    scores, data = tr.validation(return_data=True)
    N = 7
    assert len(tr.val_dataset) == N and len(tr.val_loader) == 2  # drop_last=False: 4 + 3
    assert data["real-2d"].shape == (N, 1, 32, 64) and data["fake-3d"].shape == (N, 96, 3)
    # N = 7 > local batch 4: the second generated batch must not be a second copy of the first (the generator returns
    # views of one workspace; validation() has to keep copies)
    f2 = data["fake-2d"]
    assert f2.shape == (N, 1, 32, 64) and not torch.equal(f2[:3], f2[4:7])
    assert len({float(f2[i].sum()) for i in range(N)}) == N
    want_keys = {"swd-16", "swd-32", "swd-mean", "jsd", "mmd-cd", "mmd-sample-cd", "cov-cd"} | {
        f"1-nn-{k}-cd" for k in ("tp", "fp", "fn", "tn", "precision", "recall", "accuracy_t", "accuracy_f", "accuracy")}
    assert set(scores) == want_keys and all(np.isfinite(v) for v in scores.values())
    # the real branch, sample by sample (the val loader shuffles: match samples by content)
    angle = LO.init_coordmap(torch.load(os.path.join(tmp_path, "angles.pt")), 32, 64)
    val_paths = sorted(p for p in files if "/08/" in p)
    refs = []
    for p in val_paths:
        o = LO.scan_to_polar(files[p], (32, 64))
        x, _ = O.fetch_reals(o["depth"][None], o["mask"][None])
        pts = LO.inv_to_xyz(((x + 1) / 2).clamp(0, 1), angle).flatten(2).transpose(1, 2)[0].numpy()
        refs.append((x[0], pts[MO.fps(pts, 96)]))
    got2d, got3d = data["real-2d"].cpu(), data["real-3d"].cpu()
    used = set()
    for i in range(N):
        j = min(range(N), key=lambda k: float((got2d[i] - refs[k][0]).abs().max()))
        used.add(j)
        assert float((got2d[i] - refs[j][0]).abs().max()) < 1e-5
        # FPS picks indices from distances: a last-bit difference in the point map may swap near-ties, so compare the
        # sampled SET through its Chamfer distance to the oracle's sample rather than index by index
        d = MO.pairwise_cd(got3d[i][None].numpy(), refs[j][1][None])
        assert float(d) < 1e-8, (i, float(d))
    assert used == set(range(N))
    # the scores, from the same sets, through the oracle
    f3, r3 = data["fake-3d"].cpu().numpy(), data["real-3d"].cpu().numpy()
    assert abs(scores["jsd"] - MO.compute_jsd(f3 / 2.0, r3 / 2.0)) < 1e-5
    for k, v in MO.compute_cov_mmd_1nna(f3, r3).items():
        assert abs(scores[k] - v) <= 1e-5 * max(1.0, abs(v)), (k, scores[k], v)
    # synthetic dataset: nominal angle grid, the resident pool is the validation set
    cfg2 = load_config(["model=dcgan_eqlr", "dataset=synthetic", "dataset.shape=[32,64]", "model.gen.in_ch=8",
                        "model.gen.ch_base=4", "model.gen.ch_max=16", "model.dis.ch_base=4", "model.dis.ch_max=16",
                        "solver.batch_size=4", "enable_amp=false", "solver.validation.num_points=64", "dataset.pool=2"])
    tr2 = Trainer(cfg2, {"gpu": 0, "ngpus": 1, "batch_size": 4, "num_workers": 0})
    s2 = tr2.validation()
    assert set(s2) == want_keys and all(np.isfinite(v) for v in s2.values())
