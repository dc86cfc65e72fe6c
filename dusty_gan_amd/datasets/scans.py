"""File datasets of projected LiDAR scans and their device-side batch pipeline.

Reference: datasets/kitti.py:21-98 (KITTIOdometry), datasets/mpo.py:19-104 (SparseMPO), and the DataLoader /
DistributedSampler built around them at trainers/dcgan_amp.py:80-90.  Same constructor arguments, split tables, file
lists and sample order.  What differs is WHERE the per-sample arithmetic runs: the reference's worker processes do
norm / mask / normalise / hflip / NEAREST-resize with numpy + torchvision per item and ship float tensors through
pinned memory; here the host threads only read the `.npy` bytes into a pinned staging buffer, one copy moves the
raw batch (2 MB per 64x2048x4 scan) to HBM on a side stream, and one kernel (`dg_scan_to_polar`, csrc/lidar_io.hip)
produces the whole batch at the training resolution.  At ~9000 images/s a Python per-item pipeline is the
bottleneck; this one costs one 16-byte read per output pixel.
"""
import os.path as osp
from collections import deque
from concurrent.futures import ThreadPoolExecutor
from glob import glob

import numpy as np
import torch

from .. import _lib as L

KITTI_CONFIG = {
    "split": {
        "train": [0, 1, 2, 3, 4, 5, 6, 7, 9, 10],
        "val": [8],
        "test": [11, 12, 13, 14, 15, 16, 17, 18, 19, 20, 21],
        "custom": [16],
    },
}
MPO_CONFIG = {"split": {"train": [0, 1, 2, 3, 4, 5, 6], "val": [7], "test": [8, 9, 10]}}


def scan_to_polar(scan, shape, min_depth, max_depth, flip=None, want_xyz=False, drop_const=None):
    """One launch of dg_scan_to_polar over a device batch.
    scan [B,Hs,Ws,C] fp32 (device) -> dict(depth [B,1,H,W], mask [B,1,H,W] in {0,1}[, xyz [B,3,H,W]][, x_real])."""
    if not scan.is_cuda:
        raise RuntimeError("scan_to_polar runs on the GPU only (no CPU fallback)")
    scan = scan.contiguous()
    assert scan.dtype == torch.float32 and scan.dim() == 4
    B, Hs, Ws, C = scan.shape
    H, W = int(shape[0]), int(shape[1])
    f32 = dict(dtype=torch.float32, device=scan.device)
    out = {"depth": torch.empty(B, 1, H, W, **f32), "mask": torch.empty(B, 1, H, W, **f32)}
    if want_xyz:
        out["xyz"] = torch.empty(B, 3, H, W, **f32)
    if drop_const is not None:
        out["x_real"] = torch.empty(B, 1, H, W, **f32)
    if flip is not None:
        flip = flip.to(device=scan.device, dtype=torch.uint8).contiguous()
        assert flip.numel() == B
    L.check(L.lib().dg_scan_to_polar(L.ptr(scan), B, Hs, Ws, C, H, W, L.ptr(flip), float(min_depth), float(max_depth),
                                     float(drop_const if drop_const is not None else 0.0), L.ptr(out["depth"]),
                                     L.ptr(out["mask"]), L.ptr(out.get("xyz")), L.ptr(out.get("x_real")),
                                     L.stream_ptr()), "dg_scan_to_polar")
    return out


class ScanDataset:
    """file list + raw read; base of the two reference datasets (they share preprocess/transform line for line)"""

    CONFIG = None
    graph_safe = True  # ScanLoader yields fixed-shape device batches (see Trainer._graph_eligible)

    def __init__(self, root, split, shape=(64, 256), min_depth=0.9, max_depth=120.0, flip=False, config=None,
                 modality=("depth",)):
        self.root = self._root(root)
        self.split = split
        self.config = config if config is not None else self.CONFIG
        self.subsets = np.asarray(self.config["split"][split])
        self.shape = tuple(shape)
        self.min_depth = min_depth
        self.max_depth = max_depth
        self.flip = flip
        assert "depth" in modality, '"depth" is required'
        if "reflectance" in modality:
            raise NotImplementedError("reflectance modality (datasets/kitti.py:84-85) is not on the training path")
        self.modality = modality
        self.datalist = None
        self.load_datalist()

    def read(self, index):
        """the host half of __getitem__ (datasets/kitti.py:80-81): the projected scan [Hs,Ws,C] as stored"""
        return np.load(self.datalist[index])

    def __getitem__(self, index):
        """reference item {"xyz","depth","mask"} at `shape`, computed on the GPU (flip never applied here: the
        loader draws it per sample); `mask` is bool like the reference's"""
        scan = torch.from_numpy(np.ascontiguousarray(self.read(index), dtype=np.float32)).cuda()[None]
        out = scan_to_polar(scan, self.shape, self.min_depth, self.max_depth, want_xyz=True)
        return {"xyz": out["xyz"][0], "depth": out["depth"][0], "mask": out["mask"][0] > 0}

    def __len__(self):
        return len(self.datalist)

    def __repr__(self):
        head = "Dataset " + self.__class__.__name__
        body = ["Number of datapoints: {}".format(self.__len__()), "Root location: {}".format(self.root)]
        return "\n".join([head] + ["    " + line for line in body])


class KITTIOdometry(ScanDataset):
    """datasets/kitti.py:21-52"""
    CONFIG = KITTI_CONFIG

    def _root(self, root):
        return osp.join(root, "sequences")

    def load_datalist(self):
        datalist = []
        for subset in self.subsets:
            subset_dir = osp.join(self.root, str(subset).zfill(2))
            datalist += sorted(glob(osp.join(subset_dir, "velodyne/*")))
        self.datalist = datalist


class SparseMPO(ScanDataset):
    """datasets/mpo.py:19-52"""
    CONFIG = MPO_CONFIG

    def _root(self, root):
        return osp.join(root, "Data")

    def load_datalist(self):
        datalist = []
        for subset in self.subsets:
            datalist += sorted(glob(osp.join(self.root, "*_set{}_*.npy".format(str(subset).zfill(3)))))
        self.datalist = datalist


def sampler_indices(n, world, rank, seed=0, epoch=0, shuffle=True):
    """DistributedSampler(dataset) as built at trainers/dcgan_amp.py:88: shuffle with seed 0, pad to a multiple of the
    world size by repeating the head, take rank::world.  The reference never calls set_epoch, so every epoch replays
    epoch 0's permutation; `epoch` is here for callers that want to do better."""
    if shuffle:
        g = torch.Generator()
        g.manual_seed(seed + epoch)
        idx = torch.randperm(n, generator=g).tolist()
    else:
        idx = list(range(n))
    total = -(-n // world) * world
    pad = total - len(idx)
    if pad > 0:
        idx += (idx * (-(-pad // len(idx))))[:pad]
    return idx[rank:total:world]


class _Slot:
    def __init__(self, B, Hs, Ws, C, device):
        self.pinned = torch.empty(B, Hs, Ws, C, dtype=torch.float32).pin_memory()
        self.host = self.pinned.numpy()
        self.dev = torch.empty(B, Hs, Ws, C, dtype=torch.float32, device=device)
        self.copied = torch.cuda.Event()   # H2D of this slot finished -> the pinned buffer may be refilled
        self.consumed = None               # kernel that read `dev` finished -> `dev` may be overwritten
        self.futs, self.flip = [], None


class ScanLoader:
    """DataLoader(dataset, batch_size, sampler=DistributedSampler(dataset), drop_last=True, pin_memory, num_workers)
    of trainers/dcgan_amp.py:82-90 for a ScanDataset, yielding DEVICE batches {"depth" [B,1,H,W], "mask" [B,1,H,W]}
    (plus "xyz" on request).  One pass over `iter(loader)` is one epoch; wrap in utils.cycle like the reference.

    Pipeline per batch: `num_workers` host threads np.load into a pinned slot -> one async H2D copy of the raw batch on
    a side stream -> dg_scan_to_polar on the caller's stream.  `prefetch` batches are in flight ahead of the consumer.
    """
    graph_safe = True  # fixed-shape device batches: the hipGraph-replayed step can copy them into its static inputs

    def __init__(self, dataset, batch_size, device, world=1, rank=0, num_workers=4, prefetch=2, seed=0,
                 shuffle=True, want_xyz=False, drop_last=True):
        if len(dataset) == 0:
            raise FileNotFoundError(f"no scans under {dataset.root} for split '{dataset.split}'")
        self.dataset, self.B, self.device = dataset, int(batch_size), torch.device(device)
        self.world, self.rank, self.seed, self.shuffle, self.want_xyz = world, rank, seed, shuffle, want_xyz
        self.drop_last = drop_last  # False = the reference's validation loader (trainers/dcgan_amp.py:94-101)
        probe = dataset.read(0)
        if probe.ndim != 3 or probe.shape[-1] < 3:
            raise ValueError(f"{dataset.datalist[0]}: expected a [rings, points, >=3] array, got {probe.shape}")
        self.scan_shape = tuple(probe.shape)
        self.pool = ThreadPoolExecutor(max(1, int(num_workers)))
        self.copy_stream = torch.cuda.Stream(self.device)
        self.slots = [_Slot(self.B, *self.scan_shape, self.device) for _ in range(max(1, prefetch) + 1)]
        self.epoch = 0
        self.skip = 0  # batches of the NEXT epoch to pass over without reading (checkpoint resume: Trainer._restore_position)

    def __len__(self):
        n = len(sampler_indices(len(self.dataset), self.world, self.rank, shuffle=False))
        return n // self.B if self.drop_last else -(-n // self.B)

    def _read_into(self, dst, index):
        path = self.dataset.datalist[index]
        # fast path: a C-ordered float32 .npy payload is read straight into the pinned slot (no intermediate array)
        with open(path, "rb") as f:
            try:
                version = np.lib.format.read_magic(f)
                header = (np.lib.format.read_array_header_1_0 if version == (1, 0)
                          else np.lib.format.read_array_header_2_0)(f)
            except ValueError:
                header = None
            if header is not None:
                shape, fortran, dtype = header
                if tuple(shape) != self.scan_shape:
                    raise ValueError(f"{path}: shape {tuple(shape)} != {self.scan_shape}")
                if dtype == np.float32 and not fortran:
                    if f.readinto(memoryview(dst).cast("B")) != dst.nbytes:
                        raise ValueError(f"{path}: truncated file")
                    return
        arr = self.dataset.read(index)
        if arr.shape != self.scan_shape:
            raise ValueError(f"{path}: shape {arr.shape} != {self.scan_shape}")
        np.copyto(dst, arr, casting="same_kind")  # `.astype(np.float32)` of datasets/kitti.py:81

    def _submit(self, slot, idxs, rng):
        slot.copied.synchronize()  # no-op for a never-recorded event
        slot.futs = [self.pool.submit(self._read_into, slot.host[j], i) for j, i in enumerate(idxs)]
        slot.n = len(idxs)  # < B only for the last batch of a drop_last=False loader
        # `flip = self.flip and random.random() > 0.5` (datasets/kitti.py:70), drawn per sample from a seeded stream
        slot.flip = torch.from_numpy((rng.random(self.B) > 0.5).astype(np.uint8)) if self.dataset.flip else None

    def __iter__(self):
        idx = sampler_indices(len(self.dataset), self.world, self.rank, self.seed, 0, self.shuffle)
        stop = len(idx) - self.B + 1 if self.drop_last else len(idx)
        batches = [idx[i:i + self.B] for i in range(0, stop, self.B)]
        rng = np.random.default_rng([self.seed, self.rank, self.epoch])
        self.epoch += 1
        ds = self.dataset
        pending, nxt = deque(), min(self.skip, len(batches))
        for _ in range(nxt):  # resumed mid-epoch: keep the flip stream aligned with an uninterrupted run
            if ds.flip:
                rng.random(self.B)
        self.skip = 0
        free = deque(self.slots)
        while nxt < len(batches) and free:
            s = free.popleft()
            self._submit(s, batches[nxt], rng)
            pending.append(s)
            nxt += 1
        while pending:
            s = pending.popleft()
            for f in s.futs:
                f.result()  # re-raises reader errors here
            cur = torch.cuda.current_stream(self.device)
            with torch.cuda.stream(self.copy_stream):
                if s.consumed is not None:
                    self.copy_stream.wait_event(s.consumed)
                s.dev.copy_(s.pinned, non_blocking=True)
                s.copied.record(self.copy_stream)
            cur.wait_event(s.copied)
            out = scan_to_polar(s.dev[:s.n], ds.shape, ds.min_depth, ds.max_depth,
                                flip=None if s.flip is None else s.flip[:s.n], want_xyz=self.want_xyz)
            s.consumed = torch.cuda.Event()
            s.consumed.record(cur)
            if nxt < len(batches):
                self._submit(s, batches[nxt], rng)
                pending.append(s)
                nxt += 1
            yield out
