"""Dataset registry -- reference: datasets/__init__.py:4-27 (same names, same NotImplementedError behaviour)."""
from .scans import KITTIOdometry, ScanLoader, SparseMPO  # noqa: F401


def define_dataset(cfg, phase: str = "train", modality=("depth",)):
    kwargs = dict(root=cfg.root, split=phase, shape=cfg.shape, min_depth=cfg.min_depth, max_depth=cfg.max_depth,
                  flip=bool(cfg.flip) and phase == "train", modality=modality)
    if cfg.name == "kitti_odometry":
        return KITTIOdometry(**kwargs)
    if cfg.name == "sparse_mpo":
        return SparseMPO(**kwargs)
    raise NotImplementedError(cfg.name)
