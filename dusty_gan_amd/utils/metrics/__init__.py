"""Validation metrics on MI355X -- reference: utils/metrics/ (SURVEY.md §8f row 3)."""
from .cov_mmd_1nna import compute_cov_mmd_1nna  # noqa: F401
from .distance import chamfer_dir, chamfer_distance_matrix, earth_mover_distance, emd_distance_matrix  # noqa: F401
from .jsd import compute_jsd  # noqa: F401
from .swd import compute_swd  # noqa: F401
