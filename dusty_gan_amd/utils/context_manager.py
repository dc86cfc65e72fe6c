"""Gradient accumulation schedule -- reference: utils/context_manager.py:21-35.

The reference wraps all but the last micro-batch in DDP.no_sync().  The engine has no DDP wrapper: gradients are
accumulated in the flat grad buffers and exchanged with ONE all-reduce per network after the last micro-batch, so
the schedule reduces to "which micro-batch syncs"."""


def gradient_accumulation(num_accumulation, is_ddp=True, ddp_models=()):
    """yields (i, sync) with sync True only on the final micro-batch (the reference's null_context round)."""
    for i in range(num_accumulation):
        yield i, (i == num_accumulation - 1) or not is_ddp
