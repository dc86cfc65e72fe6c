"""Gradient accumulation schedule -- reference: utils/context_manager.py:21-35.

The reference wraps all but the last micro-batch in DDP.no_sync().  The engine has no DDP wrapper: gradients are
accumulated in the flat grad buffers and exchanged with ONE all-reduce per network after the last micro-batch, so
there is no context to enter; the generator keeps the reference's interface (it yields the micro-batch index)."""


def sync_round(i, num_accumulation, is_ddp=True):
    """True on the micro-batch whose gradients are exchanged (the reference's null_context round)."""
    return (i == num_accumulation - 1) or not is_ddp


def gradient_accumulation(num_accumulation, is_ddp=True, ddp_models=()):
    """yields i = 0 .. num_accumulation-1, like the reference (utils/context_manager.py:33-35)"""
    for i in range(num_accumulation):
        yield i
