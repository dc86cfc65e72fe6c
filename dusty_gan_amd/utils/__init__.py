"""Host-side helpers mirrored from the reference's utils/__init__.py (only the four on the hot path,
SURVEY.md §2) plus the engine's Philox stream."""


def set_requires_grad(net, requires_grad: bool = True):
    """reference: utils/__init__.py:59-61.  The engine has no autograd; kept so callers need not change."""
    for param in net.parameters():
        param.requires_grad = requires_grad


def sigmoid_to_tanh(x):
    """[0,1] -> [-1,+1]  (reference: utils/__init__.py:70-73)"""
    return x * 2.0 - 1.0


def tanh_to_sigmoid(x):
    """[-1,+1] -> [0,1]  (reference: utils/__init__.py:76-79)"""
    return (x + 1.0) / 2.0


def cycle(iterable):
    """reference: utils/__init__.py:110-113"""
    while True:
        for i in iterable:
            yield i
