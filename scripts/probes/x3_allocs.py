"""Probe: which of the split-storage buffers get (re)allocated while the step is being captured?"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

from dusty_gan_amd import engine as E
from tests.test_gpu_step import make_trainer

os.environ["DUSTY_GAN_FP32_SPLIT"] = "1"
real_empty = torch.empty
def empty(*a, **k):
    t = real_empty(*a, **k)
    if t.is_cuda and torch.cuda.is_current_stream_capturing():
        import traceback
        fr = traceback.extract_stack(limit=4)[:-1]
        print("ALLOC in capture", tuple(t.shape), t.dtype, " <- ".join(f"{f.name}:{f.lineno}" for f in reversed(fr)), flush=True)
    return t
torch.empty = empty
real_like = torch.empty_like
def empty_like(x, *a, **k):
    t = real_like(x, *a, **k)
    if t.is_cuda and torch.cuda.is_current_stream_capturing():
        import traceback
        fr = traceback.extract_stack(limit=4)[:-1]
        print("ALLOC(like) in capture", tuple(t.shape), " <- ".join(f"{f.name}:{f.lineno}" for f in reversed(fr)), flush=True)
    return t
torch.empty_like = empty_like
torch.manual_seed(31)
tr = make_trainer("dusty2", True, (64, 1024), 128, 64, 256, 8, amp=False)
for i in range(4):
    s = dict(tr.step(i).items())
    print(i, {k.split("loss/")[1]: round(v, 4) for k, v in s.items()}, tr.launch_mode(), flush=True)
