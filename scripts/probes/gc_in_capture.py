"""Probe: does tests/test_gpu_step.py::test_capture_survives_an_eager_garbage_collector abort WITHOUT the collector guard of
Trainer._cap_open?  (It must, for the test to mean something.)"""
import sys

sys.path.insert(0, ".")
import torch

import tests.test_gpu_step as T
from dusty_gan_amd.trainers.dcgan_amp import Trainer


def raw_open(self):
    g = torch.cuda.CUDAGraph()
    ctx = torch.cuda.graph(g, pool=self._cap_pool, capture_error_mode="thread_local" if self._multi else "global")
    ctx.__enter__()
    self._cap_cur = (g, ctx)


Trainer._cap_open = raw_open
T.test_capture_survives_an_eager_garbage_collector()
print("survived without the guard")
