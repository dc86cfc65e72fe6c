"""How long the GPU idles around the step that captures the hipGraph (the driver's window of 20 steps after 5 warm-up steps
starts 2 replays behind it: an idle gap >= ~50 ms sends the clock down and the next ~15 steps ramp back up).
usage: python scripts/probes/capture_idle.py"""
import cProfile
import io
import pstats
import sys
import time

import torch

sys.path.insert(0, ".")
from tests.test_gpu_step import make_trainer  # noqa: E402

torch.manual_seed(1)
tr = make_trainer("none", True, (64, 1024), 512, 64, 512, 32, amp=True)
torch.cuda.synchronize()
for i in range(6):
    t0 = time.perf_counter()
    if i == 2:
        pr = cProfile.Profile()
        pr.enable()
    tr.step(i)
    if i == 2:
        pr.disable()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"step {i}: host {1e3 * (t1 - t0):8.1f} ms, + sync {1e3 * (t2 - t1):6.1f} ms   {tr.launch_mode()}", flush=True)
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(28)
print(s.getvalue()[:6000])
