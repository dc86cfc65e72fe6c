"""dg_wgrad_reduce alone on the D phase's three layers (numel, splits as the step has them), rotating over 3 workspace
sets (3 x 80 MB > the 256 MB infinity cache with the outputs): HIP-event time per launch.
usage: python scripts/bench_reduce.py"""
import ctypes as C
import sys
import torch
sys.path.insert(0, ".")
from dusty_gan_amd import _lib as L

lib = L.lib()
dev = "cuda"
layers = [(16 * 256 * 512, 4), (16 * 128 * 256, 16), (16 * 64 * 128, 32)]
if len(sys.argv) > 1:
    layers = [(int(a.split(":")[0]), int(a.split(":")[1])) for a in sys.argv[1:]]
sets = []
for r in range(3):
    items = (L.DgWgradReduce * len(layers))()
    keep = []
    for i, (numel, splits) in enumerate(layers):
        ws = torch.randn(splits * numel, device=dev)
        dw = torch.zeros(numel, device=dev)
        keep += [ws, dw]
        items[i].ws, items[i].dw, items[i].numel, items[i].splits, items[i].accumulate = ws.data_ptr(), dw.data_ptr(), numel, splits, 1
    sets.append((items, keep))


def run(k):
    L.check(lib.dg_wgrad_reduce(sets[k % 3][0], len(layers), L.stream_ptr()))


for i in range(6):
    run(i)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 30
a.record()
for i in range(n):
    run(i)
b.record()
torch.cuda.synchronize()
mb = sum(nu * (s + 2) * 4 for nu, s in layers) / 1e6
us = a.elapsed_time(b) / n * 1e3
print(f"{us:.1f} us per launch, {mb:.0f} MB -> {mb / us / 1e3 * 1e3 / 1e3:.2f} TB/s")
# check
items, keep = sets[0]
for i, (numel, splits) in enumerate(layers):
    keep[2 * i + 1].zero_()
run(0)
torch.cuda.synchronize()
for i, (numel, splits) in enumerate(layers):
    ref = keep[2 * i].view(splits, numel).sum(0)
    err = float((keep[2 * i + 1] - ref).abs().max())
    assert err < 1e-3, err
