"""debug: one process, single-graph replay: what breaks after torch.cuda.synchronize()?  DBG_SYNC: device | stream | none"""
import os, sys
sys.path.insert(0, ".")
import torch
from tests.test_gpu_step import make_trainer
mode = os.environ.get("DBG_SYNC", "device")
torch.manual_seed(300)
tr = make_trainer("none", True, (64, 1024), 512, 64, 512, int(os.environ.get("DBG_B", "8")), amp=os.environ.get("DBG_FP32") is None)

def pools():
    segs = torch.cuda.memory_snapshot()
    priv = [(s["address"], s["address"] + s["total_size"]) for s in segs if s.get("segment_pool_id", (0, 0)) != (0, 0)]
    return priv

for i in range(8):
    s = tr.step(i)
    if i % 2 == 1:
        if mode == "device": torch.cuda.synchronize()
        elif mode == "stream": torch.cuda.current_stream().synchronize()
    vals = [f"{x:.4g}" for x in s.values()]
    extra = ""
    if tr._graph is not None:
        priv = pools()
        t = torch.empty(7, device="cuda")
        inside = any(a <= t.data_ptr() < b for a, b in priv)
        inside_dev = any(a <= s._dev.data_ptr() < b for a, b in priv) if getattr(s, "_dev", None) is not None else None
        extra = f" private segs {len(priv)} fresh-alloc-in-private {inside} gout-in-private {any(a <= tr._g_out.data_ptr() < b for a, b in priv)} scal {[f'{x:.3g}' for x in tr._dev_scal.tolist()]}"
    print(mode, i, "graph" if tr._graph is not None else "eager", vals, extra, flush=True)
