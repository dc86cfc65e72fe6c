"""debug: 2 ranks on one GPU (gloo), segmented-graph replay at the benchmark's size; which host-side call between
replays breaks the scalars?  DBG_VAR: none | sync | barrier | both | sleep"""
import os, sys, time
sys.path.insert(0, ".")
import torch, torch.distributed as dist, torch.multiprocessing as mp

def worker(rank, world, init_file, var):
    from tests.test_gpu_step import make_trainer
    dist.init_process_group("gloo", init_method=f"file://{init_file}", rank=rank, world_size=world)
    torch.manual_seed(300)
    tr = make_trainer("none", True, (64, 1024), 512, 64, 512, 8, amp=True)
    for i in range(8):
        s = tr.step(i)
        if i % 2 == 1:
            if var in ("sync", "both"): torch.cuda.synchronize()
            if var in ("barrier", "both"): dist.barrier()
            if var == "sleep": time.sleep(0.3)
        v = list(s.values())
        if rank == 0:
            print(var, i, "graph" if tr._graph is not None else "eager", [f"{x:.4g}" for x in v], flush=True)
    dist.barrier()
    dist.destroy_process_group()

if __name__ == "__main__":
    import tempfile
    for var in sys.argv[1:] or ["none", "sync", "barrier", "both"]:
        with tempfile.TemporaryDirectory() as td:
            mp.spawn(worker, args=(2, os.path.join(td, "init"), var), nprocs=2, join=True)
