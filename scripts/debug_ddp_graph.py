"""debug: 2 ranks on one GPU (gloo), segmented-graph replay at the benchmark's size; DBG_MODE=imm reads the scalars of
every step immediately, DBG_MODE=delay reads them one step later (bench.py's pattern)"""
import os, sys
sys.path.insert(0, ".")
import torch, torch.distributed as dist
from bench import make_trainer, parse

def main():
    sys.argv = [sys.argv[0], "--gpus", "2", "--batch", os.environ.get("DBG_B", "8")]
    args = parse()
    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    torch.cuda.set_device(0)
    dist.init_process_group("gloo")
    tr, arch = make_trainer(args, rank, 0, world)
    mode = os.environ.get("DBG_MODE", "imm")
    if mode == "bench":  # bench.py's exact flow
        last = None
        for i in range(4):
            last = tr.step(i)
        _ = list(last.values())
        v = os.environ.get("DBG_VAR", "0")
        if v == "0": torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
        if v == "1": torch.cuda.synchronize()
        if v == "2": dist.barrier()
        if v == "4":
            import time; time.sleep(0.5)
        if v == "5": torch.cuda.current_stream().synchronize()
        if v == "6":
            import time; time.sleep(0.3 * rank)
        prev = None
        for i in range(6):
            cur = tr.step(i)
            if prev is not None:
                v = list(prev.values())
                if rank == 0:
                    print(i, "graph", v[:3], flush=True)
            prev = cur
        if rank == 0:
            print(9, "graph", list(prev.values())[:3], flush=True)
        dist.destroy_process_group()
        return
    prev = None
    for i in range(int(os.environ.get("DBG_STEPS", "8"))):
        cur = tr.step(i)
        rd = cur if mode == "imm" else prev
        if mode == "none" and i < int(os.environ.get("DBG_STEPS", "8")) - 1:
            rd = None
        if rd is not None:
            s = dict(rd.items())
            if rank == 0:
                print(i, "graph" if tr._graph is not None else "eager", {k: round(v, 3) for k, v in list(s.items())[:3]}, flush=True)
        prev = cur
    dist.destroy_process_group()

main()
