"""debug: do hipGraph memset nodes survive a stream synchronize between replays?"""
import torch
dev = "cuda"
for n in (7, 4096, 3_000_000):
    x = torch.full((n,), 5.0, device=dev)
    y = torch.zeros(n, device=dev)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        x.zero_()
        x += 1.0
        y += x
    for r in range(6):
        g.replay()
        if r == 2:
            torch.cuda.current_stream().synchronize()
        if r == 4:
            torch.cuda.synchronize()
        print(n, r, "x[:3]", x[:3].tolist(), "y[0]", float(y[0]), "x min/max", float(x.min()), float(x.max()), flush=True)
