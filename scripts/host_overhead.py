import sys, time; sys.path.insert(0, ".")
import torch, argparse
import bench
args = argparse.Namespace(arch=None, shape=[64,1024], batch=32, gp=1.0, precision="bf16", no_augment=False)
tr, arch = bench.make_trainer(args, 0, 0, 1)
for i in range(5): tr.step(i)
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(20): tr.step(i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"host enqueue {1e3*(t1-t0)/20:.2f} ms/step ; with drain {1e3*(t2-t0)/20:.2f} ms/step")
