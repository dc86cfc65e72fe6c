"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel family (short name) and counter."""
import csv, sys, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for path in sys.argv[1:]:
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            n = r["Kernel_Name"]
            if not any(t in n for t in ("conv_kernel", "conv_mfma_kernel", "wgrad_mfma_kernel")): continue
            k = n.split("Ev")[0][-60:]
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
for k in sorted(agg):
    print(k)
    for c, v in sorted(agg[k].items()):
        print(f"   {c:32s} {v:.4g}  ({cnt[(k,c)]} dispatches)")
