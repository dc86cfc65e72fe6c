"""Aggregate rocprofv3 --pmc counter_collection.csv files per kernel function and counter.
usage: python scripts/pmc_summary.py out.json a_counter_collection.csv [b_counter_collection.csv ...]
Prints a table and writes {kernel: {counter: {"sum":, "dispatches":, "per_dispatch":}}} to out.json."""
import collections
import csv
import json
import re
import sys


def short(name):
    """function identifier of an Itanium-mangled or plain kernel name (template arguments dropped)"""
    if not name.startswith("_Z"):
        return re.split(r"[<(]", name.replace("void ", "").replace("(anonymous namespace)::", ""))[0].strip()[:60]
    s, parts = name[2:], []
    if s.startswith("N"):
        s = s[1:]
    while s and s[0].isdigit():
        m = re.match(r"(\d+)", s)
        n = int(m.group(1))
        parts.append(s[m.end():m.end() + n])
        s = s[m.end() + n:]
    return "::".join(p for p in parts if not p.startswith("_GLOBAL__N"))[:60]


agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.Counter()
for path in sys.argv[2:]:
    with open(path, newline="") as f:
        for r in csv.DictReader(f):
            k = short(r["Kernel_Name"])
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[(k, r["Counter_Name"])] += 1
out = {}
for k in sorted(agg):
    out[k] = {}
    for c, v in sorted(agg[k].items()):
        n = cnt[(k, c)]
        out[k][c] = {"sum": v, "dispatches": n, "per_dispatch": v / n}
        print(f"{k:44s} {c:28s} sum {v:.5g}  n {n}  per-dispatch {v / n:.5g}")
json.dump(out, open(sys.argv[1], "w"), indent=1)
