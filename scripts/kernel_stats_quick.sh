#!/bin/bash
# per-kernel averages of a short bench run (rocprofv3 --kernel-trace --stats): scripts/kernel_stats_quick.sh <outdir> [pattern]
out=${1:-gpurun_out/kq}
pat=${2:-.}
root=$(pwd)
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/kt -- python3 $root/bench.py --soak-steps 0 --no-clock --no-other-configs --no-cpu-baseline --no-roofline --steps 20 --warmup 5 $BENCH_ARGS > $root/$out/bench.json 2> $root/$out/kt.log
cd $root
f=$(find $out/kt -name "*kernel_stats.csv" | head -1)
python3 - "$f" "$pat" <<'PY'
import csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
print(f"kernel time total {tot / 1e6:.2f} ms over the run")
for r in rows:
    if re.search(sys.argv[2], r["Name"]):
        print(f'{r["Name"][:72]:72s} n={int(r["Calls"]):4d} avg {float(r["AverageNs"]) / 1e3:8.1f} us')
PY
rm -rf $out/kt
