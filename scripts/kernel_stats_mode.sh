#!/bin/bash
# per-kernel time table of bench.py --soak-steps 0 --no-clock in a given mode (rocprofv3 --kernel-trace --stats); usage: scripts/kernel_stats_mode.sh <outdir> [bench args]
out=${1:-gpurun_out/ks}; shift
root=$(pwd); mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $root/$out/kt -- python3 $root/bench.py --soak-steps 0 --no-clock --no-other-configs --no-cpu-baseline --no-roofline --steps 10 --warmup 4 "$@" > $root/$out/bench.json 2> $root/$out/kt.log
cd $root
f=$(find $out/kt -name "*kernel_stats.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:28]:
    print(f'{float(r["TotalDurationNs"]) / tot * 100:5.1f} %  calls {r["Calls"]:>5}  avg {float(r["AverageNs"]) / 1e3:9.1f} us  {r["Name"][:110]}')
PY
rm -rf $out/kt
