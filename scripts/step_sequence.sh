#!/bin/bash
# The launches of ONE replayed training step in stream order with their durations (rocprofv3 --kernel-trace): the run of
# kernels between the last two step-begin launches (step_prologue_kernel or dg_zero_multi_kernel).   usage: scripts/step_sequence.sh <outdir>
out=${1:-gpurun_out/seq}
root=$(pwd)
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $root/$out/kt -- python3 $root/bench.py --soak-steps 0 --no-clock --no-other-configs --no-cpu-baseline --no-roofline --steps 8 --warmup 4 $BENCH_ARGS > $root/$out/bench.json 2> $root/$out/kt.log
cd $root
f=$(find $out/kt -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "dg_zero_multi" in r["Kernel_Name"] or "step_prologue" in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]["Start_Timestamp"])
prev_end = t0
tot = 0
for r in rows[a:b]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    tot += e - s
    print(f'{(s - t0) / 1e3:9.1f} us  gap {(s - prev_end) / 1e3:6.1f}  dur {(e - s) / 1e3:7.1f}  {r["Kernel_Name"][:90]}')
    prev_end = e
print(f"launches {b - a}, kernel time {tot / 1e3:.1f} us, span {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us")
PY
rm -rf $out/kt
