#!/bin/bash
# On the GPU box: time .ab_prev/bench.py (built by scripts/ab_prev.sh) and ./bench.py alternately, `reps` times per arch.
# usage: scripts/ab_bench.sh [reps] [arch ...]
reps=${1:-2}; shift
archs=${@:-none dusty2}
for a in $archs; do
  for r in $(seq $reps); do
    for side in .ab_prev .; do
      python $side/bench.py --arch $a --no-other-configs --no-cpu-baseline 2>/dev/null | grep '^{' |
        python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$side'.ljust(8), '$a'.ljust(7), d['ms_per_step'], d['roofline'].get('frac'))"
    done
  done
done
