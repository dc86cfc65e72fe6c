#!/bin/bash
# Same-box A/B of an environment switch on the replayed training step: R alternating rounds of bench.py with VAR=a / VAR=b.
#   usage: scripts/ab_env.sh R VAR a b [bench args]
cd ${GRAFT_REPO_ROOT:-.}
R=$1; VAR=$2; A=$3; B=$4; shift; shift; shift; shift
for r in $(seq 1 $R); do
  for v in $A $B; do
    env $VAR=$v python bench.py --no-other-configs --no-cpu-baseline --no-roofline "$@" 2>/dev/null | tail -1 > gpurun_out/abe_${v}_$r.json
  done
done
python3 - "$R" "$VAR" "$A" "$B" <<'PY'
import json, sys
R = int(sys.argv[1])
for v in sys.argv[3:5]:
    d = [json.load(open(f"gpurun_out/abe_{v}_{r}.json")) for r in range(1, R + 1)]
    print(sys.argv[2], "=", v, "device p50 ms:", [x["step_ms_device"]["p50"] for x in d], "ms_per_step:", [x["ms_per_step"] for x in d])
PY
