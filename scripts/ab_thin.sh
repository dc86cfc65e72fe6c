#!/bin/bash
# Same-box comparison of PREBUILT library variants on the thin (Down1 / Head) kernels: parity (tests/test_gpu_thin.py) and the
# per-kernel averages of scripts/bench_thin.py inside the eager step, then the replayed step (bench.py device p50).
#   usage: scripts/ab_thin.sh name1 [name2 ...]      ("base" = the shipped library)
cd ${GRAFT_REPO_ROOT:-.}
for n in "$@"; do
  if [ "$n" = base ]; then unset DUSTY_GAN_LIB_DIAG; else export DUSTY_GAN_LIB_DIAG=_$n; fi
  echo "== $n"
  timeout 600 python -m pytest tests/test_gpu_thin.py -x -q -m gpu 2>&1 | tail -1
  python scripts/bench_thin.py --no-other-configs --no-cpu-baseline 2>&1 | grep -v amdgpu
  for r in 1 2; do python bench.py --no-other-configs --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('step p50', d['step_ms_device']['p50'])"; done
done
