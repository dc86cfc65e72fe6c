#!/bin/bash
# Same-box A/B of several PREBUILT variant libraries (dusty_gan_amd/csrc/libdustygan_hip_diag_<name>.so, `make variant`)
# inside the replayed training step: parity first (the conv op tests and the several-tiles-per-workgroup cases under each
# variant), then R alternating rounds of bench.py.
#   usage: scripts/ab_step_multi.sh R name1 [name2 ...]
cd ${GRAFT_REPO_ROOT:-.}
R=$1; shift
for n in "$@"; do
  DUSTY_GAN_LIB_DIAG=_$n timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_timed_path.py -x -q -m gpu -k "fwd_bwd_wgrad or persistent" 2>&1 | tail -3 > gpurun_out/abm_tests_$n.txt
done
for r in $(seq 1 $R); do
  python bench.py --no-other-configs --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > gpurun_out/abm_base_$r.json
  for n in "$@"; do
    DUSTY_GAN_LIB_DIAG=_$n python bench.py --no-other-configs --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > gpurun_out/abm_${n}_$r.json
  done
done
python3 - "$R" "$@" <<'PY'
import json, sys
R = int(sys.argv[1])
for tag in ["base"] + sys.argv[2:]:
    v = [json.load(open(f"gpurun_out/abm_{tag}_{r}.json"))["step_ms_device"]["p50"] for r in range(1, R + 1)]
    print(tag, "device p50 ms per step:", v)
for n in sys.argv[2:]:
    print(n, open(f"gpurun_out/abm_tests_{n}.txt").read().strip().splitlines()[-1])
PY
