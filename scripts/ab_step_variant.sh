#!/bin/bash
# Same-box A/B of a compile-time variant of the ping-pong conv INSIDE the replayed training step (bench.py): builds the
# variant library on the GPU box, then alternates shipped / variant runs R times.
#   usage: scripts/ab_step_variant.sh R name "flags" [bench args]
cd ${GRAFT_REPO_ROOT:-.}
R=$1; name=$2; flags=$3; shift; shift; shift
make -C dusty_gan_amd/csrc variant NAME=$name VFLAGS="$flags" > gpurun_out/ab_build_$name.log 2>&1 || echo "build of $name failed"
for r in $(seq 1 $R); do
  python bench.py --no-other-configs --no-cpu-baseline --no-roofline "$@" 2>/dev/null | tail -1 > gpurun_out/abs_base_$r.json
  DUSTY_GAN_LIB_DIAG=_$name python bench.py --no-other-configs --no-cpu-baseline --no-roofline "$@" 2>/dev/null | tail -1 > gpurun_out/abs_${name}_$r.json
done
python3 - "$name" "$R" <<'PY'
import json, sys
name, R = sys.argv[1], int(sys.argv[2])
for tag in ("base", name):
    v = [json.load(open(f"gpurun_out/abs_{tag}_{r}.json"))["step_ms_device"]["p50"] for r in range(1, R + 1)]
    print(tag, "device p50 ms per step:", v)
PY
