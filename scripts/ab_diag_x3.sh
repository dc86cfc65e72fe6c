#!/bin/bash
# step time of the fp32x3 mode with the diagnostic library (make -C dusty_gan_amd/csrc diag DIAGBITS=<bits>, e.g. 128 = no conv
# output stores) against the product library, alternating on one box
for v in "" 1 "" 1; do
  DUSTY_GAN_LIB_DIAG=$v python bench.py --precision fp32x3 --no-cpu-baseline --no-other-configs --no-roofline --steps 20 --warmup 5 2>/dev/null > /tmp/ab_x3.json
  python - "${v:-0}" <<'PY'
import json, sys
d = json.loads(open("/tmp/ab_x3.json").read().strip().splitlines()[-1])
print("diag", sys.argv[1], d["ms_per_step"])
PY
done
