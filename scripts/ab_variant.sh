#!/bin/bash
# Same-box A/B of compile-time variants of the ping-pong conv: builds each variant library on the GPU box, then times the
# layer set of scripts/bench_conv.py with the shipped library and every variant, alternating, R rounds.
#   usage: scripts/ab_variant.sh R name1 "flags1" [name2 "flags2" ...]
cd ${GRAFT_REPO_ROOT:-.}
R=$1; shift
names=()
while [ $# -gt 0 ]; do
  make -C dusty_gan_amd/csrc variant NAME=$1 VFLAGS="$2" > /dev/null 2>&1 || echo "build of $1 failed"
  names+=($1); shift; shift
done
for r in $(seq 1 $R); do
  python scripts/bench_conv.py bf16 32 convonly 2>&1 | grep -v amdgpu > gpurun_out/ab_base_$r.txt
  for n in "${names[@]}"; do
    DUSTY_GAN_LIB_DIAG=_$n python scripts/bench_conv.py bf16 32 convonly 2>&1 | grep -v amdgpu > gpurun_out/ab_${n}_$r.txt
  done
done
for f in gpurun_out/ab_*_?.txt; do echo "$f $(tail -1 $f)"; done
