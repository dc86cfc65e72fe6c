#!/bin/bash
# Ablation builds of the ping-pong conv on the GPU box (make diag DIAGBITS=<bits>, DG_PP_DIAG in conv_mfma_pp.hip: 1 no DMA,
# 2 no MFMA, 4 no epilogue, 16 no fragment reads; sums combine): layer set of scripts/bench_conv.py per build.
#   usage: scripts/abl_conv.sh [bits ...]     (default 1 2 4 16)
cd ${GRAFT_REPO_ROOT:-.}
bits=${@:-1 2 4 16}
python scripts/bench_conv.py bf16 32 2>&1 | grep -v "wg" | tail -16 > gpurun_out/abl_base.txt
for b in $bits; do
  rm -f dusty_gan_amd/csrc/conv_mfma_pp_diag.o   # (the object does not depend on DIAGBITS: make would keep the last build)
  make -C dusty_gan_amd/csrc diag DIAGBITS=$b > /dev/null 2>&1
  DUSTY_GAN_LIB_DIAG=1 python scripts/bench_conv.py bf16 32 2>&1 | grep -v "wg" | tail -16 > gpurun_out/abl_$b.txt
done
