#!/bin/bash
# Ablation builds of the LDS-DMA weight-gradient kernel on the GPU box (make wgdiag WGDIAG=<bits>, DG_WG_DIAG in
# wgrad_mfma_dma.hip: 1 no DMA, 2 no MFMA, 4 no epilogue (partial-tile stores), 16 no LDS reads; sums combine): the "ws auto"
# column of scripts/bench_conv.py's weight-gradient part per build.   usage: scripts/abl_wgrad.sh [bits ...]
cd ${GRAFT_REPO_ROOT:-.}
bits=${@:-1 2 4 16 23}
python scripts/bench_conv.py bf16 32 wgradonly 2>&1 | grep -v amdgpu | tail -12 > gpurun_out/ablw_base.txt
for b in $bits; do
  make -C dusty_gan_amd/csrc wgdiag WGDIAG=$b > /dev/null 2>&1
  DUSTY_GAN_LIB_DIAG=1 python scripts/bench_conv.py bf16 32 wgradonly 2>&1 | grep -v amdgpu | tail -12 > gpurun_out/ablw_$b.txt
done
