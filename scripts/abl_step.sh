#!/bin/bash
# In-step ablations of the ping-pong conv: the replayed training step (bench.py --soak-steps 0 --no-clock under rocprofv3 --kernel-trace) with the
# diagnostic library built for each DG_PP_DIAG bit set (1 no DMA, 2 no MFMA, 4 no epilogue, 16 no fragment reads, 64 no
# mask loads, 128 no output stores).  Unlike scripts/abl_conv.sh (one layer repeated on hot buffers) the kernels here
# run in the step's own order on cold operands.   usage: scripts/abl_step.sh [bits ...]
cd ${GRAFT_REPO_ROOT:-.}
bash scripts/step_sequence.sh gpurun_out/seq_abl_base > gpurun_out/seq_abl_base.txt 2>&1
for b in ${@:-4 2 1 16}; do
  rm -f dusty_gan_amd/csrc/conv_mfma_pp_diag.o
  make -C dusty_gan_amd/csrc diag DIAGBITS=$b > /dev/null 2>&1
  DUSTY_GAN_LIB_DIAG=1 bash scripts/step_sequence.sh gpurun_out/seq_abl_$b > gpurun_out/seq_abl_$b.txt 2>&1
done
