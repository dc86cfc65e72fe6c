cd $GRAFT_REPO_ROOT
for bits in 8 31; do
  rm -f dusty_gan_amd/csrc/conv_mfma_pp_diag.o
  make -C dusty_gan_amd/csrc diag DIAGBITS=$bits > /dev/null 2>&1
  DG_CONV_DBG=8 DUSTY_GAN_LIB_DIAG=1 python scripts/bench_conv.py bf16 32 convonly 2>&1 | grep -v amdgpu > gpurun_out/stamps_$bits.txt
done
