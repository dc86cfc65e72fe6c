cd $GRAFT_REPO_ROOT
python scripts/bench_conv.py bf16 32 2>&1 | grep -v "wg" | tail -16 > gpurun_out/abl_base.txt
for bits in 1 2 4 16; do
  make -C dusty_gan_amd/csrc diag DIAGBITS=$bits > /dev/null 2>&1
  DUSTY_GAN_LIB_DIAG=1 python scripts/bench_conv.py bf16 32 2>&1 | grep -v "wg" | tail -16 > gpurun_out/abl_$bits.txt
done
