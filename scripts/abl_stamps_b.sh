# Lifetime phases + K-step stamps of workgroup 0 for every fat layer at batch B (default 32), diagnostic build DIAGBITS=8
cd ${GRAFT_REPO_ROOT:-.}
B=${1:-32}
rm -f dusty_gan_amd/csrc/conv_mfma_pp_diag.o
make -C dusty_gan_amd/csrc diag DIAGBITS=8 > /dev/null 2>&1
DG_CONV_DBG=8 DUSTY_GAN_LIB_DIAG=1 python scripts/bench_conv.py bf16 $B convonly 2>&1 | grep -v amdgpu
