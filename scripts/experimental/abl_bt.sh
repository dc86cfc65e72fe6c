cd ${GRAFT_REPO_ROOT:-.}
for v in noepi:-DBT_NOEPI nodma:"-DBT_NOEPI -DBT_NODMA" noreads:"-DBT_NOEPI -DBT_NOREADS" mfma:"-DBT_NOEPI -DBT_NODMA -DBT_NOREADS"; do
  n=${v%%:*}; f=${v#*:}
  make -C dusty_gan_amd/csrc variant NAME=bt$n VSRC=conv_mfma_bt VFLAGS="$f" > /dev/null 2>&1 || echo build $n failed
  DUSTY_GAN_LIB_DIAG=_bt$n python scripts/bench_conv.py bf16 32 convonly 2>&1 | grep -v amdgpu > gpurun_out/bc_bt_$n.txt
done
