# thin_up_mfma variants (libdustygan_hip_diag_<name>.so built by `make variant`), one replayed step each: the kernel's three
# launches and the step's kernel time
for v in "" _old _w3 _w3r11 _w3r16 "" _old; do
  DUSTY_GAN_LIB_DIAG=$v bash scripts/step_sequence.sh gpurun_out/ab_tu > gpurun_out/ab_tu.txt 2>&1
  echo "variant '${v:-product}': $(grep thin_up_mfma gpurun_out/ab_tu.txt | awk '{printf "%s ", $6}') | $(tail -1 gpurun_out/ab_tu.txt)"
done
