# one replayed step per variant library: the launches of the kernel named in $1 and the step's kernel time
k=$1; shift
for v in "" "$@" ""; do
  DUSTY_GAN_LIB_DIAG=$v bash scripts/step_sequence.sh gpurun_out/ab_tk > gpurun_out/ab_tk.txt 2>&1
  echo "variant '${v:-product}': $(grep $k gpurun_out/ab_tk.txt | awk '{printf "%s ", $6}') | $(tail -1 gpurun_out/ab_tk.txt)"
done
